# round 5, GPU call 3: (a) what the WIDE node step is sensitive to on the terrain: 32 more VALU instructions / one more load per step;
# (b) FETCH_SIZE / WRITE_SIZE calibrated on this renderer's access patterns (tools/ubench_fetch.hip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_03; mkdir -p $O
export TMPDIR=/tmp
for k in 1 2; do for V in base wsens3 wsens1; do
  echo -n "$V: "; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=c5-terrain-10m SPP=128 TIMEK=1 REPS=2 timeout 200 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/wide_sensitivity.txt 2>&1; cat $O/wide_sensitivity.txt
echo "t = $SECONDS s"
timeout 120 nori_amd/lib/ubench_fetch > $O/ubench_fetch.txt 2>&1; cat $O/ubench_fetch.txt
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/uf_r -o f -- nori_amd/lib/ubench_fetch > $O/uf_r.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/uf_w -o w -- nori_amd/lib/ubench_fetch > $O/uf_w.log 2>&1
python tools/ubench_fetch_summary.py /tmp/uf_r /tmp/uf_w > $O/ubench_fetch_counters.txt 2>&1; cat $O/ubench_fetch_counters.txt
find /tmp/uf_r /tmp/uf_w -name '*counter_collection.csv' | while read f; do cp $f $O/$(basename $(dirname $(dirname $f)))_$(basename $f); done
echo "t = $SECONDS s"
