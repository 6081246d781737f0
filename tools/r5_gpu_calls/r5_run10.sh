# round 5, GPU call 10 (calls 8 and 9 ran a stale library -- the build before them had failed): tail overlap with the bulk on the caller's stream and
# the tail on the first mask bits; first chunks claimed per workgroup against static ones on the headline and the terrain; the wavefront tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_10; mkdir -p $O
WORKLOAD=c4-table-mis SPP=512 timeout 500 python tools/tail_probe.py > $O/tail_overlap_c4.txt 2>&1; cat $O/tail_overlap_c4.txt
echo "t = $SECONDS s"
for k in 1 2 3; do for V in claimed static; do
  echo -n "$V: "; if [ $V = static ]; then export NORI_HIP_WF_STATIC_FIRST=1; else unset NORI_HIP_WF_STATIC_FIRST; fi
  TIMEK=1 REPS=3 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/first_chunk_ab_headline.txt 2>&1; cat $O/first_chunk_ab_headline.txt
for k in 1 2; do for V in claimed static; do
  echo -n "$V: "; if [ $V = static ]; then export NORI_HIP_WF_STATIC_FIRST=1; else unset NORI_HIP_WF_STATIC_FIRST; fi
  WORKLOAD=c5-terrain-10m SPP=128 TIMEK=1 REPS=2 timeout 200 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/first_chunk_ab_c5.txt 2>&1; cat $O/first_chunk_ab_c5.txt
unset NORI_HIP_WF_STATIC_FIRST
echo "t = $SECONDS s"
timeout 600 python -m pytest tests/test_gpu_wavefront.py -m gpu -x -q > $O/pytest_wavefront.log 2>&1; echo "pytest rc $?" >> $O/pytest_wavefront.log; tail -5 $O/pytest_wavefront.log
echo "t = $SECONDS s"
