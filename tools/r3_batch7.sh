cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b7; mkdir -p $O
for TM in 8 1; do
rm -rf /tmp/tr; TILE_MOD=$TM REPS=2 NORI_HIP_CENSUS=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python tools/wf_probe.py > $O/trace_$TM.log 2>&1
echo "== tile_mod $TM"; grep "path slots" $O/trace_$TM.log | tail -14; python tools/trace_tail.py /tmp/tr 30
done
