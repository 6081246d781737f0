# round 5, GPU call 4: wide node step with the slot visits as selects (A/B on the terrain), then the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_04; mkdir -p $O
for k in 1 2; do for V in base wsel0; do
  echo -n "$V: "; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=c5-terrain-10m SPP=128 TIMEK=1 REPS=2 timeout 200 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/wide_visit_selects_ab.txt 2>&1; cat $O/wide_visit_selects_ab.txt
echo "t = $SECONDS s"
timeout 600 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
echo "t = $SECONDS s"
