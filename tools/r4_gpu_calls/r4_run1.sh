# round 4, GPU call 1: the suite with the new tests, the bench line, and what wf_shade is sensitive to
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_01; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
for k in 1 2; do for V in base sh1 sh2 sh3 sh4 sh5 sh6; do
  echo -n "$V: " >> $O/ab_shade.txt
  NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1 >> $O/ab_shade.txt
done; done
cat $O/ab_shade.txt
