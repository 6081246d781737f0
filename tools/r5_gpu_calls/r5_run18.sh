# round 5, GPU call 18: the five-tap film kernel (film_gather5_kernel) against the looping one: bits (new test + the film / parity tests), time on the headline and C4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
echo "t = $SECONDS s"
for k in 1 2 3; do for V in five looping; do
  echo -n "$V: "; if [ $V = looping ]; then export NORI_HIP_FILM_GENERIC=1; else unset NORI_HIP_FILM_GENERIC; fi
  TIMEK=1 REPS=3 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/film5_ab_headline.txt 2>&1; cat $O/film5_ab_headline.txt
for k in 1 2; do for V in five looping; do
  echo -n "$V: "; if [ $V = looping ]; then export NORI_HIP_FILM_GENERIC=1; else unset NORI_HIP_FILM_GENERIC; fi
  WORKLOAD=c2-ao-icosphere TIMEK=1 REPS=3 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done > $O/film5_ab_c2.txt 2>&1; cat $O/film5_ab_c2.txt
echo "t = $SECONDS s"
