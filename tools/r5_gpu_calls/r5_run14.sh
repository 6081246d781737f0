# round 5, GPU call 14: the host tree optimised by re-insertion (default on, kept when the SAH cost drops by 2 %) -- the whole GPU suite, the C4 line with and without it, the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_14; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
echo "t = $SECONDS s"
timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 --no-cpu-baseline > $O/c4_bench.json 2> $O/c4_bench.err; head -c 300 $O/c4_bench.json; echo
NORI_HIP_REINSERT=0 timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 --no-cpu-baseline > $O/c4_bench_no_reinsertion.json 2>> $O/c4_bench.err; head -c 300 $O/c4_bench_no_reinsertion.json; echo
echo "t = $SECONDS s"
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
echo "t = $SECONDS s"
