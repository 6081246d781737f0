# round 5, GPU call 13: spatial splits in the host builder (default: budget 0.3, margin 0.99) -- the whole GPU suite, the C4 bench line with and without them, the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_13; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -18 $O/pytest.log
echo "t = $SECONDS s"
timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 > $O/c4_bench.json 2> $O/c4_bench.err; head -c 300 $O/c4_bench.json; echo
NORI_HIP_SBVH=0 timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 --no-cpu-baseline > $O/c4_bench_object_splits_only.json 2>> $O/c4_bench.err; head -c 300 $O/c4_bench_object_splits_only.json; echo
echo "t = $SECONDS s"
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
echo "t = $SECONDS s"
