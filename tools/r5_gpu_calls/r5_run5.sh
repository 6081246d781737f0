# round 5, GPU call 5 (session 2): the tree at the start of the session -- GPU suite, the bench line, kernel stats of the bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_05; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
echo "t = $SECONDS s"
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
echo "t = $SECONDS s"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $O/r5_05_kernel_stats.csv \;
head -8 $O/r5_05_kernel_stats.csv
echo "t = $SECONDS s"
