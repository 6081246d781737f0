# round 5, GPU call 12: the whole GPU suite (new: tail overlap, out-of-memory halving, device builder at 10 M triangles, the seven scene x integrator fixtures,
# a caller's stream through the reference-order ranks), then the C4 bench line at its full size (8 batches, 7 tails beside the next batch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_12; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -18 $O/pytest.log
echo "t = $SECONDS s"
timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 > $O/c4_bench.json 2> $O/c4_bench.err; tail -c 2500 $O/c4_bench.json
echo "t = $SECONDS s"
NORI_HIP_WF_TAIL_CUS=0 timeout 400 python bench.py --workload c4-table-mis --steps 3 --warmup 1 --no-cpu-baseline > $O/c4_bench_tails_serial.json 2>> $O/c4_bench.err; head -c 400 $O/c4_bench_tails_serial.json
echo "t = $SECONDS s"
