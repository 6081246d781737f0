# round 5, GPU call 11: tail overlap, the bulk on a non-blocking stream of the engine (the caller's legacy default stream ran nothing beside the tail stream: call 10)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_11; mkdir -p $O
WORKLOAD=c4-table-mis SPP=512 timeout 500 python tools/tail_probe.py > $O/tail_overlap_c4.txt 2>&1; cat $O/tail_overlap_c4.txt
echo "t = $SECONDS s"
timeout 600 python -m pytest tests/test_gpu_wavefront.py -m gpu -x -q > $O/pytest_wavefront.log 2>&1; echo "pytest rc $?" >> $O/pytest_wavefront.log; tail -5 $O/pytest_wavefront.log
echo "t = $SECONDS s"
