# round 5, GPU call 7: tail overlap -- the tail of batch k beside batch k + 1 (tools/tail_probe.py: pa5 table 2048^2 at 512 spp = 4 batches), the wavefront tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_07; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wavefront.py -m gpu -x -q > $O/pytest_wavefront.log 2>&1; echo "pytest rc $?" >> $O/pytest_wavefront.log; tail -5 $O/pytest_wavefront.log
echo "t = $SECONDS s"
WORKLOAD=c4-table-mis SPP=512 timeout 500 python tools/tail_probe.py > $O/tail_overlap_c4.txt 2>&1; cat $O/tail_overlap_c4.txt
echo "t = $SECONDS s"
