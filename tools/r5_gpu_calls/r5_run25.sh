# round 5, GPU call 25: wf_shade regrouping a round's paths by the material they hit (in-workgroup counting sort through LDS): C4 and veach / cbox_mis timing, the wavefront + parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_25; mkdir -p $O
for W in c4-table-mis; do WORKLOAD=$W SPP=128 TIMEK=1 REPS=3 timeout 200 python tools/wf_probe.py 2>&1 | tail -3; done > $O/regroup_c4.txt 2>&1; cat $O/regroup_c4.txt
echo "t = $SECONDS s"
timeout 900 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_parity.py tests/test_gpu_goldens.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
echo "t = $SECONDS s"
