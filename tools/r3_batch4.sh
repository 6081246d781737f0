cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
REPS=3 TIMEK=1 timeout 100 python tools/wf_probe.py 2>&1 | tail -3
