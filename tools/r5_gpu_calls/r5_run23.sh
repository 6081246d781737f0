# round 5, GPU call 23: the tree as it ends the round -- GPU suite, smoke(), the bench line as the driver runs it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_23; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; a=json.load(open('$O/bench.json')); print(a['value'], a['ms_per_step'], a['steps'], a['warmup'], a['pass'], a['roofline']['frac'], a['roofline'].get('traffic'), a['roofline'].get('bound_detail'), a['cpu_baseline']['value'], a['parity']['ok'])"
echo "t = $SECONDS s"
