cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_06; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) > $O/warm.log 2>&1; tail -4 $O/warm.log
P="env REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py"
for k in 1 2; do
echo -n "1 pipe: " >> $O/pipes.txt; $P 2>&1 | tail -1 >> $O/pipes.txt
echo -n "2 pipes: " >> $O/pipes.txt; NORI_HIP_WF_PIPES=2 $P 2>&1 | tail -1 >> $O/pipes.txt
echo -n "2 pipes, extend 2 wg/cu: " >> $O/pipes.txt; NORI_HIP_WF_PIPES=2 NORI_HIP_WF_EXTEND_WGS_PER_CU=2 $P 2>&1 | tail -1 >> $O/pipes.txt
done
cat $O/pipes.txt
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
