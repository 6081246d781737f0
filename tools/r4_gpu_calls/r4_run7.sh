cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_07; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) > $O/warm.log 2>&1; tail -4 $O/warm.log
export NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_tail.so
# correctness of the new tail: the wavefront suite on the variant (the knobs test compares frames across tail settings bit for bit)
timeout 600 python -m pytest tests/test_gpu_wavefront.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
P="env REPS=3 TIMEK=1 ENGINE=wavefront timeout 300 python tools/wf_probe.py"
for k in 1 2; do
echo -n "cbox tail: " >> $O/tail.txt; $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "cbox finish: " >> $O/tail.txt; NORI_HIP_WF_TAIL=0 $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "cbox 1/8 share tail: " >> $O/tail.txt; TILE_MOD=8 $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "cbox 1/8 share finish: " >> $O/tail.txt; TILE_MOD=8 NORI_HIP_WF_TAIL=0 $P 2>&1 | tail -1 >> $O/tail.txt
done
echo -n "c4 tail (64 spp): " >> $O/tail.txt; WORKLOAD=c4 SPP=64 $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "c4 finish (64 spp): " >> $O/tail.txt; WORKLOAD=c4 SPP=64 NORI_HIP_WF_TAIL=0 $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "c4 tail, 8 wg/cu: " >> $O/tail.txt; WORKLOAD=c4 SPP=64 NORI_HIP_WF_TAIL_WGS=8 $P 2>&1 | tail -1 >> $O/tail.txt
echo -n "c4 tail, 2 wg/cu: " >> $O/tail.txt; WORKLOAD=c4 SPP=64 NORI_HIP_WF_TAIL_WGS=2 $P 2>&1 | tail -1 >> $O/tail.txt
cat $O/tail.txt
