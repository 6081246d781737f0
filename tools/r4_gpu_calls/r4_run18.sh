cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_18; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
V=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_fin.so
NORI_HIP_LIBRARY=$V timeout 200 python -m pytest tests/test_gpu_wavefront.py -m gpu -x -q 2>&1 | tail -3
P="env REPS=3 TIMEK=1 ENGINE=wavefront timeout 300 python tools/wf_probe.py"
for k in 1 2; do
echo -n "c4 64spp product: " >> $O/fin.txt; WORKLOAD=c4 SPP=64 $P 2>&1 | tail -1 >> $O/fin.txt
echo -n "c4 64spp finish with image: " >> $O/fin.txt; NORI_HIP_LIBRARY=$V WORKLOAD=c4 SPP=64 $P 2>&1 | tail -1 >> $O/fin.txt
done
echo -n "cbox product: " >> $O/fin.txt; $P 2>&1 | tail -1 >> $O/fin.txt
echo -n "cbox finish with image: " >> $O/fin.txt; NORI_HIP_LIBRARY=$V $P 2>&1 | tail -1 >> $O/fin.txt
echo -n "cbox 1/8 product: " >> $O/fin.txt; TILE_MOD=8 $P 2>&1 | tail -1 >> $O/fin.txt
echo -n "cbox 1/8 finish with image: " >> $O/fin.txt; NORI_HIP_LIBRARY=$V TILE_MOD=8 $P 2>&1 | tail -1 >> $O/fin.txt
cat $O/fin.txt
