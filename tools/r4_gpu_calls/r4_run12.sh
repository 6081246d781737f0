cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_12; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
P="env WORKLOAD=c5 SPP=128 REPS=3 TIMEK=1 ENGINE=wavefront timeout 200 python tools/wf_probe.py"
echo -n "6 wg, 113 nodes: " >> $O/occ.txt; $P 2>&1 | tail -1 >> $O/occ.txt
echo -n "6 wg, 40 nodes: " >> $O/occ.txt; NORI_HIP_TOP_NODES=40 $P 2>&1 | tail -1 >> $O/occ.txt
echo -n "7 wg, 75 nodes: " >> $O/occ.txt; NORI_HIP_TOP_NODES=75 NORI_HIP_WF_EXTEND_WGS_PER_CU=7 $P 2>&1 | tail -1 >> $O/occ.txt
echo -n "8 wg, 40 nodes: " >> $O/occ.txt; NORI_HIP_TOP_NODES=40 NORI_HIP_WF_EXTEND_WGS_PER_CU=8 $P 2>&1 | tail -1 >> $O/occ.txt
echo -n "8 wg, 0 nodes: " >> $O/occ.txt; NORI_HIP_NO_TOP_IMAGE=1 NORI_HIP_WF_EXTEND_WGS_PER_CU=8 $P 2>&1 | tail -1 >> $O/occ.txt
cat $O/occ.txt
