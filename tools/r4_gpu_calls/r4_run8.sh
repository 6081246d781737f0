cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_08; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) > $O/warm.log 2>&1; tail -3 $O/warm.log
P="env REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py"
for k in 1 2; do
echo -n "default 32/16/24: " >> $O/sweep.txt; $P 2>&1 | tail -1 >> $O/sweep.txt
echo -n "refill 24: " >> $O/sweep.txt; NORI_HIP_WF_REFILL=24 $P 2>&1 | tail -1 >> $O/sweep.txt
echo -n "refill 16: " >> $O/sweep.txt; NORI_HIP_WF_REFILL=16 $P 2>&1 | tail -1 >> $O/sweep.txt
echo -n "refill 24 leaf 12: " >> $O/sweep.txt; NORI_HIP_WF_REFILL=24 NORI_HIP_WF_LEAF=12 $P 2>&1 | tail -1 >> $O/sweep.txt
echo -n "refill 40: " >> $O/sweep.txt; NORI_HIP_WF_REFILL=40 $P 2>&1 | tail -1 >> $O/sweep.txt
done
cat $O/sweep.txt
echo -n "c4 64spp: " >> $O/other.txt; WORKLOAD=c4 SPP=64 $P 2>&1 | tail -1 >> $O/other.txt
echo -n "c4 128spp (one batch at 2^29): " >> $O/other.txt; WORKLOAD=c4 SPP=128 $P 2>&1 | tail -1 >> $O/other.txt
echo -n "c4 128spp, batches of 2^28: " >> $O/other.txt; WORKLOAD=c4 SPP=128 PATHS=268435456 $P 2>&1 | tail -1 >> $O/other.txt
echo -n "c5 512spp (one batch): " >> $O/other.txt; WORKLOAD=c5 $P 2>&1 | tail -1 >> $O/other.txt
echo -n "c5 512spp, batches of 2^28: " >> $O/other.txt; WORKLOAD=c5 PATHS=268435456 $P 2>&1 | tail -1 >> $O/other.txt
cat $O/other.txt
