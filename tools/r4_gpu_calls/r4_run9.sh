cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_09; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) > $O/warm.log 2>&1; tail -3 $O/warm.log
for k in 1 2; do for V in base4 wsens; do
  echo -n "$V c5: " >> $O/ws.txt
  NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=c5 SPP=128 REPS=3 TIMEK=1 ENGINE=wavefront timeout 200 python tools/wf_probe.py 2>&1 | tail -1 >> $O/ws.txt
done; done
cat $O/ws.txt
