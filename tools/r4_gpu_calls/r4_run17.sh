cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_17; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
for k in 1 2 3; do
  echo -n "product: " >> $O/ab.txt; REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1 >> $O/ab.txt
  echo -n "filmpf: " >> $O/ab.txt; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_filmpf.so REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1 >> $O/ab.txt
done
cat $O/ab.txt
NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_filmpf.so timeout 300 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_goldens.py -m gpu -x -q 2>&1 | tail -3
