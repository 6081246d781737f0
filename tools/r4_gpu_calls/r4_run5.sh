cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_05; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) > $O/warm.log 2>&1; tail -4 $O/warm.log
for k in 1 2 3; do for V in diet film1; do
  echo -n "$V: " >> $O/ab.txt
  NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so REPS=3 TIMEK=1 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1 >> $O/ab.txt
done; done
cat $O/ab.txt
NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_film1.so timeout 600 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_parity.py tests/test_gpu_goldens.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
