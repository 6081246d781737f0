# round 5, GPU call 1: the CU split (traversal stream / shading stream) -- test, sweep on the headline, C4, C5; wf_shade per material set A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_01; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_wavefront.py -x -q > $O/pytest_wavefront.log 2>&1; echo "pytest rc $?" >> $O/pytest_wavefront.log; tail -3 $O/pytest_wavefront.log
echo "t = $SECONDS s"
SPLITS="0 16 32 48 64 80 96 112 128 0" REPS=3 timeout 200 python tools/split_sweep.py > $O/split_headline.txt 2>&1; cat $O/split_headline.txt | cut -c1-260
echo "t = $SECONDS s"
TIMEK=1 ROUNDS=2 bash tools/ab.sh base v5 v5n v4n > $O/shade_matset_ab.txt 2>&1; cat $O/shade_matset_ab.txt | cut -c1-200
echo "t = $SECONDS s"
WORKLOAD=c4-table-mis SPP=128 SPLITS="0 32 64 96 0" REPS=2 timeout 200 python tools/split_sweep.py > $O/split_c4.txt 2>&1; cat $O/split_c4.txt | cut -c1-260
echo "t = $SECONDS s"
WORKLOAD=c5-terrain-10m SPP=128 SPLITS="0 16 32 64 0" REPS=2 timeout 300 python tools/split_sweep.py > $O/split_c5.txt 2>&1; cat $O/split_c5.txt | cut -c1-260
echo "t = $SECONDS s"
