# round 5, GPU call 2: the split test again; reordering the paths between bounces -- the ceiling (global radix sort of an index, cost reported apart); wf_shade general vs per material set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_02; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_wavefront.py -x -q > $O/pytest_wavefront.log 2>&1; echo "pytest rc $?" >> $O/pytest_wavefront.log; tail -3 $O/pytest_wavefront.log
echo "t = $SECONDS s"
CONFIGS="default off 1,3 1,5 1,7 2,5 3,4 1,5,1,1 1,5,1,3" timeout 300 python tools/sort_probe.py > $O/sort_headline.txt 2>&1; cat $O/sort_headline.txt | cut -c1-250
echo "t = $SECONDS s"
for k in 1 2; do
  echo -n "matset: "; TIMEK=1 REPS=3 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
  echo -n "any:    "; NORI_HIP_SHADE_ANY_BSDF=1 TIMEK=1 REPS=3 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done > $O/shade_matset_vs_any.txt 2>&1; cat $O/shade_matset_vs_any.txt
echo "t = $SECONDS s"
WORKLOAD=c5-terrain-10m SPP=128 CONFIGS="default off 1,5 1,7 1,9 3,6" timeout 400 python tools/sort_probe.py > $O/sort_c5.txt 2>&1; cat $O/sort_c5.txt | cut -c1-250
echo "t = $SECONDS s"
WORKLOAD=c4-table-mis SPP=128 CONFIGS="default off 1,5 1,7" timeout 300 python tools/sort_probe.py > $O/sort_c4.txt 2>&1; cat $O/sort_c4.txt | cut -c1-250
echo "t = $SECONDS s"
