# round 4, GPU call 34: leaf threshold 8 on wide trees -- GPU suite, then the five profile sets again (r4_14)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_34; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
echo "t = $SECONDS s"
timeout 300 bash tools/profile_round.sh r4_14 pa4-cbox-path_mis > gpurun_out/prof_r4_14.log 2>&1; tail -1 gpurun_out/prof_r4_14.log | cut -c1-200
timeout 300 bash tools/profile_round.sh r4_14_c5 c5-terrain-10m > gpurun_out/prof_r4_14_c5.log 2>&1; tail -1 gpurun_out/prof_r4_14_c5.log | cut -c1-200
echo "t = $SECONDS s"
if [ $SECONDS -lt 240 ]; then timeout 120 bash tools/profile_round.sh r4_14_c2 c2-ao-icosphere > gpurun_out/prof_r4_14_c2.log 2>&1; tail -1 gpurun_out/prof_r4_14_c2.log | cut -c1-200; fi
if [ $SECONDS -lt 270 ]; then timeout 100 bash tools/profile_round.sh r4_14_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r4_14_c1.log 2>&1; tail -1 gpurun_out/prof_r4_14_c1.log | cut -c1-200; fi
if [ $SECONDS -lt 280 ]; then timeout 200 bash tools/profile_round.sh r4_14_c4 c4-table-mis lite > gpurun_out/prof_r4_14_c4.log 2>&1; tail -1 gpurun_out/prof_r4_14_c4.log | cut -c1-200; fi
echo "t = $SECONDS s"
