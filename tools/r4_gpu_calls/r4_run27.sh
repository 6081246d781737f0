# round 4, GPU call 30: the terrain's and the AO scene's profile sets WITH the SQ counters (VALU busy next to the HBM share)
cd $GRAFT_REPO_ROOT
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 330 bash tools/profile_round.sh r4_13_c5 c5-terrain-10m > gpurun_out/prof_r4_13_c5.log 2>&1; tail -1 gpurun_out/prof_r4_13_c5.log | cut -c1-200
echo "t = $SECONDS s"
if [ $SECONDS -lt 300 ]; then timeout 150 bash tools/profile_round.sh r4_13_c2 c2-ao-icosphere > gpurun_out/prof_r4_13_c2.log 2>&1; tail -1 gpurun_out/prof_r4_13_c2.log | cut -c1-200; fi
echo "t = $SECONDS s"
