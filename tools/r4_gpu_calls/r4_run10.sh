# round 4, GPU call 10: the profile set of the final build (tools/profile_round.sh): kernel stats, counters, bench lines
cd $GRAFT_REPO_ROOT
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 700 bash tools/profile_round.sh r4_06 pa4-cbox-path_mis > gpurun_out/prof_r4_06.log 2>&1; tail -3 gpurun_out/prof_r4_06.log | cut -c1-400
timeout 500 bash tools/profile_round.sh r4_06_c5 c5-terrain-10m lite > gpurun_out/prof_r4_06_c5.log 2>&1; tail -1 gpurun_out/prof_r4_06_c5.log | cut -c1-300
timeout 500 bash tools/profile_round.sh r4_06_c4 c4-table-mis lite > gpurun_out/prof_r4_06_c4.log 2>&1; tail -1 gpurun_out/prof_r4_06_c4.log | cut -c1-300
timeout 300 bash tools/profile_round.sh r4_06_c2 c2-ao-icosphere lite > gpurun_out/prof_r4_06_c2.log 2>&1; tail -1 gpurun_out/prof_r4_06_c2.log | cut -c1-300
timeout 300 bash tools/profile_round.sh r4_06_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r4_06_c1.log 2>&1; tail -1 gpurun_out/prof_r4_06_c1.log | cut -c1-300
