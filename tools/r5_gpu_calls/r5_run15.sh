# round 5, GPU call 15: the profile set of the final build (tools/profile_round.sh): kernel stats, counters, bench lines for the five BASELINE configurations
cd $GRAFT_REPO_ROOT
timeout 800 bash tools/profile_round.sh r5_15 pa4-cbox-path_mis > gpurun_out/prof_r5_15.log 2>&1; tail -1 gpurun_out/prof_r5_15.log | cut -c1-300
echo "t = $SECONDS s"
timeout 700 bash tools/profile_round.sh r5_15_c4 c4-table-mis lite > gpurun_out/prof_r5_15_c4.log 2>&1; tail -1 gpurun_out/prof_r5_15_c4.log | cut -c1-300
echo "t = $SECONDS s"
timeout 700 bash tools/profile_round.sh r5_15_c5 c5-terrain-10m > gpurun_out/prof_r5_15_c5.log 2>&1; tail -1 gpurun_out/prof_r5_15_c5.log | cut -c1-300
echo "t = $SECONDS s"
timeout 400 bash tools/profile_round.sh r5_15_c2 c2-ao-icosphere > gpurun_out/prof_r5_15_c2.log 2>&1; tail -1 gpurun_out/prof_r5_15_c2.log | cut -c1-300
echo "t = $SECONDS s"
timeout 300 bash tools/profile_round.sh r5_15_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r5_15_c1.log 2>&1; tail -1 gpurun_out/prof_r5_15_c1.log | cut -c1-300
echo "t = $SECONDS s"
