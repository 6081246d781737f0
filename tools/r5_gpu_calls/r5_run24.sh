# round 5, GPU call 24: the C4 profile set once more WITH the SQ counters (VALU busy / lanes per instruction for the table scene's kernels)
cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/profile_round.sh r5_15_c4 c4-table-mis > gpurun_out/prof_r5_15_c4.log 2>&1; tail -1 gpurun_out/prof_r5_15_c4.log | cut -c1-300
grep -E "dominant|wf_finish|wf_shade<6, false" gpurun_out/prof_r5_15_c4.log | cut -c1-420
echo "t = $SECONDS s"
