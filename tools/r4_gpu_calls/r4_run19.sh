# round 4, GPU call 19: the final build -- full GPU suite, profile set of all five configurations, the terrain with the device builder
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_19; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 600 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -10 $O/pytest.log
timeout 700 bash tools/profile_round.sh r4_10 pa4-cbox-path_mis > gpurun_out/prof_r4_10.log 2>&1; tail -1 gpurun_out/prof_r4_10.log | cut -c1-200
timeout 500 bash tools/profile_round.sh r4_10_c5 c5-terrain-10m lite > gpurun_out/prof_r4_10_c5.log 2>&1; tail -1 gpurun_out/prof_r4_10_c5.log | cut -c1-200
timeout 500 bash tools/profile_round.sh r4_10_c4 c4-table-mis lite > gpurun_out/prof_r4_10_c4.log 2>&1; tail -1 gpurun_out/prof_r4_10_c4.log | cut -c1-200
timeout 300 bash tools/profile_round.sh r4_10_c2 c2-ao-icosphere lite > gpurun_out/prof_r4_10_c2.log 2>&1; tail -1 gpurun_out/prof_r4_10_c2.log | cut -c1-200
timeout 300 bash tools/profile_round.sh r4_10_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r4_10_c1.log 2>&1; tail -1 gpurun_out/prof_r4_10_c1.log | cut -c1-200
timeout 300 python bench.py --workload c5-terrain-10m --builder auto 2> $O/c5_auto.err | tail -1 > $O/r4_10_c5_device_builder_bench.json; cut -c1-200 $O/r4_10_c5_device_builder_bench.json
timeout 300 python bench.py --workload c2-ao-icosphere --builder ploc 2> $O/c2_ploc.err | tail -1 > $O/r4_10_c2_device_builder_bench.json; cut -c1-200 $O/r4_10_c2_device_builder_bench.json
