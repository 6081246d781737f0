# round 4, GPU call 13: final build -- full GPU suite, fuzzers, profile set of all five configurations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_13; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 200 python tests/fuzz_intersect.py --seconds 90 --seed 5000 > $O/fuzz.txt 2>&1
timeout 300 python tests/fuzz_engines.py --seconds 120 --seed 900 --oracle >> $O/fuzz.txt 2>&1
cat $O/fuzz.txt | tail -4
timeout 700 bash tools/profile_round.sh r4_08 pa4-cbox-path_mis > gpurun_out/prof_r4_08.log 2>&1; tail -1 gpurun_out/prof_r4_08.log | cut -c1-200
timeout 500 bash tools/profile_round.sh r4_08_c5 c5-terrain-10m lite > gpurun_out/prof_r4_08_c5.log 2>&1; tail -1 gpurun_out/prof_r4_08_c5.log | cut -c1-200
timeout 500 bash tools/profile_round.sh r4_08_c4 c4-table-mis lite > gpurun_out/prof_r4_08_c4.log 2>&1; tail -1 gpurun_out/prof_r4_08_c4.log | cut -c1-200
timeout 300 bash tools/profile_round.sh r4_08_c2 c2-ao-icosphere lite > gpurun_out/prof_r4_08_c2.log 2>&1; tail -1 gpurun_out/prof_r4_08_c2.log | cut -c1-200
timeout 300 bash tools/profile_round.sh r4_08_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r4_08_c1.log 2>&1; tail -1 gpurun_out/prof_r4_08_c1.log | cut -c1-200
