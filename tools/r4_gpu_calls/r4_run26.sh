# round 4, GPU call 28: the build with the block-row shares -- full GPU suite, the table scene's profile set, a minute of fuzzing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_28; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 420 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
echo "t = $SECONDS s"
if [ $SECONDS -lt 330 ]; then timeout 200 bash tools/profile_round.sh r4_13_c4 c4-table-mis lite > gpurun_out/prof_r4_13_c4.log 2>&1; tail -1 gpurun_out/prof_r4_13_c4.log | cut -c1-200; fi
echo "t = $SECONDS s"
if [ $SECONDS -lt 450 ]; then timeout 100 python tests/fuzz_engines.py --seconds 60 --seed 4100 --oracle > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt; fi
echo "t = $SECONDS s"
