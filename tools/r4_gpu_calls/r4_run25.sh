# round 4, GPU call 26: film_order = reference shared out by block rows (tests), then the profile set again for the build that has it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_26; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 400 python -m pytest tests/test_gpu_wavefront.py -m gpu -q --durations=5 -k "reference_order or device_group or native_library" > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log; tail -12 $O/pytest_new.log
echo "t = $SECONDS s"
timeout 420 bash tools/profile_round.sh r4_13 pa4-cbox-path_mis > gpurun_out/prof_r4_13.log 2>&1; tail -1 gpurun_out/prof_r4_13.log | cut -c1-300
echo "t = $SECONDS s"
if [ $SECONDS -lt 430 ]; then timeout 200 bash tools/profile_round.sh r4_13_c5 c5-terrain-10m lite > gpurun_out/prof_r4_13_c5.log 2>&1; tail -1 gpurun_out/prof_r4_13_c5.log | cut -c1-200; fi
echo "t = $SECONDS s"
if [ $SECONDS -lt 560 ]; then timeout 120 bash tools/profile_round.sh r4_13_c2 c2-ao-icosphere lite > gpurun_out/prof_r4_13_c2.log 2>&1; tail -1 gpurun_out/prof_r4_13_c2.log | cut -c1-200; fi
echo "t = $SECONDS s"
if [ $SECONDS -lt 620 ]; then timeout 100 bash tools/profile_round.sh r4_13_c1 c1-bunny-normals lite megakernel > gpurun_out/prof_r4_13_c1.log 2>&1; tail -1 gpurun_out/prof_r4_13_c1.log | cut -c1-200; fi
echo "t = $SECONDS s"
