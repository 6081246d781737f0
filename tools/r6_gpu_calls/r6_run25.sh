#!/bin/bash
# round 6, call 25: the profile set of the final build (auto = the device builder: every line rendered through a tree built by HIP kernels) -- bound evidence (perturbations + counters), rocprof kernel stats + PMC passes and the
# bench line of all five configurations, the GPU suite, smoke()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 bash tools/collect_bound_evidence.sh r6_31 > gpurun_out/r6_31_evidence.log 2>&1; tail -12 gpurun_out/r6_31_evidence.log
cp gpurun_out/evidence_r6_31/r6_31_bound_evidence.json profiles/      # (bench.py reads it from profiles/, matched by device-source hash)
timeout 1500 bash tools/profile_round.sh r6_31 pa4-cbox-path_mis > gpurun_out/r6_31_profile.log 2>&1; tail -3 gpurun_out/r6_31_profile.log
timeout 900 bash tools/profile_round.sh r6_31_c5 c5-terrain-10m > gpurun_out/r6_31_c5_profile.log 2>&1; tail -1 gpurun_out/r6_31_c5_profile.log
timeout 1500 bash tools/profile_round.sh r6_31_c4 c4-table-mis lite > gpurun_out/r6_31_c4_profile.log 2>&1; tail -1 gpurun_out/r6_31_c4_profile.log
timeout 600 bash tools/profile_round.sh r6_31_c2 c2-ao-icosphere lite > gpurun_out/r6_31_c2_profile.log 2>&1; tail -1 gpurun_out/r6_31_c2_profile.log
timeout 600 bash tools/profile_round.sh r6_31_c1 c1-bunny-normals lite megakernel > gpurun_out/r6_31_c1_profile.log 2>&1; tail -1 gpurun_out/r6_31_c1_profile.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_31_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r6_31_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_31_smoke.txt 2>&1; tail -2 gpurun_out/r6_31_smoke.txt
timeout 600 python bench.py > gpurun_out/r6_31_bench_default.json 2> gpurun_out/r6_31_bench_default.err; tail -c 600 gpurun_out/r6_31_bench_default.json
{ echo "== python tests/fuzz_intersect.py --seconds 120 --seed 700"; timeout 400 python tests/fuzz_intersect.py --seconds 120 --seed 700 2>&1 | tail -3
  echo "== python tests/fuzz_engines.py --seconds 240 --seed 7000"; timeout 600 python tests/fuzz_engines.py --seconds 240 --seed 7000 2>&1 | tail -3; } > gpurun_out/r6_31_fuzz_final_build.txt 2>&1
cat gpurun_out/r6_31_fuzz_final_build.txt
