#!/bin/bash
# round 6, call 9: the fuzzers on the final build (engines: pool and batch drawn independently -- regeneration included; intersect: hits against the scan)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "== python tests/fuzz_intersect.py --seconds 240 --seed 600"; timeout 600 python tests/fuzz_intersect.py --seconds 240 --seed 600 2>&1 | tail -6
  echo "== python tests/fuzz_engines.py --seconds 420 --seed 6000"; timeout 900 python tests/fuzz_engines.py --seconds 420 --seed 6000 2>&1 | tail -6; } > gpurun_out/r6_11_fuzz_final_build.txt 2>&1
cat gpurun_out/r6_11_fuzz_final_build.txt
