#!/bin/bash
# round 6, call 24: the fuzzers on the final build, longer (intersect: hits against the scan, all builders, the device builder's knobs drawn; engines: both engines + pool / batch drawn)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "== python tests/fuzz_intersect.py --seconds 600 --seed 9000"; timeout 900 python tests/fuzz_intersect.py --seconds 600 --seed 9000 2>&1 | tail -3
  echo "== python tests/fuzz_engines.py --seconds 420 --seed 90000"; timeout 700 python tests/fuzz_engines.py --seconds 420 --seed 90000 2>&1 | tail -3; } > gpurun_out/r6_29_fuzz_long.txt 2>&1
cat gpurun_out/r6_29_fuzz_long.txt
