#!/bin/bash
# round 6, call 27: the GPU suite repeated with a per-test limit (pytest-timeout, stacks dumped): does the stall of call 25 come back, and where
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 1 2 3 4 5; do
  S=$(date +%s)
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread -o faulthandler_timeout=200 > gpurun_out/r6_33_suite_run$k.txt 2>&1; RC=$?
  echo "suite run $k: rc $RC $(( $(date +%s) - S )) s: $(tail -1 gpurun_out/r6_33_suite_run$k.txt | cut -c1-150)"
done > gpurun_out/r6_33_suite_repeated.txt 2>&1
cat gpurun_out/r6_33_suite_repeated.txt
