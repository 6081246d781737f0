#!/bin/bash
# round 6, call 26: the tail-overlap test of the GPU suite repeated (it did not return within the suite's limit on one box of call 25): durations, outcome
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in $(seq 1 14); do
  S=$(date +%s.%N)
  timeout 240 python -m pytest tests/test_gpu_wavefront.py -x -q -k "tail_of_a_batch_beside or cus_split_between" > /tmp/t.log 2>&1; RC=$?
  E=$(date +%s.%N)
  echo "run $k: rc $RC $(python -c "print(round($E-$S,1))") s  $(tail -1 /tmp/t.log)"
done > gpurun_out/r6_32_tail_test_repeated.txt 2>&1
cat gpurun_out/r6_32_tail_test_repeated.txt
