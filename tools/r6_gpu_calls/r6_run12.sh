#!/bin/bash
# round 6, call 12: per-launch durations of one device's share of eight (kernel trace) next to the paths each pass held (census readback every pass)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
NORI_HIP_CENSUS=1 NORI_HIP_WF_SYNC_EVERY=1 TILE_MOD=8 REPS=1 timeout 300 python tools/wf_probe.py 2>&1 | grep "wavefront\]" > gpurun_out/r6_14_share_census.txt
TILE_MOD=8 REPS=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python tools/wf_probe.py > /tmp/kt.log 2>&1
find /tmp/kt -name '*kernel_trace.csv' -exec cp {} gpurun_out/r6_14_share_kernel_trace.csv \;
cat gpurun_out/r6_14_share_census.txt | head -20; wc -l gpurun_out/r6_14_share_kernel_trace.csv
