#!/bin/bash
# round 6, call 21: every shipped scene through the host's and the device's builder (counts, wf_extend time, build time)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/builders_on_goldens.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_25_builders_on_goldens.txt
cat gpurun_out/r6_25_builders_on_goldens.txt
