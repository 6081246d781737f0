#!/bin/bash
# round 6, call 20: why the device tree's node tests cost more each (Cornell box: 5.5 % fewer of them, 1 % more time): wave-level census of wf_extend
# (counting build: trips of the loop, lanes per node step / triangle step) with the host tree and the device tree, headline and C4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for CFG in "pa4-cbox-path_mis 256" "c4-table-mis 128"; do set -- $CFG
  for B in 0 2; do
    echo "== $1, builder $B (0 host, 2 auto = device)"
    NORI_HIP_CENSUS=1 COUNT=1 REPS=1 BUILDER=$B WORKLOAD=$1 SPP=$2 timeout 600 python tools/wf_probe.py 2>&1 | grep -v "amdgpu.ids\|path slots"
  done
done
} > gpurun_out/r6_23_census_host_vs_device_tree.txt 2>&1
cat gpurun_out/r6_23_census_host_vs_device_tree.txt
