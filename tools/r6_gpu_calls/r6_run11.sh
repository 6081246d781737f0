#!/bin/bash
# round 6, call 11: the ray counters summed per workgroup (wgsum) against per wave (wavesum: 8192 waves x 2 - 3 atomics on one cache line at the end of every launch)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/nori_amd/lib
{ for CFG in "pa4-cbox-path_mis 256 1" "pa4-cbox-path_mis 256 8" "c4-table-mis 128 1" "c4-table-mis 128 8" "c5-terrain-10m 128 1" "c2-ao-icosphere 64 1"; do set -- $CFG
  for k in 1 2 3; do for V in wavesum wgsum; do
    echo -n "$1 tile_mod $3 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_$V.so WORKLOAD=$1 SPP=$2 TILE_MOD=$3 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done; } > gpurun_out/r6_13_counters_per_workgroup_ab.txt 2>&1
cat gpurun_out/r6_13_counters_per_workgroup_ab.txt
