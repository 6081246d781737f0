#!/bin/bash
# round 6, call 23: the hybrid stack of wf_extend with 15 entries in LDS (the image of hot records keeps 359 records) against 16 (230 records),
# on the device-built trees of the headline, C4, C2 -- variants alternated on one box; frames must be the same
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/nori_amd/lib
{ for CFG in "pa4-cbox-path_mis 256" "c4-table-mis 128" "c2-ao-icosphere 64"; do set -- $CFG
  for k in 1 2 3; do for V in hyb16 hyb15; do
    echo -n "$1 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_$V.so WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done; } > gpurun_out/r6_27_hybrid_stack_ab.txt 2>&1
cat gpurun_out/r6_27_hybrid_stack_ab.txt
