#!/bin/bash
# round 6, call 8: wf_shade compiled for 5 / 6 workgroups per SIMD (launch bounds) against 4, C4 and headline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/nori_amd/lib
for CFG in "c4-table-mis 128" "pa4-cbox-path_mis 256"; do set -- $CFG
  for k in 1 2; do for V in lab_base shade5 shade6; do
    echo -n "$1 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_$V.so WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done > gpurun_out/r6_08_shade_wgs_ab.txt 2>&1
cat gpurun_out/r6_08_shade_wgs_ab.txt
