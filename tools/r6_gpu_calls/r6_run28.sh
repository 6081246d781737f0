#!/bin/bash
# round 6, call 28: wf_extend's thresholds (refill 32, leaf vote 16, inner repeat 24: tuned on the host's trees) swept on the device-built trees of the headline and C4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for CFG in "pa4-cbox-path_mis 256" "c4-table-mis 128"; do set -- $CFG
  for k in 1 2; do
    echo -n "$1 shipped: "; WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
    for V in 8 12 20 24; do echo -n "$1 leaf $V: "; NORI_HIP_WF_LEAF=$V WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1; done
    for V in 24 40 48; do echo -n "$1 refill $V: "; NORI_HIP_WF_REFILL=$V WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1; done
    for V in 16 20 32 40; do echo -n "$1 inner repeat $V: "; NORI_HIP_WF_INNER_REPEAT=$V WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1; done
  done
done; } > gpurun_out/r6_35_thresholds_device_tree.txt 2>&1
cat gpurun_out/r6_35_thresholds_device_tree.txt
