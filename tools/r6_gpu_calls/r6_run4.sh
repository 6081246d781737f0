#!/bin/bash
# round 6, call 4: the stack pop requested at the start of the node step (early pop) against the loop as it was, same box, alternated
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_04
L=$GRAFT_REPO_ROOT/nori_amd/lib
for CFG in "pa4-cbox-path_mis 256" "c4-table-mis 64" "c2-ao-icosphere 64"; do set -- $CFG
  for k in 1 2 3; do for V in prevloop earlypop; do
    echo -n "$1 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_$V.so WORKLOAD=$1 SPP=$2 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done > ${O}_early_pop_ab.txt 2>&1
cat ${O}_early_pop_ab.txt
NORI_HIP_LIBRARY=$L/libnori_hip_earlypop.so timeout 1200 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_parity.py -x -q > ${O}_pytest.txt 2>&1; tail -3 ${O}_pytest.txt
