#!/bin/bash
# round 6, call 3: the counter half of call 2 again (rocprofv3 was started outside the repo: wf_probe.py did not find the package)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_02
L=$GRAFT_REPO_ROOT/nori_amd/lib
for CFG in "pa4-cbox-path_mis 256 hl" "c5-terrain-10m 128 c5"; do set -- $CFG
  for V in base valu load idle; do
    NORI_HIP_LIBRARY=$L/libnori_hip_lab_$V.so WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pc_$3_$V -o c -- python tools/wf_probe.py > /tmp/pc_$3_$V.log 2>&1
    tail -2 /tmp/pc_$3_$V.log
    find /tmp/pc_$3_$V -name '*counter_collection.csv' -exec cp {} ${O}_sens_$3_${V}_counter_collection.csv \; ; rm -rf /tmp/pc_$3_$V
  done
  WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d /tmp/pm_$3 -o c -- python tools/wf_probe.py > /tmp/pm.log 2>&1
  tail -2 /tmp/pm.log
  find /tmp/pm_$3 -name '*counter_collection.csv' -exec cp {} ${O}_mix_$3_counter_collection.csv \; ; rm -rf /tmp/pm_$3
  WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pg_$3 -o c -- python tools/wf_probe.py > /tmp/pg.log 2>&1
  find /tmp/pg_$3 -name '*counter_collection.csv' -exec cp {} ${O}_elapsed_$3_counter_collection.csv \; ; rm -rf /tmp/pg_$3
done
ls -la gpurun_out | grep r6_02
