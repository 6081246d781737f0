#!/bin/bash
# round 6, call 2: the GPU suite on the regeneration build; what binds wf_extend, by perturbation and by counters (review item 2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_02
timeout 2400 python -m pytest tests -m gpu -x -q > ${O}_pytest_gpu.txt 2>&1; tail -4 ${O}_pytest_gpu.txt
L=$GRAFT_REPO_ROOT/nori_amd/lib
# ---- timing: the four lab variants alternated, per configuration (trace = wf_extend alone, HIP events)
for CFG in "pa4-cbox-path_mis 256" "c5-terrain-10m 128" "c4-table-mis 64"; do set -- $CFG
  for k in 1 2; do for V in base valu load idle; do
    echo -n "$1 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_lab_$V.so WORKLOAD=$1 SPP=$2 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done > ${O}_sens_times.txt 2>&1
echo -n "c5-terrain-10m base_no_lds_image: " >> ${O}_sens_times.txt; NORI_HIP_NO_TOP_IMAGE=1 NORI_HIP_LIBRARY=$L/libnori_hip_lab_base.so WORKLOAD=c5-terrain-10m SPP=128 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1 >> ${O}_sens_times.txt
cat ${O}_sens_times.txt
# ---- counters: what each variant added, measured (instructions by kind), one render pass each
cd /tmp
for CFG in "pa4-cbox-path_mis 256 hl" "c5-terrain-10m 128 c5"; do set -- $CFG
  for V in base valu load idle; do
    NORI_HIP_LIBRARY=$L/libnori_hip_lab_$V.so WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pc_$3_$V -o c -- python $GRAFT_REPO_ROOT/tools/wf_probe.py > /tmp/pc.log 2>&1
    find /tmp/pc_$3_$V -name '*counter_collection.csv' -exec cp {} ${O}_sens_$3_${V}_counter_collection.csv \; ; rm -rf /tmp/pc_$3_$V
  done
  # the dynamic instruction mix of the product kernels: VALU instructions by class
  WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d /tmp/pm_$3 -o c -- python $GRAFT_REPO_ROOT/tools/wf_probe.py > /tmp/pm.log 2>&1
  find /tmp/pm_$3 -name '*counter_collection.csv' -exec cp {} ${O}_mix_$3_counter_collection.csv \; ; rm -rf /tmp/pm_$3
  WORKLOAD=$1 SPP=$2 REPS=1 timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pg_$3 -o c -- python $GRAFT_REPO_ROOT/tools/wf_probe.py > /tmp/pg.log 2>&1
  find /tmp/pg_$3 -name '*counter_collection.csv' -exec cp {} ${O}_elapsed_$3_counter_collection.csv \; ; rm -rf /tmp/pg_$3
done
# ---- the class counters on kernels of ONE instruction each: which class an opcode is counted in, and its cycles
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/ubench_valu.hip -o /tmp/ubench_valu > /tmp/ub_build.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d /tmp/ub1 -o c -- /tmp/ubench_valu > ${O}_ubench_stdout.txt 2>&1
find /tmp/ub1 -name '*counter_collection.csv' -exec cp {} ${O}_ubench_mix_counter_collection.csv \;
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/ub2 -o c -- /tmp/ubench_valu > /dev/null 2>&1
find /tmp/ub2 -name '*counter_collection.csv' -exec cp {} ${O}_ubench_elapsed_counter_collection.csv \;
ls -la $GRAFT_REPO_ROOT/gpurun_out | grep r6_02 | head -40
