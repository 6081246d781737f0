#!/bin/bash
# What binds wf_extend, measured (run on the GPU box after tools/build_lab_variants.sh):   bash tools/collect_bound_evidence.sh <tag>
#   1. times: the four lab variants alternated per configuration (HIP events around wf_extend)            <tag>_sens_times.txt
#   2. counters per variant: what it added (instructions by kind), one render pass each                    <tag>_sens_<cfg>_<variant>_counter_collection.csv
#   3. the dynamic VALU instruction mix of the kernels + elapsed cycles                                     <tag>_mix_<cfg>_*, <tag>_elapsed_<cfg>_*
#   4. the same class counters on kernels of one opcode each (tools/ubench_valu.hip): class and cycles of every opcode
#   5. tools/bound_evidence.py -> profiles/<tag>_bound_evidence.{json,txt}   (bench.py reads the json, matched by device-source hash)
set -u
TAG=${1:-r6}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence_$TAG; mkdir -p $OUT
O=$OUT/$TAG
L=$GRAFT_REPO_ROOT/nori_amd/lib
for CFG in "pa4-cbox-path_mis 256" "c5-terrain-10m 128"; do set -- $CFG
  for k in 1 2 3; do for V in base valu load idle; do
    echo -n "$1 $V: "; NORI_HIP_LIBRARY=$L/libnori_hip_lab_$V.so WORKLOAD=$1 SPP=$2 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
  done; done
done > ${O}_sens_times.txt 2>&1
echo -n "c5-terrain-10m base_no_lds_image: " >> ${O}_sens_times.txt; NORI_HIP_NO_TOP_IMAGE=1 NORI_HIP_LIBRARY=$L/libnori_hip_lab_base.so WORKLOAD=c5-terrain-10m SPP=128 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1 >> ${O}_sens_times.txt
pmc() { local NAME=$1; shift; local LOG=/tmp/pmc_$NAME.log
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$NAME -o c -- ${CMD:-python tools/wf_probe.py} > $LOG 2>&1
  find /tmp/pmc_$NAME -name '*counter_collection.csv' -exec cp {} ${O}_${NAME}_counter_collection.csv \; ; rm -rf /tmp/pmc_$NAME; }
export REPS=1
for CFG in "pa4-cbox-path_mis 256 hl" "c5-terrain-10m 128 c5"; do set -- $CFG
  export WORKLOAD=$1 SPP=$2
  for V in base valu load idle; do
    NORI_HIP_LIBRARY=$L/libnori_hip_lab_$V.so pmc sens_$3_$V SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  done
  pmc mix_$3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
  pmc elapsed_$3 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu > /tmp/ub_build.log 2>&1
CMD=/tmp/ubench_valu pmc ubench_mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
CMD=/tmp/ubench_valu pmc ubench_elapsed GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
python tools/bound_evidence.py $OUT $TAG
cp profiles/${TAG}_bound_evidence.* $OUT/
