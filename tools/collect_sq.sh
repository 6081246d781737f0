# SQ / TCC / TCP counter passes on the wavefront engine's kernels (wf_extend, wf_shade) for the headline
# workload -- the evidence behind bench.py's `roofline` (bound = VALU issue vs HBM).
# usage (on the GPU box): bash tools/collect_sq.sh <tag> [probe script]   -> gpurun_out/sq_<tag>/
# Each pass is its own rocprofv3 run with --kernel-trace only (never combined with sys/hip traces).
set -u
TAG=${1:-r2}
PROBE=${2:-tools/wf_probe.py}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
export REPS=${REPS:-1}
run_pass() {   # name, counters...
  local NAME=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/sq_$NAME -o c -- python $PROBE > $OUT/${NAME}.log 2>&1
  find /tmp/sq_$NAME -name '*counter_collection.csv' -exec cp {} $OUT/${TAG}_${NAME}_counter_collection.csv \;
  rm -rf /tmp/sq_$NAME
}
run_pass sq_issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass sq_mix SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run_pass sq_misc SQ_WAVES SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_CYCLES
run_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run_pass tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
python tools/pmc_summary.py $OUT wf_ > $OUT/${TAG}_summary.txt 2>&1
cat $OUT/${TAG}_summary.txt
