# Diagnostic counter passes of one library variant on the headline workload (run on the GPU box):
#   bash tools/pmc_probe.sh <variant> <tag> [pass names...]     -> gpurun_out/pmc_<tag>/ + summary on stdout
# Each pass is its own rocprofv3 run with --kernel-trace only; one render pass each (tools/wf_probe.py, REPS=1).
set -u
VAR=$1; TAG=$2; shift 2
PASSES=${@:-"ta tcp lds issue ea"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp REPS=1
[ "$VAR" != "default" ] && export NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$VAR.so
run_pass() {
  local NAME=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$NAME -o c -- python tools/wf_probe.py > $OUT/${NAME}.log 2>&1
  find /tmp/pmc_$NAME -name '*counter_collection.csv' -exec cp {} $OUT/${TAG}_${NAME}_counter_collection.csv \;
  rm -rf /tmp/pmc_$NAME
}
for P in $PASSES; do case $P in
  ta)    run_pass ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE ;;
  tcp)   run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum ;;
  lds)   run_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS ;;
  issue) run_pass issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU ;;
  ea)    run_pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ;;
  fetch) run_pass fetch FETCH_SIZE ;;
  write) run_pass write WRITE_SIZE ;;
  tcc)   run_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE ;;
esac; done
python tools/pmc_summary.py $OUT wf_ > $OUT/${TAG}_summary.txt 2>&1
cat $OUT/${TAG}_summary.txt
