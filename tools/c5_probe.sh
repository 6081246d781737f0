# C5 (10 M-triangle terrain) on one GPU: rate, traversal counters and the counter passes, per builder.
# usage (GPU box): bash tools/c5_probe.sh <tag> [spp]
set -u
TAG=${1:-c5}; SPP=${2:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
export WORKLOAD=c5-terrain-10m SPP=$SPP REPS=2
for BUILDER in ${BUILDERS:-0}; do
  export BUILDER
  python tools/wf_probe.py > $OUT/probe_b$BUILDER.txt 2>&1
  NORI_HIP_CENSUS=1 COUNT=1 REPS=1 python tools/wf_probe.py > $OUT/census_b$BUILDER.txt 2>&1
  export REPS=1
  for P in "sq_issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE" "tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
    set -- $P; NAME=$1; shift
    timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/c5_$NAME -o c -- python tools/wf_probe.py > $OUT/${NAME}_b$BUILDER.log 2>&1
    find /tmp/c5_$NAME -name '*counter_collection.csv' -exec cp {} $OUT/${TAG}b${BUILDER}_${NAME}_counter_collection.csv \;
    rm -rf /tmp/c5_$NAME
  done
  python tools/summarize_profile.py $OUT ${TAG}b${BUILDER} c5-terrain-10m > $OUT/summary_b$BUILDER.txt 2>&1
  cat $OUT/probe_b$BUILDER.txt $OUT/summary_b$BUILDER.txt; tail -3 $OUT/census_b$BUILDER.txt
done
