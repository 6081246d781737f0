# One round's profile set for bench.py's roofline (run on the GPU box):
#   bash tools/profile_round.sh <tag> [workload] [lite]     -> gpurun_out/prof_<tag>/, then copy the summaries into profiles/
#   1. rocprofv3 --kernel-trace --stats of the bench command                          <tag>_kernel_stats.csv
#   2. counter passes (each its own rocprofv3 run, --kernel-trace only), ONE render pass each (tools/wf_probe.py, REPS=1):
#        sq_issue (skipped with `lite`: the SQ counters serialise the kernels, ~2 min)   SQ_INSTS_VALU, lanes, wait / issue shares
#        fetch    FETCH_SIZE + GRBM_GUI_ACTIVE (elapsed cycles)                          <tag>_<pass>_counter_collection.csv
#        write    WRITE_SIZE + TCC_HIT_sum + TCC_MISS_sum
#   3. tools/summarize_profile.py -> <tag>_counters.json  (what bench.py's `traffic` / `valu_busy_frac` read; matched by device-source hash)
#   4. the bench line itself, with those counters in place                              <tag>_bench.json
set -u
TAG=${1:-r3}
WL=${2:-pa4-cbox-path_mis}
LITE=${3:-}
ENG=${4:-wavefront}      # the engine the library picks for this workload (c1: megakernel)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --workload $WL"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- $B --no-cpu-baseline > $OUT/stats.log 2>&1
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
rm -rf /tmp/prof_stats
export REPS=1 WORKLOAD=$WL ENGINE=$ENG
run_pass() {
  local NAME=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/prof_$NAME -o c -- python tools/wf_probe.py > $OUT/${NAME}.log 2>&1
  find /tmp/prof_$NAME -name '*counter_collection.csv' -exec cp {} $OUT/${TAG}_${NAME}_counter_collection.csv \;
  rm -rf /tmp/prof_$NAME
}
[ "$LITE" != "lite" ] && run_pass sq_issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python tools/summarize_profile.py $OUT $TAG $WL $ENG > $OUT/${TAG}_summary.txt 2>&1
# the bench line last, with the counters of THIS build in place (bench.py reads profiles/*_counters.json matched by build hash)
cp $OUT/${TAG}_counters.json profiles/
timeout 900 $B 2>$OUT/bench.err | tail -1 > $OUT/${TAG}_bench.json
cat $OUT/${TAG}_summary.txt; cat $OUT/${TAG}_bench.json
