# Collects the rocprofv3 summaries bench.py's roofline numbers are checked against.
# usage (on the GPU box): bash tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ENGINE in wavefront megakernel; do
  B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --engine $ENGINE"
  timeout 200 $B 2>/dev/null | tail -1 > $OUT/${TAG}_${ENGINE}_bench.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$ENGINE -o s -- $B > /dev/null 2>&1
  cp $OUT/tmp_$ENGINE/*kernel_stats.csv $OUT/${TAG}_${ENGINE}_kernel_stats.csv 2>/dev/null || find $OUT/tmp_$ENGINE -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_${ENGINE}_kernel_stats.csv \;
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/tmp_${ENGINE}_$C -o c -- $B > /dev/null 2>&1
    find $OUT/tmp_${ENGINE}_$C -name '*counter_collection.csv' -exec cp {} $OUT/${TAG}_${ENGINE}_pmc_$C.csv \;
  done
  rm -rf $OUT/tmp_*
done
ls -la $OUT
