#!/bin/bash
# round 6, call 17: the scale of the splitting (cuts per multiple of the scene's typical priority) on every configuration
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_19
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| builder',d['accel']['builder'],'build',d['accel']['build_ms'],'ms depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
{
for k in 1 2; do
for WL in c4-table-mis pa4-cbox-path_mis c2-ao-icosphere c5-terrain-10m; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"; [ $WL = c5-terrain-10m ] && SPP="--spp 128"
  echo -n "$WL host: "; timeout 900 python bench.py --workload $WL --builder host $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  for SC in 4 3 2 1.5; do
    echo -n "$WL ploc, scale $SC: "; NORI_HIP_SPLIT_SCALE=$SC timeout 900 python bench.py --workload $WL --builder ploc $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
done; done
} > ${O}_split_scale.txt 2>&1
cat ${O}_split_scale.txt
