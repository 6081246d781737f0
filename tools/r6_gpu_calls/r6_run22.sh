#!/bin/bash
# round 6, call 22: the leaf collapse's price of a pair test (NORI_HIP_LBVH_PAIR_COST, default 1.5) on the device trees of the headline and C4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
{
for k in 1 2; do
for WL in pa4-cbox-path_mis c4-table-mis; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"
  for PC in 1.0 1.5 2.0 3.0; do
    echo -n "$WL pair cost $PC: "; NORI_HIP_LBVH_PAIR_COST=$PC timeout 900 python bench.py --workload $WL $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
done; done
} > gpurun_out/r6_26_pair_cost.txt 2>&1
cat gpurun_out/r6_26_pair_cost.txt
