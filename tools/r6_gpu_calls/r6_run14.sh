#!/bin/bash
# round 6, call 14: (a) what the device builder's phases cost on the 10 M-triangle terrain and which mix of sweeps / re-insertion iterations
# fits 150 ms; (b) the Cornell box's host tree priced as if it were deeper than 16 (hybrid stack + smaller LDS image): what depth alone costs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_16
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| build',d['accel']['build_ms'],'ms depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
{
echo "== phases (NORI_HIP_BUILD_TIMING: a synchronisation per lap), 2 sweeps + 8 iterations + 1 sweep"
NORI_HIP_BUILD_TIMING=1 timeout 900 python bench.py --workload c5-terrain-10m --builder ploc --spp 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "lbvh\]"
for CFG in "2 8 1 1" "2 4 1 0" "2 3 1 0" "2 2 1 0" "1 4 1 0" "1 6 1 0" "1 3 1 1" "0 6 1 1" "2 8 8 0" "2 8 4 0" "1 16 8 0" "2 2 1 1"; do set -- $CFG
  echo -n "c5 sweeps $1 iterations $2 stride $3 sweeps after $4: "
  NORI_HIP_TREELET_SWEEPS=$1 NORI_HIP_REINSERT_ITERS=$2 NORI_HIP_REINSERT_STRIDE=$3 NORI_HIP_REINSERT_SWEEPS_AFTER=$4 timeout 900 python bench.py --workload c5-terrain-10m --builder ploc --spp 128 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
done
} > ${O}_c5_build_matrix.txt 2>&1
cat ${O}_c5_build_matrix.txt
{
for k in 1 2; do
echo -n "cbox host tree: "; timeout 900 python bench.py --builder host --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
echo -n "cbox host tree, priced 8 levels deeper: "; NORI_HIP_LAB_DEPTH_ADD=8 timeout 900 python bench.py --builder host --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
echo -n "cbox ploc + 16 iterations: "; NORI_HIP_REINSERT_ITERS=16 timeout 900 python bench.py --builder ploc --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
done
} > ${O}_cbox_depth_price.txt 2>&1
cat ${O}_cbox_depth_price.txt
