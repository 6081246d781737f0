#!/bin/bash
# round 6, call 16: triangle splitting in front of the device builder (references): the builders' GPU tests; every configuration host / ploc; C4 with the knobs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_18
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "builder or lbvh or ploc or wide or terrain or auto" > ${O}_pytest_builders.txt 2>&1; tail -3 ${O}_pytest_builders.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| builder',d['accel']['builder'],'build',d['accel']['build_ms'],'ms depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
{
for k in 1 2; do
for WL in c4-table-mis pa4-cbox-path_mis c5-terrain-10m c2-ao-icosphere; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"; [ $WL = c5-terrain-10m ] && SPP="--spp 128"
  for B in host ploc; do
    echo -n "$WL $B: "; timeout 900 python bench.py --workload $WL --builder $B $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
done; done
echo "== C4, knobs of the splitting"
for CFG in "0 15 4 4" "0.3 15 2 4" "0.3 15 8 4" "0.3 15 4 0" "0.3 15 4 16" "0.3 63 4 4" "0.3 3 4 4" "1.0 15 2 4"; do set -- $CFG
  echo -n "c4 budget $1 cap $2 scale $3 inside $4: "; NORI_HIP_SPLIT_BUDGET=$1 NORI_HIP_SPLIT_CAP=$2 NORI_HIP_SPLIT_SCALE=$3 NORI_HIP_SPLIT_INSIDE=$4 timeout 900 python bench.py --workload c4-table-mis --builder ploc --spp 128 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
done
echo "== phases, C4 and C5"
NORI_HIP_BUILD_TIMING=1 timeout 900 python bench.py --workload c4-table-mis --builder ploc --spp 16 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "lbvh\]" | grep -v "re-insertion"
NORI_HIP_BUILD_TIMING=1 timeout 900 python bench.py --workload c5-terrain-10m --builder ploc --spp 16 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "lbvh\]" | grep -v "re-insertion"
} > ${O}_builders_split.txt 2>&1
cat ${O}_builders_split.txt
