#!/bin/bash
# round 6, call 13: parallel re-insertion in the device builder (review item 3): the builders' GPU tests, then every configuration with
# PLOC + sweeps + 0 / 8 / 16 / 32 re-insertion iterations next to the host tree (wf_extend ms, build ms, node / triangle tests)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_15
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "builder or lbvh or ploc or wide or terrain or auto" > ${O}_pytest_builders.txt 2>&1; tail -3 ${O}_pytest_builders.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| build',d['accel']['build_ms'],'ms depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
for WL in pa4-cbox-path_mis c4-table-mis c5-terrain-10m c2-ao-icosphere; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"; [ $WL = c5-terrain-10m ] && SPP="--spp 128"
  echo -n "$WL host: "; timeout 900 python bench.py --workload $WL --builder host $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  for IT in 0 8 16 32; do
    echo -n "$WL ploc reinsert $IT: "; NORI_HIP_REINSERT_ITERS=$IT timeout 900 python bench.py --workload $WL --builder ploc $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
  for ST in 4 8; do
    echo -n "$WL ploc reinsert 8 stride $ST: "; NORI_HIP_REINSERT_ITERS=8 NORI_HIP_REINSERT_STRIDE=$ST timeout 900 python bench.py --workload $WL --builder ploc $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
done > ${O}_builders_reinsert.txt 2>&1
cat ${O}_builders_reinsert.txt
