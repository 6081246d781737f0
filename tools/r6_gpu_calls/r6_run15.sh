#!/bin/bash
# round 6, call 15: the device builder with its shipped tuning (build_tuning, lbvh.h) on every configuration next to the host tree; the builders' GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_17
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "builder or lbvh or ploc or wide or terrain or auto" > ${O}_pytest_builders.txt 2>&1; tail -3 ${O}_pytest_builders.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| builder',d['accel']['builder'],'build',d['accel']['build_ms'],'ms depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"; }
for k in 1 2; do
for WL in pa4-cbox-path_mis c4-table-mis c5-terrain-10m c2-ao-icosphere c1-bunny-normals; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"; [ $WL = c5-terrain-10m ] && SPP="--spp 128"
  for B in host ploc auto; do
    echo -n "$WL $B: "; timeout 900 python bench.py --workload $WL --builder $B $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | line
  done
done; done > ${O}_builders_shipped.txt 2>&1
cat ${O}_builders_shipped.txt
