#!/bin/bash
# round 6, call 5: the frame-sized RCCL self-check through a one-rank group; the projection of the 1 / 2 / 4 / 8-GPU lines from one GPU;
# every BASELINE configuration with the device builder next to the host builder (review item 3: where the device tree stands today)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_05
timeout 900 python -m pytest tests/test_gpu_wavefront.py -x -q -k "group or rccl or distributed" > ${O}_pytest_group.txt 2>&1; tail -3 ${O}_pytest_group.txt
timeout 1500 python tools/project_scaling.py gpurun_out/r6_05_projected_scaling.json > ${O}_projected_scaling.txt 2>&1; grep -v "amdgpu.ids" ${O}_projected_scaling.txt | tail -20
for WL in pa4-cbox-path_mis c4-table-mis c5-terrain-10m c2-ao-icosphere; do for B in host ploc lbvh; do
  SPP=""; [ $WL = c4-table-mis ] && SPP="--spp 128"; [ $WL = c5-terrain-10m ] && SPP="--spp 128"
  echo -n "$WL $B: "; timeout 900 python bench.py --workload $WL --builder $B $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s', d['ms_per_step'],'ms | trace',d['pass']['trace_ms'],'shade',d['pass']['shade_ms'],'| build',d['accel']['build_ms'],'ms sah',d['accel']['sah_cost'],'depth',d['accel']['max_depth'],'nodes',d['accel']['n_nodes'], '| node tests', d['roofline']['node_tests'], 'tri tests', d['roofline']['tri_tests'])"
done; done > ${O}_builders.txt 2>&1
cat ${O}_builders.txt
