#!/bin/bash
# round 6, call 10: regeneration with the new paths of a pass traced by their own launch (two pure wf_extend kernels instead of one mixed): tests, pool sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_12
timeout 1500 python -m pytest tests/test_gpu_wavefront.py -x -q > ${O}_pytest.txt 2>&1; tail -3 ${O}_pytest.txt
{
echo "== headline, pool 2^29 (one pass starts every sample)"
HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
echo "== headline, same schedule, every later pass as one that may hold new paths (both launches, kMixed wf_shade)"
NORI_HIP_WF_FORCE_MIXED=1 HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
for P in 134217728 67108864 33554432 16777216; do
echo "== headline, pool $P"
NORI_HIP_WF_POOL=$P HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
done
echo "== share of eight (tile_mod 8), pool = batch"
TILE_MOD=8 HASH=1 TIMEK=1 REPS=5 timeout 300 python tools/wf_probe.py
for P in 16777216 8388608; do
echo "== share of eight, pool $P"
NORI_HIP_WF_POOL=$P TILE_MOD=8 HASH=1 TIMEK=1 REPS=5 timeout 300 python tools/wf_probe.py
done
echo "== C4 at 128 spp (2^29 samples), one batch on a pool of 2^29"
WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
for P in 268435456 134217728 67108864; do
echo "== C4 at 128 spp, pool $P"
NORI_HIP_WF_POOL=$P WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
done
echo "== C5 at 128 spp (2^27 samples), pool = batch"
WORKLOAD=c5-terrain-10m SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
for P in 67108864 33554432; do
echo "== C5 at 128 spp, pool $P"
NORI_HIP_WF_POOL=$P WORKLOAD=c5-terrain-10m SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
done
} 2>&1 | grep -v "amdgpu.ids\|n_triangles" > ${O}_pool_sweep_two_launches.txt
cat ${O}_pool_sweep_two_launches.txt
