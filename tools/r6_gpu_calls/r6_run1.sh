#!/bin/bash
# round 6, call 1: regeneration -- correctness, then what the pool size is worth (headline, a 1/8 share, C4 at 128 spp), and the
# counters this device offers for VALU busy (review item 2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_01
timeout 900 python -m pytest tests/test_gpu_wavefront.py -x -q -k "regeneration or batching or tail_of_a_batch or out_of_memory or contexts_own" > ${O}_pytest_regen.txt 2>&1; tail -5 ${O}_pytest_regen.txt
{
echo "== headline, pool 2^29 (stored-only kernels after the first pass: round 5's schedule)"
HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
echo "== headline, same schedule, the kMixed kernels on every pass"
NORI_HIP_WF_FORCE_MIXED=1 HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
for P in 134217728 67108864 33554432 16777216 8388608; do
echo "== headline, pool $P"
NORI_HIP_WF_POOL=$P HASH=1 TIMEK=1 REPS=4 timeout 300 python tools/wf_probe.py
done
echo "== share of eight (tile_mod 8), pool = batch"
TILE_MOD=8 HASH=1 TIMEK=1 REPS=5 timeout 300 python tools/wf_probe.py
for P in 16777216 8388608 4194304 2097152; do
echo "== share of eight, pool $P"
NORI_HIP_WF_POOL=$P TILE_MOD=8 HASH=1 TIMEK=1 REPS=5 timeout 300 python tools/wf_probe.py
done
} > ${O}_pool_sweep.txt 2>&1
tail -30 ${O}_pool_sweep.txt
{
echo "== C4 at 128 spp (2^29 samples), one batch on a pool of 2^29"
WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
for P in 268435456 134217728 67108864 33554432; do
echo "== C4 at 128 spp, pool $P"
NORI_HIP_WF_POOL=$P WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py
done
} > ${O}_pool_sweep_c4.txt 2>&1
tail -12 ${O}_pool_sweep_c4.txt
cd /tmp; rocprofv3 -L > $GRAFT_REPO_ROOT/${O}_rocprof_counters_list.txt 2>&1; grep -c . $GRAFT_REPO_ROOT/${O}_rocprof_counters_list.txt
grep -i -o "SQ_[A-Z_0-9]*VALU[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*" $GRAFT_REPO_ROOT/${O}_rocprof_counters_list.txt | sort -u
