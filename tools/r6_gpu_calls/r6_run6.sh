#!/bin/bash
# round 6, call 6: wf_finish through the LDS image of the hot records (off / on), C4 and the share of eight; frames must be identical
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_06
{
for k in 1 2; do for I in 0 1; do
echo -n "c4 256 spp (2 batches), image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I WORKLOAD=c4-table-mis SPP=256 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
echo -n "c4 128 spp share of eight (tile_mod 8), image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I TILE_MOD=8 WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
echo -n "c4 128 spp one batch, image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I WORKLOAD=c4-table-mis SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
echo -n "headline share of eight, image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I TILE_MOD=8 HASH=1 TIMEK=1 REPS=4 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
echo -n "headline, image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
echo -n "c5 128 spp, image $I: "; NORI_HIP_WF_FINISH_IMAGE=$I WORKLOAD=c5-terrain-10m SPP=128 HASH=1 TIMEK=1 REPS=3 timeout 600 python tools/wf_probe.py 2>&1 | tail -1
done; done
} > ${O}_finish_image_ab.txt 2>&1
cat ${O}_finish_image_ab.txt
timeout 1500 python -m pytest tests/test_gpu_wavefront.py -x -q > ${O}_pytest.txt 2>&1; tail -3 ${O}_pytest.txt
