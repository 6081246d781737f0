cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "builder or fuzz or wide or lds_image" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
P="env REPS=2 TIMEK=1 timeout 300 python tools/wf_probe.py"
for B in 0 1 2; do echo -n "c5 64spp builder $B: "; WORKLOAD=c5-terrain-10m SPP=64 BUILDER=$B $P 2>&1 | tail -3 | tr '\n' ' '; echo; done
for SW in 0 1 3; do echo -n "c5 64spp PLOC sweeps $SW: "; NORI_HIP_TREELET_SWEEPS=$SW WORKLOAD=c5-terrain-10m SPP=64 BUILDER=2 $P 2>&1 | tail -3 | tr '\n' ' '; echo; done
for B in 0 2; do echo -n "cbox builder $B: "; BUILDER=$B $P 2>&1 | tail -3 | tr '\n' ' '; echo; done
for B in 0 2; do echo -n "c2 builder $B: "; WORKLOAD=c2-ao-icosphere BUILDER=$B $P 2>&1 | tail -3 | tr '\n' ' '; echo; done
