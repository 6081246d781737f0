cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b9; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "builder or wide or lds_image" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
P="env REPS=2 TIMEK=1 timeout 120 python tools/wf_probe.py"
for SW in 0 1 2 3; do echo -n "c5 64spp PLOC sweeps $SW: "; NORI_HIP_TREELET_SWEEPS=$SW WORKLOAD=c5-terrain-10m SPP=64 BUILDER=3 $P 2>&1 | tail -3 | tr '\n' ' ' | cut -c1-420; echo; done
echo -n "c5 64spp host SAH: "; WORKLOAD=c5-terrain-10m SPP=64 BUILDER=0 $P 2>&1 | tail -2 | tr '\n' ' '; echo
for B in 0 3; do echo -n "cbox builder $B: "; BUILDER=$B $P 2>&1 | tail -3 | tr '\n' ' ' | cut -c1-420; echo; done
for B in 0 3; do echo -n "c2 builder $B: "; WORKLOAD=c2-ao-icosphere BUILDER=$B $P 2>&1 | tail -3 | tr '\n' ' ' | cut -c1-420; echo; done
