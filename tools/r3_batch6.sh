cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b6; mkdir -p $O
P="env REPS=3 TIMEK=1 timeout 100 python tools/wf_probe.py"
for TM in 8 4 1; do for F in 131072 262144 524288 1048576 2097152 4194304; do echo -n "tile_mod $TM finish $F: "; TILE_MOD=$TM NORI_HIP_WF_FINISH_PATHS=$F $P 2>&1 | tail -1; done; done
for S in 1 2 4 6; do echo -n "tile_mod 8 sync $S: "; TILE_MOD=8 NORI_HIP_WF_SYNC_EVERY=$S $P 2>&1 | tail -1; done
echo -n "tile_mod 8 nofinish: "; TILE_MOD=8 NORI_HIP_WF_FINISH=0 $P 2>&1 | tail -1
bash tools/pmc_probe.sh default r3b6 fetch write tcc > $O/pmc.txt 2>&1; tail -40 $O/pmc.txt
