# round 5, GPU call 6: the tail of a batch on a few CUs / in its persistent form (tools/tail_probe.py), pa5 table 2048^2 at 128 spp = one batch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_06; mkdir -p $O
WORKLOAD=c4-table-mis SPP=128 timeout 500 python tools/tail_probe.py > $O/tail_probe_c4.txt 2>&1; cat $O/tail_probe_c4.txt
echo "t = $SECONDS s"
