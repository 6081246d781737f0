# round 5, GPU call 16: wf_extend in 512-thread workgroups at 6 waves per SIMD, plane coefficients kept in registers (tools/block_probe.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_16; mkdir -p $O
timeout 300 python tools/block_probe.py > $O/block_probe_headline.txt 2>&1; cat $O/block_probe_headline.txt
WORKLOAD=c4-table-mis SPP=128 timeout 300 python tools/block_probe.py > $O/block_probe_c4.txt 2>&1; cat $O/block_probe_c4.txt
echo "t = $SECONDS s"
