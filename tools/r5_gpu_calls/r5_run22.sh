# round 5, GPU call 22: traversal beside shading on the same CUs (two plain streams, wf_extend at 6 waves per SIMD so that a wf_shade workgroup fits) -- tools/overlap2_probe.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_22; mkdir -p $O
timeout 300 python tools/overlap2_probe.py > $O/overlap2_headline.txt 2>&1; cat $O/overlap2_headline.txt
echo "t = $SECONDS s"
