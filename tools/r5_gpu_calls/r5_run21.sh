# round 5, GPU call 21: chunking of wf_extend (static chunk limit, dynamic divisor) on one device's share of the headline at N = 8
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_21; mkdir -p $O
{
for S in 1024 0 128 256 512 2048 4095; do for D in 8 2 4 15; do
  echo -n "static $S dyndiv $D: "
  NORI_HIP_WF_STATIC=$S NORI_HIP_WF_DYNDIV=$D TILE_MOD=8 TIMEK=1 REPS=4 timeout 100 python tools/wf_probe.py 2>&1 | grep -v amdgpu | tail -2 | tr '\n' '|'; echo
done; done
} > $O/share_chunking.txt 2>&1; cat $O/share_chunking.txt | cut -c1-200
echo "t = $SECONDS s"
