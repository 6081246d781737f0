# round 5, GPU call 20: one device's share of the headline at N = 8 (tile_mod 8) -- where do its milliseconds go? readback cadence, hand-over threshold
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_20; mkdir -p $O
{
for V in "default" "NORI_HIP_WF_SYNC_EVERY=1" "NORI_HIP_WF_SYNC_EVERY=3" "NORI_HIP_WF_SYNC_EVERY=12" "NORI_HIP_WF_FINISH_PATHS=131072" "NORI_HIP_WF_FINISH_PATHS=1048576" "NORI_HIP_WF_FINISH_PATHS=2097152" "NORI_HIP_CENSUS=1"; do
  echo "== $V"
  if [ "$V" = default ]; then E=""; else E="$V"; fi
  env $E TILE_MOD=8 TIMEK=1 REPS=4 timeout 100 python tools/wf_probe.py 2>&1 | grep -v amdgpu | tail -5
done
} > $O/share_of_eight.txt 2>&1; cat $O/share_of_eight.txt | cut -c1-200
echo "t = $SECONDS s"
