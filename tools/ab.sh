# A/B timing of library variants on ONE box (box-to-box spread is ~2 %, larger than most tuning effects):
#   bash tools/ab.sh base v1 v2      -> nori_amd/lib/libnori_hip_<name>.so, alternated ROUNDS times
cd $GRAFT_REPO_ROOT
for k in $(seq ${ROUNDS:-3}); do for V in "$@"; do
  echo -n "$V: "
  NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so REPS=3 ENGINE=${ENGINE:-wavefront} timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done
