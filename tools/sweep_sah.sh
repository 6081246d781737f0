cd $GRAFT_REPO_ROOT
for C in 0.5 1 1.5 2 3 4; do
  echo -n "tri cost $C: "
  NORI_HIP_SAH_TRI_COST=$C NORI_HIP_CENSUS= REPS=2 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done
