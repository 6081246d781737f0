cd $GRAFT_REPO_ROOT
for L in 2 4 6 8; do for C in 0.35 0.6 1; do
  echo -n "leaf $L tri cost $C: "
  NORI_HIP_SAH_LEAF=$L NORI_HIP_SAH_TRI_COST=$C REPS=2 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done
