# sweeps wf_extend's refill / leaf thresholds (GPU box): bash tools/sweep_wf.sh
cd $GRAFT_REPO_ROOT
for R in 16 24 32 40 48; do for L in 8 16 24 32; do
  echo -n "refill $R leaf $L: "
  NORI_HIP_WF_REFILL=$R NORI_HIP_WF_LEAF=$L REPS=2 ENGINE=wavefront timeout 100 python tools/wf_probe.py 2>&1 | tail -1
done; done
