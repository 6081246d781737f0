cd $GRAFT_REPO_ROOT
for W in 2 3 4 5 6 7 8; do echo -n "wgs/cu $W: "; NORI_HIP_WF_EXTEND_WGS_PER_CU=$W REPS=3 python tools/wf_probe.py 2>&1 | tail -1; done
