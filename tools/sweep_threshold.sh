for R in 8 16 24 32 48; do for LT in 1 8 16; do echo -n "refill=$R leaf=$LT: "; NORI_HIP_WF_REFILL=$R NORI_HIP_WF_LEAF=$LT REPS=2 timeout 60 python tools/wf_probe.py 2>&1 | tail -1; done; done
