cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b2; mkdir -p $O
TIMEK=1 ROUNDS=2 bash tools/ab.sh base v1 v5 v6 > $O/ab.txt 2>&1
cat $O/ab.txt
export NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_v5.so
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_v5.txt 2>&1; tail -3 $O/pytest_v5.txt
P="env REPS=2 TIMEK=1 timeout 100 python tools/wf_probe.py"
for R in 24 40; do echo -n "refill $R: "; NORI_HIP_WF_REFILL=$R $P 2>&1 | tail -1; done
for L in 8 24; do echo -n "leaf $L: "; NORI_HIP_WF_LEAF=$L $P 2>&1 | tail -1; done
for I in 16 32; do echo -n "inner $I: "; NORI_HIP_WF_INNER_REPEAT=$I $P 2>&1 | tail -1; done
echo -n "notop: "; NORI_HIP_NO_TOP_IMAGE=1 $P 2>&1 | tail -1
for W in c2-ao-icosphere c4-table-mis; do for V in base v5; do echo -n "$W $V: "; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=$W SPP=64 $P 2>&1 | tail -1; done; done
for V in base v5; do echo -n "c5 $V: "; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=c5-terrain-10m SPP=32 REPS=2 TIMEK=1 timeout 200 python tools/wf_probe.py 2>&1 | tail -1; done
