cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
TIMEK=1 ROUNDS=2 bash tools/ab.sh cur sort > $O/ab.txt 2>&1; cat $O/ab.txt
echo -n "no premul: "; NORI_HIP_FILM_NO_PREMUL=1 REPS=3 TIMEK=1 timeout 100 python tools/wf_probe.py 2>&1 | tail -1
for W in c2-ao-icosphere c4-table-mis; do for V in cur sort; do echo -n "$W $V: "; NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_$V.so WORKLOAD=$W SPP=64 REPS=2 TIMEK=1 timeout 100 python tools/wf_probe.py 2>&1 | tail -1; done; done
