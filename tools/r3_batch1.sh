# round-3 batch 1: A/B of the node-fetch / stack / streaming-store variants + diagnostic counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3b1; mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.txt 2>&1
TIMEK=1 ROUNDS=2 bash tools/ab.sh base v1 v2 v3 v4 > $O/ab.txt 2>&1
cat $O/ab.txt
NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_v4.so timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_v4.txt 2>&1; tail -3 $O/pytest_v4.txt
bash tools/pmc_probe.sh base r3b1_base ta tcp lds issue ea > $O/pmc_base.txt 2>&1
bash tools/pmc_probe.sh v2 r3b1_v2 ta tcp lds issue ea > $O/pmc_v2.txt 2>&1
tail -60 $O/pmc_base.txt; tail -60 $O/pmc_v2.txt
