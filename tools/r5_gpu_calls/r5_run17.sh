# round 5, GPU call 17: fuzzers on the final build (host builder with spatial splits + re-insertion by default; a second run with the splits forced: margin 1, budget 1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5_17; mkdir -p $O
timeout 160 python tests/fuzz_intersect.py --seconds 100 --seed 9000 > $O/fuzz.txt 2>&1
NORI_HIP_SBVH=1.0 NORI_HIP_SBVH_MARGIN=1.0 timeout 160 python tests/fuzz_intersect.py --seconds 100 --seed 9500 >> $O/fuzz.txt 2>&1
timeout 220 python tests/fuzz_engines.py --seconds 140 --seed 2500 --oracle >> $O/fuzz.txt 2>&1
NORI_HIP_SBVH=1.0 NORI_HIP_SBVH_MARGIN=1.0 timeout 160 python tests/fuzz_engines.py --seconds 90 --seed 3500 --oracle >> $O/fuzz.txt 2>&1
grep -v amdgpu.ids $O/fuzz.txt | tail -8
echo "t = $SECONDS s"
