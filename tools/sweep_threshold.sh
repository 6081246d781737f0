run() { echo -n "S=$1 Q=$2 L=$3: "; NORI_HIP_TH_SHADE=$1 NORI_HIP_TH_QUICK=$2 NORI_HIP_TH_LEAF=$3 timeout 120 python bench.py --steps 2 --warmup 1 --spp 64 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for Q in 1 4 8 16 32; do run 44 $Q 1; done
for S in 32 40 48 56; do run $S 8 1; done
run 44 8 8
NORI_HIP_CENSUS=1 timeout 120 python bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline 2>&1 | grep census
