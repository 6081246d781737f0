run() { echo -n "WGS=$1: "; NORI_HIP_TARGET_WGS=$1 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['workgroups'])"; }
for W in 4096 8192 16384 32768 65536; do run $W; done
