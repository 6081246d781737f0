cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_20; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
# the multi-rank code path of bench.py on real CUDA tensors and RCCL, as far as one GPU allows: a launcher-started group of one rank
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 3 > $O/bench_launcher_1rank.json 2> $O/bench_launcher_1rank.err; echo "rc $?"; tail -1 $O/bench_launcher_1rank.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ranks'), d['parity']['ok'], d['cpu_baseline']['value'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 3 --merge gather > $O/bench_launcher_1rank_gather.json 2> $O/bench_launcher_1rank_gather.err; echo "rc $?"; tail -1 $O/bench_launcher_1rank_gather.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ranks'), d['parity']['ok'])"
tail -3 $O/bench_launcher_1rank.err
