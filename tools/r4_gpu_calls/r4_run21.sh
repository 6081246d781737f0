cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_21; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 300 python tools/engine_switch_probe.py > $O/engine_switch.txt 2>&1; tail -8 $O/engine_switch.txt
