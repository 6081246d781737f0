# round 4, GPU call 24: fuzzers on the final build (device builders with treelet sweeps included: builder 3 is one of the three drawn)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_23; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 260 python tests/fuzz_intersect.py --seconds 200 --seed 7000 > $O/fuzz.txt 2>&1
timeout 320 python tests/fuzz_engines.py --seconds 240 --seed 1500 --oracle >> $O/fuzz.txt 2>&1
tail -3 $O/fuzz.txt
