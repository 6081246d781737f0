#!/bin/bash
# round 6, call 18: the whole GPU suite with auto = the device builder (the C++ host's path), smoke, the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_21
timeout 2400 python -m pytest tests -m gpu -x -q > ${O}_pytest_gpu.txt 2>&1; tail -15 ${O}_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.txt 2>&1; tail -2 ${O}_smoke.txt
timeout 600 python bench.py > ${O}_bench_default.json 2> ${O}_bench_default.err; tail -c 1500 ${O}_bench_default.json
