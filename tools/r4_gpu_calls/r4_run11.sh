cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_11; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
cat > /tmp/c5sweep.py <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
sc = workloads.load("c5", spp=128).scene
r = Renderer(0).upload(sc)
r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda")
def run(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    best = None
    for i in range(3):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["trace_ms"] < best["trace_ms"]: best = st
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    print(env, "trace", round(best["trace_ms"], 2), "shade", round(best["shade_ms"], 2), "frame", round(best["kernel_ms"], 1), flush=True)
run({})
for refill in (16, 24, 40, 48): run({"NORI_HIP_WF_REFILL": refill})
for leaf in (8, 12, 24, 32): run({"NORI_HIP_WF_LEAF": leaf})
for rep in (12, 16, 32, 40, 65): run({"NORI_HIP_WF_INNER_REPEAT": rep})
for wgs in (5, 4): run({"NORI_HIP_WF_EXTEND_WGS_PER_CU": wgs})
run({"NORI_HIP_WF_REFILL": 24, "NORI_HIP_WF_LEAF": 12})
run({"NORI_HIP_WF_REFILL": 40, "NORI_HIP_WF_LEAF": 24})
run({})
PY
timeout 600 python /tmp/c5sweep.py > $O/c5_sweep.txt 2>&1; cat $O/c5_sweep.txt | tail -22
