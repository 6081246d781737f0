cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_15; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
cat > /tmp/tb.py <<'PY'
import os, sys, time
sys.path.insert(0, ".")
from nori_amd.render import Renderer
from nori_amd import workloads
sc = workloads.load("c5", spp=64).scene
for sweeps in (0, 1, 2):
    os.environ["NORI_HIP_TREELET_SWEEPS"] = str(sweeps)
    r = Renderer(0).upload(sc, builder=3); info = r.accel_info()
    print(f"sweeps {sweeps}: build_ms {info['build_ms']:.1f} nodes {info['n_nodes']} depth {info['max_depth']}", flush=True)
    r.close()
PY
NORI_HIP_LIBRARY=$GRAFT_REPO_ROOT/nori_amd/lib/libnori_hip_nofence.so timeout 600 python /tmp/tb.py > $O/nofence.txt 2>&1; tail -4 $O/nofence.txt
