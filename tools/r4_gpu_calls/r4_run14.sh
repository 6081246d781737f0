# round 4, GPU call 14: the wave-per-treelet builder: same tree as the serial form, and what a sweep costs on 10 M triangles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_14; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "treelet or lbvh or builder" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
cat > /tmp/tb.py <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
sc = workloads.load("c5", spp=64).scene
for serial, sweeps in ((0, 0), (0, 1), (0, 2), (0, 3), (1, 1)):
    os.environ["NORI_HIP_TREELET_SERIAL"] = str(serial); os.environ["NORI_HIP_TREELET_SWEEPS"] = str(sweeps)
    t0 = time.time(); r = Renderer(0).upload(sc, builder=3); t1 = time.time() - t0
    info = r.accel_info()
    r.set_option("engine", "wavefront")
    f = torch.zeros(r.frame_shape(), device="cuda")
    best = None
    for i in range(3):
        f.zero_(); st = r.render_into(f, time_kernels=True)
        if best is None or st["trace_ms"] < best["trace_ms"]: best = st
    print(f"serial {serial} sweeps {sweeps}: build_ms {info['build_ms']:.1f} (upload+build wall {t1:.2f} s) nodes {info['n_nodes']} depth {info['max_depth']} sah {info['sah_cost']:.3f} | wf_extend {best['trace_ms']:.2f} ms frame {best['kernel_ms']:.1f}", flush=True)
    r.close()
os.environ.pop("NORI_HIP_TREELET_SERIAL"); os.environ.pop("NORI_HIP_TREELET_SWEEPS")
r = Renderer(0).upload(sc, builder=0); info = r.accel_info(); r.set_option("engine", "wavefront")
f = torch.zeros(r.frame_shape(), device="cuda"); best = None
for i in range(3):
    f.zero_(); st = r.render_into(f, time_kernels=True)
    if best is None or st["trace_ms"] < best["trace_ms"]: best = st
print(f"host SAH: build_ms {info['build_ms']:.1f} nodes {info['n_nodes']} depth {info['max_depth']} sah {info['sah_cost']:.3f} | wf_extend {best['trace_ms']:.2f} ms frame {best['kernel_ms']:.1f}")
PY
timeout 900 python /tmp/tb.py > $O/treelet_10m.txt 2>&1; tail -8 $O/treelet_10m.txt
