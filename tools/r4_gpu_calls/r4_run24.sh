# round 4, GPU call 25: does a device tree with more treelet sweeps reach the host's SAH tree on the small scenes?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_24; mkdir -p $O
( time timeout 600 python -c "import torch; torch.zeros(1).cuda(); print('torch ok')" ) 2>&1 | tail -3
cat > /tmp/sw.py <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from nori_amd.render import Renderer
from nori_amd import workloads
for name, spp in (("pa4-cbox-path_mis", 64), ("c2-ao-icosphere", 64), ("c4-table-mis", 16)):
    sc = workloads.load(name, spp=spp).scene
    for builder, sweeps in ((0, 0), (3, 0), (3, 2), (3, 3), (3, 4), (3, 6), (3, 8)):
        os.environ["NORI_HIP_TREELET_SWEEPS"] = str(sweeps)
        r = Renderer(0).upload(sc, builder=builder); info = r.accel_info(); r.set_option("engine", "wavefront")
        f = torch.zeros(r.frame_shape(), device="cuda"); best = None
        for i in range(4):
            f.zero_(); st = r.render_into(f, time_kernels=True)
            if best is None or st["trace_ms"] < best["trace_ms"]: best = st
        print(f"{name} {'host SAH' if builder == 0 else 'PLOC + %d sweeps' % sweeps}: build {info['build_ms']:.1f} ms depth {info['max_depth']} nodes {info['n_nodes']} | wf_extend {best['trace_ms']:.2f} ms", flush=True)
        r.close()
        if builder == 0: continue
PY
timeout 400 python /tmp/sw.py > $O/sweeps_small.txt 2>&1; grep -v amdgpu $O/sweeps_small.txt | tail -24
