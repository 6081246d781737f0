"""Build-time guards: properties of the COMPILED kernels that no result check can see.

The node loop of wf_extend is a block of hand-written gfx950 assembly on fixed registers (v32-v45, vcc; wavefront.hip,
bvh2q_node_loop_asm).  It is correct under any register allocation of the C++ around it (clobber list), but it is only FAST
while that C++ stays within 64 VGPRs without scratch -- 8 waves per SIMD, 2 workgroups of 1024 threads per CU, which is what
the LDS image and the persistent grid are sized for.  A register regression would cost milliseconds silently; here it fails.
hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def wavefront_kernels(tmp_path_factory):
    import __graft_entry__ as ge
    from kernel_resources import kernels
    out = tmp_path_factory.mktemp("asm") / "wavefront.s"
    flags = [f for f in ge.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    p = subprocess.run([ge.HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(ge.DEV, "wavefront.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rows = {r["demangled"].replace("(anonymous namespace)::", ""): r for r in kernels(str(out))}
    return rows, open(out).read()


def _find(rows, prefix):
    hit = [r for name, r in rows.items() if name.startswith("void " + prefix)]
    assert len(hit) == 1, (prefix, [n for n in rows if prefix.split("<")[0] in n][:8])
    return hit[0]


def test_the_production_traversal_kernels_keep_their_register_budget(wavefront_kernels):
    rows, _ = wavefront_kernels
    # wf_extend<STACK, SPILL, COUNT, MODE, WIDE, ASM, BLOCK>, MODE 0 = stored paths only: the hand-written loop, later passes
    # (88 % of the kernel's time)
    for stack in (16, 24, 32):
        k = _find(rows, f"wf_extend<{stack}, false, false, 0, false, true, 1024>")
        assert k["vgpr"] <= 64 and k["agpr"] == 0 and k["scratch"] == 0, k
        # trees deeper than the LDS stack (LdsStackHybrid): same budget
        k = _find(rows, f"wf_extend<{stack}, true, false, 0, false, true, 1024>")
        assert k["vgpr"] <= 64 and k["scratch"] == 0, k
    # every BVH2 variant must fit 8 waves per SIMD (two 1024-thread workgroups per CU), scratch or not
    for name, k in rows.items():
        if name.startswith("void wf_extend<") and name.split(">")[0].endswith(", 1024"):
            assert k["vgpr"] <= 64, (name, k)
    # wide trees: 7 waves per SIMD (5 in the first pass) -- wavefront_render sizes the persistent grids for that
    k = _find(rows, "wf_extend<16, true, false, 0, true, false, 256>")
    assert k["vgpr"] <= 72 and k["scratch"] == 0, k      # 7 waves per SIMD
    k = _find(rows, "wf_extend<16, true, false, 1, true, false, 256>")      # MODE 1: a pass's NEW paths (the first pass; regeneration)
    assert k["vgpr"] <= 96 and k["scratch"] == 0, k
    # stored and new paths are traced by separate launches of the two pure kernels: no wf_extend carries both refills
    assert not [n for n in rows if n.startswith("void wf_extend<") and n.split(",")[3].strip() == "2"]


def test_the_shading_kernels_do_not_spill(wavefront_kernels):
    rows, _ = wavefront_kernels
    for integ in range(7):
        # wf_shade<INTEG, MODE, LDSTAB, MATSET>: integrators that ask a BSDF are compiled per material set (rt_path.h: 1 = all
        # diffuse, 7 = no microfacet, 15 = any); the others once.  MODE 0 stored paths, 1 the first pass, 2 both (regeneration:
        # two loops over two instantiations of the round -- in one loop body the kernel spilled)
        for matset in ((1, 7, 15) if integ >= 3 else (15,)):
            for mode in (0, 1, 2):
                for lds_tables in ("true", "false"):
                    k = _find(rows, f"wf_shade<{integ}, {mode}, {lds_tables}, {matset}>")
                    assert k["vgpr"] <= 128 and k["scratch"] == 0, k
    # what the specialisation is for: the all-diffuse kernel of the headline scene is smaller than the general one
    assert _find(rows, "wf_shade<6, 0, true, 1>")["vgpr"] < _find(rows, "wf_shade<6, 0, true, 15>")["vgpr"]


def test_the_node_loop_reads_a_node_in_two_loads(wavefront_kernels):
    """The point of the 32-B records: two vector-memory instructions per node step.  The loop body is
    the text between the labels the assembly block defines; the compiler must not have widened, split or duplicated it."""
    _, text = wavefront_kernels
    start = text.index("_ZN12_GLOBAL__N_19wf_extendILi16ELb0ELb0ELi0ELb0ELb1ELi1024EEEvN3nrt8DevSceneENS_5WfBufEiiNS_7WfBatchE:")
    body = text[start:text.index(".end_amdhsa_kernel", start)]
    i = body.index("v_bfi_b32 v40")
    loop = body[body.rindex("s_and_b64 exec", 0, i):body.index("s_cbranch_scc1", i)]
    assert loop.count("global_load_dwordx4") == 2 and loop.count("ds_read_b128") == 2
    assert loop.count("v_cvt_f32_u32_sdwa") == 12 and loop.count("v_fma_f32") == 12
    assert "scratch_" not in loop and "buffer_load" not in loop


def test_hand_declared_rccl_interface_agrees_with_the_installed_header(tmp_path):
    """group.hip reaches RCCL through dlopen and hand-written prototypes (nori_amd/csrc/device/rccl_abi.h) -- which no run with more
    than one rank has ever exercised.  tests/abi/rccl_abi_check.cpp holds them against <rccl/rccl.h>: arity, ABI class of every
    parameter and return value, the enum values the calls pass.  And the check itself is checked: a header with one constant
    changed, or one parameter dropped, must fail to compile."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no RCCL header in this image")
    src = os.path.join(ROOT, "tests", "abi", "rccl_abi_check.cpp")
    cmd = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-fsyntax-only"]
    p = subprocess.run(cmd + [src], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    abi = open(os.path.join(ROOT, "nori_amd", "csrc", "device", "rccl_abi.h")).read()
    body = open(src).read().replace('#include "../../nori_amd/csrc/device/rccl_abi.h"', '#include "rccl_abi.h"')
    for k, (old, new) in enumerate((("int ncclFloat = 7", "int ncclFloat = 8"), (", ncclSum = 0;", ", ncclSum = 1;"),
                                    ("size_t count, int datatype, int op, int root,", "size_t count, int datatype, int root,"),
                                    ("(*Recv_t)(void *recvbuff,", "(*Recv_t)(int recvbuff,"))):
        assert abi.count(old) == 1, old
        d = tmp_path / f"m{k}"
        d.mkdir()
        (d / "rccl_abi.h").write_text(abi.replace(old, new))
        (d / "check.cpp").write_text(body)
        q = subprocess.run(cmd + ["-I", str(d), str(d / "check.cpp")], capture_output=True, text=True, timeout=300)
        assert q.returncode != 0 and "static assertion failed" in q.stderr, (old, q.stderr[-500:])
