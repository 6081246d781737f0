"""Static VALU instruction mix of the render kernels -> profiles/<tag>_valu_mix.json.

SQ_ACTIVE_INST_VALU counts ONE unit per VALU instruction (two per transcendental), not the cycles the instruction keeps
its SIMD busy (profiles/r3_valu_counter_calibration.txt: 2.4 cycles for v_fma, 2.5 - 3.2 for v_mov / v_add / v_mul / v_and, ~4.3 for
v_min / v_max / v_cmp / v_cndmask / v_mul_lo / v_pk_* / v_cvt_* / v_bfi / v_perm / shifts / three-operand integer ops, ~8.2 for
v_rcp / v_sqrt; the fast class is priced at its lower end, 2.4, so the busy fraction is a lower bound).  How busy the VALU is therefore needs
the kernel's instruction mix; this script takes it from the compiler's own assembly of the kernel (all instructions of
the kernel weighted equally -- the hot loops are most of the text) and tools/summarize_profile.py multiplies:
    valu_busy_frac = SQ_INSTS_VALU x cycles_per_instr / (SIMDs x elapsed cycles)          (<= 1 by construction of the table)
usage: python tools/valu_mix.py <tag>      (compiles wavefront.hip / film.hip / nori_hip.hip with --save-temps into a temp dir)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
FAST, SLOW, TRANS = 2.4, 4.2, 8.2
SLOW_RE = re.compile(r"^v_(min|max|med3|cmp|cmpx|cndmask|mul_lo|mul_hi|pk_|mad_u64|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_f64|mul_f64|fma_f64|"
                     r"cvt_|perm_|bfi_|and_or_|lshl_add_|lshl_or_|bfe_|lshlrev_|lshrrev_|ashrrev_|sad_|fma_mix|mad_mix|mad_u32|mad_i32|add3_|xad_|or3_|alignbit)")
TRANS_RE = re.compile(r"^v_(rcp|sqrt|rsq|exp|log|sin|cos)")
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
out = {}
with tempfile.TemporaryDirectory() as tmp:
    for src in ("wavefront.hip", "film.hip", "nori_hip.hip"):
        flags = [f for f in g.HIP_FLAGS if f != "-shared"]
        subprocess.run([g.HIPCC] + flags + ["--save-temps", "-c", os.path.join(g.DEV, src), "-o", os.path.join(tmp, src + ".o")], cwd=tmp, check=True, capture_output=True)
        asm = open(os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end\d+:", asm, re.M | re.S):
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", name).split("(")[0]
            if not name.startswith(("wf_", "film_", "render_kernel")):
                continue
            n = {"fast": 0, "slow": 0, "trans": 0}
            for line in m.group(0).splitlines():
                op = line.strip().split(" ")[0]
                if not op.startswith("v_"):
                    continue
                n["trans" if TRANS_RE.match(op) else "slow" if SLOW_RE.match(op) else "fast"] += 1
            tot = sum(n.values())
            if tot:
                out[name] = dict(n, cycles_per_instr=round((n["fast"] * FAST + n["slow"] * SLOW + n["trans"] * TRANS) / tot, 3),
                                 active_units_per_instr=round((tot + n["trans"]) / tot, 4))
path = os.path.join(ROOT, "profiles", f"{tag}_valu_mix.json")
json.dump({"cycles": {"fast": FAST, "slow": SLOW, "trans": TRANS}, "source": "profiles/r3_valu_counter_calibration.txt", "kernels": out}, open(path, "w"), indent=1)
for k, v in sorted(out.items()):
    if "<6," in k or "<16, false, false" in k or k.startswith("film"): print(k, v)
print(path)
