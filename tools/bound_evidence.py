"""What binds wf_extend, from measurements only (review of round 5, item 2)  ->  profiles/<tag>_bound_evidence.{json,txt}
   python tools/bound_evidence.py <dir with the csv / txt files tools/collect_bound_evidence.sh wrote> <tag> [output dir, default profiles/]

Two independent measurements per configuration (hl = the headline, c5 = the 10 M-triangle terrain):
 (1) VALU busy from the DYNAMIC instruction mix: rocprofv3 counts the VALU instructions a kernel executed by class
     (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F32, _INT32, _INT64, _CVT; the rest = moves, selects, compares, min / max, bit ops).  The same
     counters on kernels of ONE opcode each (tools/ubench_valu.hip, same box) say which class an opcode is counted in and how many
     SIMD cycles it takes; a class holds opcodes of different cost (FMA_F32: v_fma 2.4 cycles, v_pk_fma / v_fma_mix 4.3; INT32: v_add_u32
     3.0, shifts / bfe / mul_lo 4.3; rest: v_mov / v_and 3.0 - 3.3, everything else 4.3), so the figure is an interval:
         busy_lo / busy_hi = sum over classes of N_class x (cheapest / dearest opcode of the class) / (1024 SIMDs x elapsed cycles)
     No static listing of the kernel text, no clamp.
 (2) Sensitivities: the lab variants add to EVERY node step 16 (BVH2 loop) / 32 (wide step) v_mov, one more 16-B load of the node, or 64
     idle cycles (s_nop).  What a variant added is measured (difference of the instruction counters), what it cost is measured (HIP
     events around wf_extend, variants alternated on one box).  Normalised: s = (dt / t) / (dR / R) for R = VALU cycles (instructions x
     cycles of the mix) or vector-memory read instructions; s = 1 means the kernel's time is that resource, 0 that it has slack.
     The idle-cycle variant prices per-wave latency: cycles_added_per_wave_step x steps / (8 waves per SIMD x 1024 SIMDs) is what 64 more
     cycles per step cost if every SIMD always had its 8 waves to choose from and no instruction slot to spare."""
import collections, csv, json, os, re, sys

d, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
CLK_GHZ = 2.4
CLASSES = ["FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "INT64", "CVT"]


def load(path, want=lambda k: True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", r["Kernel_Name"]).split("(")[0]
        if want(k):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def summed(acc, prefix):
    out = collections.defaultdict(float)
    for k, c in acc.items():
        if k.startswith(prefix):
            for n, v in c.items():
                out[n] += v
    return out


# ---- (0) the one-opcode kernels: class and cycles of every opcode
ub_mix, ub_el = load(os.path.join(d, f"{tag}_ubench_mix_counter_collection.csv")), load(os.path.join(d, f"{tag}_ubench_elapsed_counter_collection.csv"))
opcodes = {}
for k, v in ub_mix.items():
    n = v.get("SQ_INSTS_VALU", 0.0)
    e = ub_el.get(k, {})
    if not n or not e.get("SQ_INSTS_VALU") or k == "k_cndmask":      # (k_cndmask's chain waits for its own vcc: latency, not issue cost)
        continue
    shares = {c: v.get("SQ_INSTS_VALU_" + c, 0.0) / n for c in CLASSES}
    cls = max(shares, key=shares.get) if max(shares.values()) > 0.5 else "REST"
    per = 2.0 if k == "k_cmp_cnd" else 1.0      # two instructions per body
    opcodes[k] = {"class": cls, "cycles": round(1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0 / e["SQ_INSTS_VALU"], 2)}
cost = {}
for c in CLASSES + ["REST"]:
    cyc = [o["cycles"] for o in opcodes.values() if o["class"] == c]
    cost[c] = (min(cyc), max(cyc)) if cyc else (4.3, 4.3)

times = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open(os.path.join(d, f"{tag}_sens_times.txt")):
    m = re.match(r"(\S+) (\w+): trace ([\d.]+) shade", line)
    if m:
        times[m.group(1)][m.group(2)].append(float(m.group(3)))

def source_sha():      # as bench.py device_source_sha / tools/summarize_profile.py
    import glob, hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "nori_amd", "csrc", "device", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    m = re.search(r"^HIP_FLAGS = (.*)$", open(os.path.join(ROOT, "__graft_entry__.py")).read(), re.M)
    h.update((m.group(1) if m else "").encode())
    return h.hexdigest()[:16]


out = {"tag": tag, "device_source_sha": os.environ.get("NORI_EVIDENCE_SHA") or source_sha(), "opcode_calibration": opcodes, "class_cycles_lo_hi": cost, "configs": {}}
lines = []
for short, wl in (("hl", "pa4-cbox-path_mis"), ("c5", "c5-terrain-10m")):
    mix = summed(load(os.path.join(d, f"{tag}_mix_{short}_counter_collection.csv")), "wf_extend")
    el = summed(load(os.path.join(d, f"{tag}_elapsed_{short}_counter_collection.csv")), "wf_extend")
    n = mix["SQ_INSTS_VALU"]
    by_class = {c: mix.get("SQ_INSTS_VALU_" + c, 0.0) for c in CLASSES}
    by_class["REST"] = n - sum(by_class.values())
    elapsed = el["GRBM_GUI_ACTIVE"] / 8.0
    lo = sum(by_class[c] * cost[c][0] for c in by_class) / (1024.0 * elapsed)
    hi = sum(by_class[c] * cost[c][1] for c in by_class) / (1024.0 * elapsed)
    cyc_lo, cyc_hi = lo * 1024.0 * elapsed / n, hi * 1024.0 * elapsed / n
    cfg = {"workload": wl, "wf_extend_valu_instructions": int(n), "elapsed_cycles": elapsed,
           "class_share": {c: round(v / n, 4) for c, v in by_class.items()},
           "valu_cycles_per_instr_lo_hi": [round(cyc_lo, 3), round(cyc_hi, 3)],
           "valu_busy_measured_lo_hi": [round(lo, 4), round(hi, 4)],
           "valu_lanes_per_instr": round(el["SQ_THREAD_CYCLES_VALU"] / el["SQ_ACTIVE_INST_VALU"], 2) if el.get("SQ_ACTIVE_INST_VALU") else None}
    base = summed(load(os.path.join(d, f"{tag}_sens_{short}_base_counter_collection.csv")), "wf_extend")
    t0 = sum(times[wl]["base"]) / len(times[wl]["base"])
    cfg["base"] = {"trace_ms": round(t0, 3), "valu_instr": int(base["SQ_INSTS_VALU"]), "vmem_rd_instr": int(base["SQ_INSTS_VMEM_RD"]), "lds_instr": int(base["SQ_INSTS_LDS"]),
                   "salu_instr": int(base["SQ_INSTS_SALU"]), "wait_inst_any_frac": round(base["SQ_WAIT_INST_ANY"] / base["SQ_WAVE_CYCLES"], 4),
                   "active_inst_any_frac": round(base["SQ_ACTIVE_INST_ANY"] / base["SQ_WAVE_CYCLES"], 4)}
    per_step = 16 if short == "hl" else 32
    steps = None
    sens = {}
    for v in ("valu", "load", "idle"):
        c = summed(load(os.path.join(d, f"{tag}_sens_{short}_{v}_counter_collection.csv")), "wf_extend")
        t = sum(times[wl][v]) / len(times[wl][v])
        dv, dm = c["SQ_INSTS_VALU"] - base["SQ_INSTS_VALU"], c["SQ_INSTS_VMEM_RD"] - base["SQ_INSTS_VMEM_RD"]
        e = {"trace_ms": round(t, 3), "dt_over_t": round(t / t0 - 1.0, 4), "added_valu_instr": int(dv), "added_vmem_rd_instr": int(dm)}
        if v == "valu":
            steps = dv / per_step                                   # wave-level node steps of one render pass
            mov = opcodes["k_mov"]["cycles"]
            r_lo, r_hi = dv * mov / (n * cyc_hi), dv * mov / (n * cyc_lo)      # share of the kernel's VALU cycles that was added
            e["added_valu_cycles_share_lo_hi"] = [round(r_lo, 4), round(r_hi, 4)]
            e["sensitivity_lo_hi"] = [round(e["dt_over_t"] / r_hi, 3), round(e["dt_over_t"] / r_lo, 3)]
            e["ms_if_valu_throughput_bound"] = round(dv * mov / 1024.0 / (CLK_GHZ * 1e6), 3)
        if v == "load":
            e["added_share_of_vmem_rd"] = round(dm / base["SQ_INSTS_VMEM_RD"], 4)
            e["sensitivity"] = round(e["dt_over_t"] / (dm / base["SQ_INSTS_VMEM_RD"]), 3) if dm > 0 else None
        if v == "idle" and steps:
            e["ms_if_latency_bound_at_8_waves_per_simd"] = round(64.0 * steps / (8.0 * 1024.0) / (CLK_GHZ * 1e6), 3)
            e["effective_waves_per_simd"] = round(8.0 * e["ms_if_latency_bound_at_8_waves_per_simd"] / max(1e-9, t - t0), 2) if t > t0 else None
        sens[v] = e
    cfg["wave_level_node_steps"] = int(steps) if steps else None
    cfg["variants"] = sens
    if times[wl].get("base_no_lds_image"):
        cfg["no_lds_image_trace_ms"] = round(sum(times[wl]["base_no_lds_image"]) / len(times[wl]["base_no_lds_image"]), 3)
    # the verdict, from the numbers above only
    sv = sum(sens["valu"]["sensitivity_lo_hi"]) / 2.0
    sm = sens["load"].get("sensitivity") or 0.0
    idle_cost = sens["idle"]["dt_over_t"]
    if idle_cost >= 0.5 * sens["valu"]["dt_over_t"]:
        verdict = "per-wave latency"
        why = ("64 idle cycles per node step cost %.1f %% of wf_extend, as much as %d more v_mov (%.1f %%): a wave's own instruction chain, not the SIMD's VALU throughput "
               "(VALU sensitivity %.2f of 1, busy %.2f - %.2f by the dynamic mix), sets the pace -- 8 waves per SIMD (the hardware's maximum) behave like %s"
               % (100 * idle_cost, per_step, 100 * sens["valu"]["dt_over_t"], sv, lo, hi, sens["idle"].get("effective_waves_per_simd")))
    elif abs(sm - sv) < 0.1:
        verdict = "valu + vector-memory issue"
        why = ("throughput-bound on two resources at once: %d more v_mov per node step cost %.1f %% (sensitivity %.2f of 1), one more load (+%.1f %% of the kernel's read instructions) %.1f %% "
               "(sensitivity %.2f), 64 idle cycles %.1f %% (absorbed: the waves hide latency); VALU busy %.2f - %.2f by the dynamic mix"
               % (per_step, 100 * sens["valu"]["dt_over_t"], sv, 100 * sens["load"]["added_share_of_vmem_rd"], 100 * sens["load"]["dt_over_t"], sm, 100 * idle_cost, lo, hi))
    elif sm > sv:
        verdict = "vector-memory issue"
        why = ("one more load per node step (+%.1f %% of the kernel's read instructions) costs %.1f %% (sensitivity %.2f), %d more v_mov %.1f %% (sensitivity %.2f), 64 idle cycles %.1f %%: "
               "the waves hide latency, the VALU (busy %.2f - %.2f by the dynamic mix) has slack, the memory pipe's instruction rate does not"
               % (100 * sens["load"]["added_share_of_vmem_rd"], 100 * sens["load"]["dt_over_t"], sm, per_step, 100 * sens["valu"]["dt_over_t"], sv, 100 * idle_cost, lo, hi))
    else:
        verdict = "valu"
        why = "VALU sensitivity %.2f, load sensitivity %.2f, idle %.1f %%" % (sv, sm, 100 * idle_cost)
    cfg["verdict"], cfg["verdict_from"] = verdict, why
    out["configs"][short] = cfg
    lines += [f"== {wl}: wf_extend {t0:.2f} ms, {n:.4g} VALU instructions, {int(steps):d} wave-level node steps",
              "   dynamic mix: " + ", ".join(f"{c} {v / n:.3f}" for c, v in by_class.items()),
              f"   VALU busy (dynamic mix x measured cycles per opcode class): {lo:.3f} - {hi:.3f}; {cfg['valu_lanes_per_instr']} of 64 lanes per instruction",
              f"   + {per_step} v_mov per node step: {sens['valu']['dt_over_t'] * 100:+.1f} % ({sens['valu']['ms_if_valu_throughput_bound']} ms if VALU-throughput bound; sensitivity {sens['valu']['sensitivity_lo_hi']})",
              f"   + one 16-B load per node step: {sens['load']['dt_over_t'] * 100:+.1f} % for {100 * sens['load']['added_share_of_vmem_rd']:+.1f} % read instructions (sensitivity {sens['load'].get('sensitivity')})",
              f"   + 64 idle cycles per node step: {sens['idle']['dt_over_t'] * 100:+.1f} % ({sens['idle'].get('ms_if_latency_bound_at_8_waves_per_simd')} ms if latency bound with 8 waves per SIMD to choose from)",
              (f"   without the LDS image: {cfg['no_lds_image_trace_ms']} ms" if "no_lds_image_trace_ms" in cfg else ""),
              f"   => bound: {verdict} -- {why}"]
json.dump(out, open(os.path.join(OUT_DIR, f"{tag}_bound_evidence.json"), "w"), indent=1)
open(os.path.join(OUT_DIR, f"{tag}_bound_evidence.txt"), "w").write(__doc__ + "\n" + "\n".join(l for l in lines if l) + "\n")
print("\n".join(l for l in lines if l))
