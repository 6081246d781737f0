"""Sum rocprofv3 --pmc counters per kernel.  usage: python tools/pmc_summary.py <dir> [kernel substring]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", r["Kernel_Name"]).split("(")[0]
        if sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()): print(f"   {n:28s} {v:.4g}")
    if "SQ_INSTS_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"   lanes per VALU instr ~ {c['SQ_THREAD_CYCLES_VALU'] / max(c['SQ_ACTIVE_INST_VALU'], 1):.1f} (THREAD_CYCLES/ACTIVE_INST)")
