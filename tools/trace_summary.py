"""Summarise a rocprofv3 kernel trace: per-kernel totals and the first launches of the path loop.
usage: python tools/trace_summary.py <dir with *kernel_trace.csv> [n_first]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; nfirst = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", n)
    return n.split("(")[0]
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = short(r["Kernel_Name"]); dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[k][0] += 1; tot[k][1] += dt
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k:50s} calls {c:5d} total {t/1e6:9.3f} ms avg {t/c/1e3:9.1f} us")
# last full pass: from the last wf_generate on
gi = [i for i, r in enumerate(rows) if "wf_generate" in r["Kernel_Name"] or re.search(r"wf_extend<[^>]*true>\(", r["Kernel_Name"])]
if gi:
    seq = rows[gi[-1]:]
    t0 = int(seq[0]["Start_Timestamp"])
    print("last pass:")
    for r in seq[: 3 * nfirst]:
        k = short(r["Kernel_Name"])
        if "swap" in k: continue
        print(f"  +{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms {k:40s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us")
    print(f"  pass wall {(int(seq[-1]['End_Timestamp'])-t0)/1e6:.3f} ms")
