"""Durations of the last N kernel launches of a rocprofv3 kernel trace, in launch order.
usage: python tools/trace_tail.py <dir with *kernel_trace.csv> [N]"""
import csv, glob, re, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-n:]:
    k = re.sub(r"\(anonymous namespace\)::|nrt::|void ", "", r["Kernel_Name"]).split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{k[:44]:44s} {(e - s) / 1e3:9.1f} us   gap {gap:7.1f} us")
    prev_end = e
