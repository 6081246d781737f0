"""Cycles per VALU instruction from a rocprofv3 counter run of tools/ubench_valu.hip:
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/ub -o u -- <ubench_valu>
   python tools/ubench_counters.py /tmp/ub"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
print("# kernel              INSTS_VALU   ACTIVE/INST  elapsed cycles  SIMD-cycles per instruction (= 1024 x GRBM_GUI_ACTIVE / 8 / INSTS)")
for k, v in acc.items():
    n = v.get("SQ_INSTS_VALU", 0.0)
    if n:
        el = v["GRBM_GUI_ACTIVE"] / 8.0
        print("%-20s %.4g  %.2f  %.4g  %.2f" % (k, n, v["SQ_ACTIVE_INST_VALU"] / n, el, 1024.0 * el / n))
