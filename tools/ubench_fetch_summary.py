"""FETCH_SIZE / WRITE_SIZE of tools/ubench_fetch.hip against its known byte counts:
    python tools/ubench_fetch_summary.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run>
Both counters are in KB; every kernel runs twice (warm + timed): per launch = sum / 2."""
import collections, csv, glob, sys
GB = {"k_stream_read": 2 * 2**30, "k_gather<4>": 2**24 * 16 * 64, "k_gather<2>": 2**24 * 16 * 32, "k_gather<6>": 2**24 * 16 * 96,
      "k_stream_write": 2 * 2**30, "k_scatter16<false>": 2**24 * 16 * 16, "k_scatter16<true>": 2**24 * 16 * 16}
def load(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0].replace("void ", ""); acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}
rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# kernel                algorithmic GB   FETCH_SIZE GB (x1)  algorithmic / FETCH_SIZE   WRITE_SIZE GB (x1)  algorithmic / WRITE_SIZE")
for k, b in GB.items():
    f = rd.get(k, 0.0) * 1024.0; w = wr.get(k, 0.0) * 1024.0
    reads = "read" in k or "gather" in k
    print(f"{k:22s} {b / 1e9:10.3f} {f / 1e9:16.3f} {(b / f if reads and f else float('nan')):18.3f} {w / 1e9:22.3f} {(b / w if not reads and w else float('nan')):18.3f}")
