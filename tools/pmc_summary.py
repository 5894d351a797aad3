"""Summarise a rocprofv3 --pmc counter_collection CSV: sum of each counter per kernel name (+ dispatch count)."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void w2x::", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_INSTS_VALU", 0))):
    n = max(cnt[(k, c)] for c in d)
    print(k, "dispatches", n)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:16.0f}   per-dispatch {v / n:14.0f}")
