"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel name: sum of each counter + dispatch count."""
import csv, glob, os, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(short, r["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    n = max(cnt[(k, c)] for c in agg[k])
    print("%-62s dispatches=%d" % (k, n))
    for c, v in sorted(agg[k].items()):
        print("    %-34s %16.0f  per-dispatch %14.1f" % (c, v, v / max(1, cnt[(k, c)])))
