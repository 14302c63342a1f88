"""Summarise a rocprofv3 --kernel-trace --stats CSV output dir: per-kernel calls / total / average."""
import csv, glob, os, sys
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not files:
    files = glob.glob(os.path.join(d, "**", "*stats*.csv"), recursive=True)
for f in files:
    print("#", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        name = r.get("Name", "")[:90]
        print("%-90s calls=%-6s total_ns=%-12s avg_ns=%-10s pct=%s" % (name, r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
