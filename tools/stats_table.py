"""Per-kernel table of a rocprofv3 kernel_stats.csv: short name, calls, average us, total us per step.
usage: python tools/stats_table.py <kernel_stats.csv> [steps_in_run]   (steps: graph steps the run executed, to print us/step)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = 0.0
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)[:64]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    line = "%-64s calls %5s  avg %9.2f us  total %10.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, t)
    if steps:
        line += "  %8.2f us/step" % (t / steps)
    print(line)
print("sum %.1f us" % tot + ("  %.2f us/step" % (tot / steps) if steps else ""))
