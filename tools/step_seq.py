"""Kernel sequence of ONE graph-replayed train step from a rocprofv3 --kernel-trace CSV: name, duration, gap to the previous kernel.
usage: python tools/step_seq.py <kernel_trace.csv> [which_step_from_the_end=3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:56]
# a step starts at the feature kernel
starts = [i for i, r in enumerate(rows) if "feat512_stream" in r["Kernel_Name"] or "fused_feat512" in r["Kernel_Name"]]
i0 = starts[-back]; i1 = starts[-back + 1] if back > 1 else len(rows)
prev_end = None
tot = 0.0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-56s %9.2f us   gap %7.2f   grid %s" % (short(r["Kernel_Name"]), (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
    tot += (e - s) / 1e3
    prev_end = e
print("kernels %.1f us, span %.1f us" % (tot, (int(rows[i1 - 1]["End_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3))
