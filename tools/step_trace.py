"""Per-dispatch kernel timeline of ONE graph-replayed train step from a rocprofv3 --kernel-trace csv.
usage: python tools/step_trace.py <kernel_trace.csv>   (prints the dispatches of the last complete step, in order)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "feat512_stream" in r["Kernel_Name"] or "fused_feat512" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    g = [r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))]
    print("%8.1f  +%5.1f gap  %7.1f us  %-46s grid %s wg %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm[:46], g[0], g[1]))
    prev_end = e
    tot += e - s
print("step span %.1f us, kernel time %.1f us, %d dispatches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, tot / 1e3, b - a))
