"""Per-dispatch counter table from a rocprofv3 --kernel-trace --pmc run (csv output): one line per (kernel, grid size) with the
mean of every counter plus the mean duration of the (profiled) dispatches -- and what follows from them:
  clock_MHz            = GRBM_GUI_ACTIVE / duration          (GRBM_GUI_ACTIVE: cycles the GPU was busy, summed over the 8 XCDs -> / 8)
  waves                = SQ_WAVES                            (wavefronts launched)
  valu_per_wave        = SQ_INSTS_VALU / SQ_WAVES
  quad_cycles_per_valu = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (SQ_* cycle counters tick every 4 shader cycles)
  valu_busy            = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES * 4 ...) -- reported raw, see profiles/r03_valu_issue_counters.txt
usage: python tools/pmc_dispatches.py <rocprof output dir> [kernel name filter]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size"]))
            ctr[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])      # ns, of the profiled dispatch
dur = {k: list(v.values()) for k, v in dur.items()}
for key in sorted(ctr):
    m = {k: sum(v) / len(v) for k, v in ctr[key].items()}
    us = (sum(dur[key]) / len(dur[key]) / 1e3) if dur.get(key) else float("nan")
    line = "%-62s grid %8d  n=%-3d %8.1f us" % (key[0], key[1], len(next(iter(ctr[key].values()))), us)
    for k in sorted(m):
        line += "  %s=%.4g" % (k, m[k])
    if "GRBM_GUI_ACTIVE" in m and us == us:
        line += "  | clock %.0f MHz" % (m["GRBM_GUI_ACTIVE"] / 8.0 / us)
    if m.get("SQ_WAVES") and "SQ_INSTS_VALU" in m:
        line += "  valu/wave %.1f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"])
    if m.get("SQ_INSTS_VALU") and "SQ_ACTIVE_INST_VALU" in m:
        line += "  quad-cycles/valu %.3f" % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"])
    if m.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in m:
        line += "  active_valu/busy %.3f" % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CYCLES"])
    print(line)
