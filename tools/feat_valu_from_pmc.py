"""VALU issue utilisation of the fused feature kernel from a rocprofv3 --pmc pass over tools/bench_features.py
(SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE), per kind (first template
argument: 0 spectrogram, 1 mel, 2 log-mel, 3 MFCC) and launch size (grid = threads; B = 256 / 1024 / 2048).
  kernel cycles       = GRBM_GUI_ACTIVE / 8 XCDs
  valu_insts_per_simd = SQ_INSTS_VALU / 1024 SIMDs              (wave-level instructions issued per SIMD)
  valu_issue_frac     = valu_insts_per_simd * 4 / kernel cycles  (a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles)
usage: feat_valu_from_pmc.py <rocprof output dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "feat512_stream_kernel" not in r["Kernel_Name"]:
            continue
        key = (re.search(r"feat512_stream_kernel<[^>]*>", r["Kernel_Name"]).group(0), int(r["Grid_Size"]))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (name, grid), c in sorted(rows.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    ips = m["SQ_INSTS_VALU"] / 1024.0
    out["%s grid=%d" % (name, grid)] = {
        "launches": len(c["SQ_INSTS_VALU"]), "kernel_cycles": round(cyc), "us_at_2.4GHz": round(cyc / 2400.0, 1),
        "valu_insts_per_launch": round(m["SQ_INSTS_VALU"]), "lds_insts_per_launch": round(m.get("SQ_INSTS_LDS", 0)),
        "valu_insts_per_simd": round(ips), "valu_issue_frac": round(ips * 4.0 / cyc, 3),
        "waves_resident_avg_per_simd": round(m["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc), 2)}
json.dump({"note": __doc__.split("usage")[0].strip(), "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
