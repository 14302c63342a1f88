"""MFMA utilisation per kernel from tools/pmc_summary.py output (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  tflops      = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 flop / (kernel cycles / 2.4 GHz)
  wait_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in s_waitcnt / barrier)
usage: mfma_util_from_pmc.py <pmc_summary.txt> <out.json>"""
import json, re, sys
cur, rows = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
    if m:
        cur = m.group(1).strip(); rows[cur] = {"dispatches": int(m.group(2))}
        continue
    m = re.match(r"^\s+(\w+)\s+([\d.]+)\s+per-dispatch\s+([\d.]+)", line)
    if m and cur:
        rows[cur][m.group(1)] = float(m.group(3))
out = {}
for k, r in rows.items():
    if r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    cyc = r["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = {"dispatches": r["dispatches"], "kernel_cycles": round(cyc), "us_at_2.4GHz": round(cyc / 2400.0, 1),
              "mfma_util": round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4),
              "tflops_from_mops": round(r["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (cyc / 2.4e9) / 1e12, 1),
              "wait_frac": round(r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], 3)}
json.dump({"note": __doc__.split("usage")[0].strip(), "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["dispatches"]):
    print("%-44s n=%-4d %8.1f us  MFMA util %5.1f %%  %6.1f TF/s  waiting %4.1f %%" %
          (k[:44], v["dispatches"], v["us_at_2.4GHz"], 100 * v["mfma_util"], v["tflops_from_mops"], 100 * v["wait_frac"]))
