"""Register / scratch / occupancy report of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py features.hip [name-filter] [-DFLAG ...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidbox_amd import build as b
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
flags = [a for a in sys.argv[2:] if a.startswith("-")]
r = subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, src), "-o", "/tmp/kres.o"],
                   capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|TotalSGPRs|LDS Size \[bytes/block\]): (.*?) \[-R", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, d in rows.items():
    if filt and filt not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)
    print("%-70s vgpr %3s agpr %3s sgpr %3s scratch %4s vspill %3s sspill %3s occ %s lds %s" % (
        short[:70], d.get("VGPRs"), d.get("AGPRs"), d.get("TotalSGPRs"), d.get("ScratchSize"), d.get("VGPRs Spill"), d.get("SGPRs Spill"),
        d.get("Occupancy"), d.get("LDS Size")))
