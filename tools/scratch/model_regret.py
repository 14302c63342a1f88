"""how far the cost model's rows choice is from the swept best, from sweep logs (no GPU): usage model_regret.py sweep*.txt"""
import os, re, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LIDBOX_GEMM_NO_TUNED"] = "1"
from lidbox_amd import _native as nv
tot_m = tot_b = 0.0
for f in sys.argv[1:]:
    for line in open(f):
        m = re.match(r"(\S+ \S+)\s+kind=(\d) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+model\((\d+),(\d+),(\d+)\)\s+([\d.]+) us", line)
        if not m or m.group(2) == "2": continue
        kind, M, N, K = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
        if nv.lib.lidbox_gemm_plan_is_stream_k(kind, M, N, K, 768 << 20): continue
        times = dict((k, float(v)) for k, v in re.findall(r"(\d+,\d+,\d+(?:,notail)?|model):([\d.]+)", line.split("|")[-1]))
        out = (ctypes.c_int * 4)(); nv.lib.lidbox_gemm_plan_query(kind, M, N, K, 768 << 20, out)
        key = "%d,%d,%d" % (out[0], out[1], out[2])
        best = min(times.values())
        t = times.get(key, times.get(key + ",notail"))
        flag = "" if t is not None else "  (choice not among the top three: >= %.1f)" % max(v for k, v in times.items() if k != "model")
        if t is None: t = max(v for k, v in times.items() if k != "model")
        tot_m += t; tot_b += best
        if t > 1.02 * best: print("%-28s %-16s model->%s %.1f  best %.1f%s" % (os.path.basename(f), m.group(1), key, t, best, flag))
print("model total %.1f  best %.1f" % (tot_m, tot_b))
