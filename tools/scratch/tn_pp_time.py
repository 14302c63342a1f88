"""bf16-storage wgrad: the 256 x 256 ping-pong tile (LIDBOX_GEMM16_TN_PP=1) against the four-wave 128 x 128 kernel (=0) on the
x-vector's wgrad shapes, as the step issues them (implicit conv rows, bias gradient).  usage: python tools/scratch/tn_pp_time.py [B=512]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
# (name, T_in, C, k, stride, Co)
LAYERS = [("frame1", 198, 40, 5, 1, 512), ("frame2", 198, 512, 3, 2, 512), ("frame3", 99, 512, 3, 3, 512), ("frame4", 33, 512, 1, 1, 512),
          ("frame5", 33, 512, 1, 1, 1504)]
if os.environ.get("TN_LAYERS"):
    LAYERS = [l for l in LAYERS if l[0] in os.environ["TN_LAYERS"].split(",")]
MODES = os.environ.get("TN_MODES", "0,1").split(",")
st = nv.current_stream()
def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    return statistics.median(ts)
tot = {}
for name, T, C, k, s, Co in LAYERS:
    pad = k - 1
    To = (T - 1) // s + 1
    x = torch.zeros(B, pad + T, C, device="cuda"); x[:, pad:] = torch.randn(B, T, C, device="cuda")
    x16 = x.bfloat16(); dy16 = torch.randn(B, To, Co, device="cuda").bfloat16()
    M, K1 = B * To, k * C
    ra = nv.Rows(x16.data_ptr(), (pad + T) * C, s * C, B, To); rb = nv.Rows(dy16.data_ptr(), To * Co, Co, B, To)
    res = {}
    line = "%-7s M=%6d K1=%5d N=%5d" % (name, M, K1, Co)
    for mode in MODES:
        os.environ["LIDBOX_GEMM16_TN_PP"] = mode
        wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dW = torch.empty(K1, Co, device="cuda"); db = torch.empty(Co, device="cuda")
        f = lambda: nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
        t = timeit(f)
        res[mode] = (dW.clone(), db.clone())
        tot[mode] = tot.get(mode, 0.0) + t
        line += "  pp=%s %7.1f us %6.1f TF (ws %5.1f MB)" % (mode, t, 2.0 * M * K1 * Co / t * 1e-6, wsb / 1e6)
    if len(res) == 2:
        dw = float((res["0"][0] - res["1"][0]).abs().max() / res["0"][0].abs().max())
        dbb = float((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max())
        line += "  |dW| %.1e |db| %.1e" % (dw, dbb)
    print(line, flush=True)
print("sum " + "  ".join("pp=%s %.1f us" % (m, tot[m]) for m in MODES) + " (GEMM + stand-alone slice sum)")
