"""Per-wave phase timeline of the fp32 rows kernels (a -DLBX_GEMM_TIMING build of gemm.hip, tools/ab_build.py).
usage: LIDBOX_HIP_LIB=tools/ab/libgtime.so python tools/gemm_phases.py M K N kind [bm,bn,splits]   kind: nn | nt | ntmask | tn
(tn: wgrad C[K,N] = A[M,K]^T B[M,N], contraction over the M rows; the plan string then goes to LIDBOX_GEMM_TN_PLAN)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidbox_amd import _native as nv

M, K, N = (int(v) for v in sys.argv[1:4])
kind = sys.argv[4]
if len(sys.argv) > 5:
    os.environ["LIDBOX_GEMM_TN_PLAN" if kind == "tn" else "LIDBOX_GEMM_PLAN"] = sys.argv[5]
lib = nv.lib
lib.lidbox_gemm_debug_set_stamps.restype = C.c_int
lib.lidbox_gemm_debug_set_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(10 * (1 << 19), dtype=torch.int64, device="cuda")
nv.check(lib.lidbox_gemm_debug_set_stamps(stamps.data_ptr()))
a = torch.randn(M, K, device="cuda")
b = torch.randn((M, N) if kind == "tn" else ((N, K) if kind != "nn" else (K, N)), device="cuda")
cw = torch.empty(K, N, device="cuda")
bg = torch.empty(N, device="cuda")
tws = torch.empty(max(16, lib.lidbox_gemm_tn_workspace(M, K, N)), dtype=torch.uint8, device="cuda")
c = torch.empty(M, N, device="cuda")
mask = torch.randn(M, N, device="cuda")
st = nv.current_stream()
ra, rc = nv.Rows(a.data_ptr(), 0, K, 1, M), nv.Rows(c.data_ptr(), 0, N, 1, M)
ws = torch.empty(max(16, lib.lidbox_gemm_rows_workspace(M, N, K)), dtype=torch.uint8, device="cuda")


def call():
    if kind == "tn":
        return lib.lidbox_gemm_tn(ra, nv.Rows(b.data_ptr(), 0, N, 1, M), nv.ptr(cw), N, K, N, 0, nv.ptr(bg), nv.ptr(tws), tws.numel(), st)
    if kind == "nn":
        return lib.lidbox_gemm_nn(ra, nv.ptr(b), N, rc, K, N, nv.EPI_BIAS_RELU, nv.ptr(c[0]), nv.ptr(ws), ws.numel(), st)
    epi, aux = (nv.EPI_RELU_MASK, nv.ptr(mask)) if kind == "ntmask" else (nv.EPI_NONE, None)
    return lib.lidbox_gemm_nt(ra, nv.ptr(b), K, rc, K, N, epi, aux, nv.ptr(ws), ws.numel(), st)


for _ in range(3):
    nv.check(call())
torch.cuda.synchronize()
stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); nv.check(call()); e1.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 10)
s = s[s[:, 0] != 0]
tick = 1.0       # s_memtime ticks = shader cycles (MI355X_MICROARCH); counters of different XCDs are not comparable, so only
                 # differences inside one wave are reported
print("M=%d K=%d N=%d %s plan=%s: %.1f us by events, %d waves stamped, nk=%d" % (M, K, N, kind, os.environ.get("LIDBOX_GEMM_PLAN"), e0.elapsed_time(e1) * 1e3, len(s), s[0, 9]))
def pct(x): return "min %8.0f  p10 %8.0f  med %8.0f  p90 %8.0f  max %8.0f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
print(" prologue (cycles)    ", pct((s[:, 1] - s[:, 0]) * tick))
print(" K loop (cycles)      ", pct((s[:, 2] - s[:, 1]) * tick))
print(" epilogue (cycles)    ", pct((s[:, 3] - s[:, 2]) * tick))
life = (s[:, 3] - s[:, 0]).astype(np.float64)
print(" wave lifetime        ", pct(life), " sum / (1024 SIMDs) = %.0f cycles of one wave slot" % (life.sum() / 1024))
loop = (s[:, 2] - s[:, 1]).astype(np.float64)
for name, col in (("mma", 4), ("load issue", 5), ("lds store+wait", 6), ("barrier", 7)):
    print("  %-16s %5.1f %% of the K loop  (%.0f cycles per K step)" % (name, 100 * s[:, col].sum() / loop.sum(), (s[:, col] / s[:, 9]).mean() * tick))
