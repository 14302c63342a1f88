"""frame1's bf16-storage wgrad as the step issues it (implicit windows over [B, 4 + 198, 40], dY [B, 198, 512]) on the K1-resident kernel
(gemm16_tn_kres.h) and on the four-wave 128 x 128 tiles / the ping-pong tile, with its carried slice sum run as a launch of its own.
usage: python tools/tn_kres_time.py [B]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, C, k, Co = 198, 40, 5, 512
x16 = torch.zeros(B, k - 1 + T, C, device="cuda", dtype=torch.bfloat16)
x16[:, k - 1:] = torch.randn(B, T, C, device="cuda").bfloat16()
dy16 = torch.randn(B, T, Co, device="cuda").bfloat16()
M, K1 = B * T, k * C
ra = nv.Rows(x16.data_ptr(), (k - 1 + T) * C, C, B, T)
rb = nv.Rows(dy16.data_ptr(), T * Co, Co, B, T)
st = nv.current_stream()
res = {}
for name, env in (("four-wave 128 x 128", {"LIDBOX_GEMM16_TN_KRES": "0", "LIDBOX_GEMM16_TN_PP": "0"}),
                  ("ping-pong 256 x 256", {"LIDBOX_GEMM16_TN_KRES": "0", "LIDBOX_GEMM16_TN_PP": "1"}),
                  ("K1-resident", {"LIDBOX_GEMM16_TN_KRES": "1", "LIDBOX_GEMM16_TN_PP": "0"})):
    os.environ.update(env)
    wsb = nv.lib.lidbox_gemm_bf16s_tn_workspace(M, K1, Co)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dW = torch.empty(K1, Co, device="cuda"); db = torch.empty(Co, device="cuda")
    def call():
        nv.check(nv.lib.lidbox_gemm_bf16s_tn(ra, rb, nv.ptr(dW), Co, K1, Co, 0, nv.ptr(db), nv.ptr(ws), wsb, st))
    call(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    res[name] = (statistics.median(ts), dW.clone(), wsb)
ref = res["four-wave 128 x 128"][1]
for name, (us, dW, wsb) in res.items():
    print("B=%d  %-22s %7.1f us (GEMM + slice sum)  %6.1f TFLOP/s  workspace %5.1f MB  max |dW - four-wave| %.2e" %
          (B, name, us, 2.0 * M * K1 * Co / us / 1e6, wsb / 1e6, float((dW - ref).abs().max())))
