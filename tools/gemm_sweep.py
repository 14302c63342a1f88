"""Brute-force sweep of the fp32 GEMM decompositions (tile shape x split count) for every x-vector GEMM launch of one
train step, against the cost model's own choice (lidbox_amd/csrc/gemm.hip: choose_rows / tn_plan).  Uses the tuning
overrides LIDBOX_GEMM_PLAN / LIDBOX_GEMM_TN_PLAN / LIDBOX_GEMM_NO_TAIL_SPLIT.  Measure, don't guess: the winners that
beat the model by a margin go into the tuned table in gemm.hip.
usage: python tools/gemm_sweep.py [B]"""
import ctypes as C
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidbox_amd import _native as nv

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_pos[0]) if _pos else 256
CNN = "--cnn" in sys.argv          # lidbox.models.cnn (BASELINE configs[3]) instead of the x-vector
REPS = 10


def rows(t, bs, rs, batch, rpb, off=0):
    return nv.Rows(t.data_ptr() + 4 * off, bs, rs, batch, rpb)


def timeit(fn, reps=REPS):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def setenv(**kw):
    for k in ("LIDBOX_GEMM_PLAN", "LIDBOX_GEMM_TN_PLAN", "LIDBOX_GEMM_NO_TAIL_SPLIT"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = v


def main():
    st = nv.current_stream()
    big = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    cases = []   # (name, kind, M, N, K, flops, fn(ws_ptr, ws_bytes))
    keep = []
    layers = [("frame1", 198, 40, 5, 1, 512), ("frame2", 198, 512, 3, 2, 512), ("frame3", 99, 512, 3, 3, 512),
              ("frame4", 33, 512, 1, 1, 512), ("frame5", 33, 512, 1, 1, 1500)]
    denses = [("segment1", 3000, 512), ("segment2", 512, 512), ("outputs", 512, 4)]
    if CNN:
        layers = [("conv_1", 198, 12, 5, 1, 500), ("conv_2", 198, 500, 7, 2, 500), ("conv_3", 99, 500, 1, 1, 500),
                  ("conv_4", 99, 500, 1, 1, 3000)]
        denses = [("fc_1", 3000, 1500), ("fc_2", 1500, 600), ("output", 600, 4)]
    for name, T, Cc, k, s, Co in layers:
        To, Tp = (T - 1) // s + 1, T + k - 1
        x = torch.randn(B, Tp, Cc, device="cuda"); W = torch.randn(k * Cc, Co, device="cuda") * 0.05
        bias = torch.randn(Co, device="cuda"); y = torch.zeros(B, To, Co, device="cuda")
        dy = torch.randn(B, To, Co, device="cuda"); dx = torch.zeros(B, Tp, Cc, device="cuda")
        dW = torch.zeros(k * Cc, Co, device="cuda")
        keep += [x, W, bias, y, dy, dx, dW]
        M, K = B * To, k * Cc
        A, Y, DY = rows(x, Tp * Cc, s * Cc, B, To), rows(y, To * Co, Co, B, To), rows(dy, To * Co, Co, B, To)
        cases.append((name + " fwd", 0, M, Co, K, 2.0 * M * K * Co,
                      lambda wp, wn, A=A, W=W, Co=Co, Y=Y, K=K, bias=bias: nv.lib.lidbox_gemm_nn(A, nv.ptr(W), Co, Y, K, Co, nv.EPI_BIAS_RELU, nv.ptr(bias), wp, wn, st)))
        cases.append((name + " wgrad", 2, M, Co, K, 2.0 * M * K * Co,
                      lambda wp, wn, A=A, DY=DY, dW=dW, Co=Co, K=K, bias=bias: nv.lib.lidbox_gemm_tn(A, DY, nv.ptr(dW), Co, K, Co, 0, nv.ptr(bias), wp, wn, st)))
        if name not in ("frame1", "conv_1"):
            for g in range((k + s - 1) // s):
                nt = min(s, k - g * s)
                Cd = rows(dx, Tp * Cc, s * Cc, B, To, off=g * s * Cc)
                Wg = C.c_void_p(W.data_ptr() + 4 * g * s * Cc * Co)
                mask = C.c_void_p(x.data_ptr() + 4 * g * s * Cc)
                epi = nv.EPI_RELU_MASK if g == 0 else nv.EPI_ACCUM_RELU_MASK
                cases.append(("%s dgrad%d" % (name, g), 1, M, nt * Cc, Co, 2.0 * M * Co * nt * Cc,
                              lambda wp, wn, DY=DY, Wg=Wg, Co=Co, Cd=Cd, n=nt * Cc, epi=epi, mask=mask: nv.lib.lidbox_gemm_nt(DY, Wg, Co, Cd, Co, n, epi, mask, wp, wn, st)))
    for name, K, N in denses:
        x = torch.randn(B, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda"); y = torch.zeros(B, N, device="cuda")
        dy = torch.randn(B, N, device="cuda"); dx = torch.zeros(B, K, device="cuda"); dW = torch.zeros(K, N, device="cuda")
        keep += [x, W, bias, y, dy, dx, dW]
        A, Y, DY, DX = rows(x, 0, K, 1, B), rows(y, 0, N, 1, B), rows(dy, 0, N, 1, B), rows(dx, 0, K, 1, B)
        fl = 2.0 * B * K * N
        cases.append((name + " fwd", 0, B, N, K, fl, lambda wp, wn, A=A, W=W, N=N, Y=Y, K=K, bias=bias: nv.lib.lidbox_gemm_nn(A, nv.ptr(W), N, Y, K, N, nv.EPI_BIAS_RELU, nv.ptr(bias), wp, wn, st)))
        cases.append((name + " wgrad", 2, B, N, K, fl, lambda wp, wn, A=A, DY=DY, dW=dW, N=N, K=K, bias=bias: nv.lib.lidbox_gemm_tn(A, DY, nv.ptr(dW), N, K, N, 0, nv.ptr(bias), wp, wn, st)))
        cases.append((name + " dgrad", 1, B, K, N, fl, lambda wp, wn, DY=DY, W=W, N=N, DX=DX, K=K, x=x: nv.lib.lidbox_gemm_nt(DY, nv.ptr(W), N, DX, N, K, nv.EPI_RELU_MASK, nv.ptr(x), wp, wn, st)))

    tot_def = tot_best = 0.0
    table = []
    for cname, kind, M, N, K, fl, fn in cases:
        out = (C.c_int * 4)()
        variants = [("model", {})]
        if kind == 2:
            for bm in (128, 64):
                for bn in (128, 64):
                    tiles = -(-K // bm) * -(-N // bn)
                    seen = set()
                    for target in (256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072):
                        sp = max(1, min(target // tiles, M // 64))
                        if sp in seen:
                            continue
                        seen.add(sp)
                        variants.append(("%d,%d,%d" % (bm, bn, sp), {"LIDBOX_GEMM_TN_PLAN": "%d,%d,%d" % (bm, bn, sp)}))
        else:
            for bm in (128, 64):
                for bn in (128, 64):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                        if sp > 1 and (sp * M * N * 4 > big.numel() or K // sp < 64):
                            continue
                        p = "%d,%d,%d" % (bm, bn, sp)
                        variants.append((p, {"LIDBOX_GEMM_PLAN": p}))
                        if sp == 1:
                            variants.append((p + ",notail", {"LIDBOX_GEMM_PLAN": p, "LIDBOX_GEMM_NO_TAIL_SPLIT": "1"}))
                        if sp == 1 and bm == 128 and "--waves8" in sys.argv:      # 8-wave variant of the 128-row tiles
                            variants.append((p + ",8", {"LIDBOX_GEMM_PLAN": p + ",8"}))
                            variants.append((p + ",8,notail", {"LIDBOX_GEMM_PLAN": p + ",8", "LIDBOX_GEMM_NO_TAIL_SPLIT": "1"}))
        res = []
        for vname, env in variants:
            setenv(**env)
            if kind == 2:
                wsb = nv.lib.lidbox_gemm_tn_workspace(M, K, N)
            else:                                  # the engine hands every launch its one large workspace (tdnn.py)
                wsb = big.numel()
            if wsb > big.numel():
                continue
            wp = nv.ptr(big)
            if fn(wp, max(wsb, 16)) != 0:
                continue
            torch.cuda.synchronize()
            res.append((timeit(lambda: fn(wp, max(wsb, 16))), vname, env, wsb))
        # re-time the model's choice and the three fastest, interleaved
        model = [r for r in res if r[1] == "model"]
        top = sorted(res)[:3] + model
        t = {r[1]: [] for r in top}
        for _ in range(5):
            for _, vname, env, wsb in top:
                setenv(**env)
                t[vname].append(timeit(lambda: fn(nv.ptr(big), max(wsb, 16))))
        med = {k: statistics.median(v) for k, v in t.items()}
        best = min(med, key=med.get)
        setenv()
        nv.lib.lidbox_gemm_plan_query(kind, M, N, K, big.numel() if kind != 2 else 0, out)
        tot_def += med["model"]; tot_best += med[best]
        print("%-15s kind=%d M=%6d N=%5d K=%5d  model(%d,%d,%d) %7.1f us %5.1f TF | best %-16s %7.1f us %5.1f TF  (%+.1f%%)  | %s" %
              (cname, kind, M, N, K, out[0], out[1], out[2], med["model"], fl / med["model"] / 1e6, best, med[best],
               fl / med[best] / 1e6, 100 * (med[best] / med["model"] - 1),
               " ".join("%s:%.1f" % (k, v) for k, v in sorted(med.items(), key=lambda kv: kv[1]))), flush=True)
        if best != "model" and med[best] < 0.98 * med["model"] and M > 256:     # small-M dense layers: noise-level differences
            f = best.split(",")
            nums = [t for t in f if t.isdigit()]
            table.append("    {%d, %d, %d, %d, %s, %s, %s, %d, %s},   // %s B=%d: %.1f -> %.1f us" %
                         (kind, M, N, K, nums[0], nums[1], nums[2], int("notail" in f), nums[3] if len(nums) > 3 else "4",
                          cname, B, med["model"], med[best]))
    print("TOTAL model %.1f us   best-of-sweep %.1f us" % (tot_def, tot_best))
    print("// tuned entries {kind, M, N, K, bm, bn, splits, no_tail_split, waves}:")
    print("\n".join(table))


if __name__ == "__main__":
    main()
