"""dense-head GEMM shapes (M = batch) of the bf16 family against a forced split count (LIDBOX_GEMM16_SPLITS / _TN_SPLITS):
one process per setting.  usage: python tools/head16_sweep.py [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from lidbox_amd import _native as nv
    B = int(sys.argv[2])
    def timeit(fn, n=40):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    out = []
    for kind, M, K, N in (("nn", B, 3000, 512), ("nn", B, 512, 512), ("nn", B, 512, 4), ("nt", B, 4, 512), ("nt", B, 512, 512),
                          ("nt", B, 512, 3000), ("tn", B, 512, 4), ("tn", B, 512, 512), ("tn", B, 3000, 512)):
        a = torch.randn(M, K, device="cuda")
        ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        if kind == "tn":
            b = torch.randn(M, N, device="cuda"); c = torch.empty(K, N, device="cuda"); bg = torch.empty(N, device="cuda")
            f = lambda: nv.check(nv.lib.lidbox_gemm_bf16_tn(nv.Rows(a.data_ptr(), 0, K, 1, M), nv.Rows(b.data_ptr(), 0, N, 1, M), nv.ptr(c), N, K, N, 0, nv.ptr(bg), nv.ptr(ws), ws.numel(), nv.current_stream()))
        else:
            b = torch.randn(K, N, device="cuda") if kind == "nn" else torch.randn(N, K, device="cuda")
            c = torch.empty(M, N, device="cuda")
            fn = nv.lib.lidbox_gemm_bf16_nn if kind == "nn" else nv.lib.lidbox_gemm_bf16_nt
            f = lambda: nv.check(fn(nv.Rows(a.data_ptr(), 0, K, 1, M), nv.ptr(b), N if kind == "nn" else K, nv.Rows(c.data_ptr(), 0, N, 1, M), K, N, 0, None, nv.ptr(ws), ws.numel(), nv.current_stream()))
        out.append("%6.1f" % timeit(f))
    print(" ".join(out))
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "256"
print("splits   nn3000x512 nn512x512 nn512x4 nt4x512 nt512x512 nt512x3000 | tn512x4 tn512x512 tn3000x512   (us per call incl. reduce, graph replay)")
for s in ("default", "1", "2", "4", "8", "16", "32", "64"):
    env = dict(os.environ)
    if s != "default":
        env["LIDBOX_GEMM16_SPLITS"] = s; env["LIDBOX_GEMM16_TN_SPLITS"] = s
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B], env=env, capture_output=True, text=True)
    print("%-8s %s" % (s, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
