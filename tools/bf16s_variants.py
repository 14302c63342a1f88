"""Every LDS-DMA variant of the bf16-storage rows GEMM (LIDBOX_GEMM16S_DMA=bm,bn,stages; 0 = register-staged kernel) on the
rows launches of the x-vector step at B utterances, as the step issues them: shadow-only output (+ fp32 for the layers that
keep it), bf16 ReLU mask on the dgrads.  One process per variant (the override is read per call, but the kernels' LDS
attributes are set once).  usage: python tools/bf16s_variants.py [B=256]  ->  table of us per call (medians of 7 x 5)"""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["policy", "64,128,2", "128,128,2", "256,256,2", "256,256,1", "256,128,2", "256,128,1"]
if os.environ.get("BF16S_VARIANTS"):
    VARIANTS = os.environ["BF16S_VARIANTS"].split(";")
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch
    from lidbox_amd import _native as nv
    B = int(sys.argv[2])
    # (name, M, K, N, fp32 output too, mask)
    CALLS = [("frame1 fwd", 198 * B, 200, 512, False, False), ("frame2 fwd", 99 * B, 1536, 512, False, False), ("frame3 fwd", 33 * B, 1536, 512, False, False),
             ("frame4 fwd", 33 * B, 512, 512, False, False), ("frame5 fwd", 33 * B, 512, 1504, True, False),
             ("frame5 dgrad", 33 * B, 1504, 512, False, True), ("frame4 dgrad", 33 * B, 512, 512, False, True), ("frame3 dgrad", 33 * B, 512, 1536, False, True),
             ("frame2 dgrad r0", 99 * B, 1024, 512, False, True), ("frame2 dgrad r1", 99 * B, 512, 512, False, True)]
    if os.environ.get("BF16S_FRAME5_SHADOW_ONLY", "1") == "1":      # the all-shadow mode's pooling reads frame5's shadow (round 5)
        CALLS = [(n, M, K, N, False, m) for n, M, K, N, _, m in CALLS]
    if os.environ.get("BF16S_CALLS"):
        CALLS = [c for c in CALLS if c[0] in os.environ["BF16S_CALLS"].split(";")]
    st = nv.current_stream()
    for name, M, K, N, keep32, mask in CALLS:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        c32 = torch.empty(M, N, device="cuda") if keep32 else None
        c16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        mk = torch.randn(M, N, device="cuda").bfloat16() if mask else None
        bias = torch.randn(N, device="cuda")
        wsb = max(16, nv.lib.lidbox_gemm_bf16_rows_workspace(M, N, K)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        ra = nv.Rows(a.data_ptr(), 0, K, 1, M); rc = nv.Rows(c32.data_ptr() if keep32 else None, 0, N, 1, M)
        epi, aux = (nv.EPI_RELU_MASK | nv.EPI_MASK_BF16, nv.ptr(mk)) if mask else (nv.EPI_BIAS_RELU, nv.ptr(bias))
        f = lambda: nv.check(nv.lib.lidbox_gemm_bf16s_nt(ra, nv.ptr(b), K, rc, nv.ptr(c16), K, N, epi, aux, nv.ptr(ws), wsb, st))
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        print("%s|%.1f" % (name, statistics.median(ts)), flush=True)
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "256"
table = {}
for v in VARIANTS:
    env = dict(os.environ)
    env.pop("LIDBOX_GEMM16S_DMA", None)
    if v != "policy":
        env["LIDBOX_GEMM16S_DMA"] = v
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", B], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if "|" in l:
            n, t = l.split("|"); table.setdefault(n, {})[v] = float(t)
print("%-18s" % "call" + "".join("%11s" % v for v in VARIANTS))
for n, row in table.items():
    print("%-18s" % n + "".join("%11.1f" % row.get(v, -1) for v in VARIANTS))
print("%-18s" % "sum" + "".join("%11.1f" % sum(r.get(v, 0) for r in table.values()) for v in VARIANTS))
