"""
CPU-only checks (no GPU, no compute launches): the C-ABI library loads and exports every symbol
include/lidbox_hip.h declares, the host-side entry points agree with the oracle and with the
reference's own test grid, the product path refuses to run without a HIP device, and the
data-parallel host logic works across two gloo ranks.
"""
import os
import re
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from oracle import features_np as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nv():
    from lidbox_amd import build
    build.build(verbose=False)              # hipcc cross-compiles for gfx950 without a GPU
    from lidbox_amd import _native
    return _native


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lidbox_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lidbox_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(nv):
    names = _declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(nv.lib, n), "missing export: " + n
    # and the ctypes table covers the header (no declared function left unbound)
    assert set(names) == set(nv._SIGS), (set(names) ^ set(nv._SIGS))
    assert nv.lib.lidbox_hip_abi_version() == nv.ABI_VERSION


def test_ms_to_frames_reference_grid(nv):
    """reference tests/test_features_audio.py:125-129"""
    for sr in range(1000, 60000, 1000):
        for ms in range(1, 5000, 100):
            assert nv.lib.lidbox_ms_to_frames(sr, ms) == (sr // 1000) * ms
    assert nv.lib.lidbox_num_frames(32000, 400, 160) == 198
    assert nv.lib.lidbox_num_frames(399, 400, 160) == 0
    assert nv.lib.lidbox_num_frames(48000, 400, 160) == 298


def test_host_mel_matrix_and_window_match_oracle(nv):
    from lidbox_amd.features import mel_ops
    for (M, F, sr, lo, hi) in [(40, 257, 16000, 0.0, 8000.0), (20, 129, 8000, 125.0, 3800.0), (85, 257, 16000, 0.0, 8000.0),
                               (10, 513, 44100, 20.0, 11000.0)]:
        W = mel_ops.linear_to_mel_weight_matrix_host(M, F, sr, lo, hi)
        ref = fo.linear_to_mel_weight_matrix(M, F, sr, lo, hi)
        assert W.shape == (F, M) and (W[0] == 0).all()
        assert np.abs(W - ref).max() < 5e-5
        assert ((W > 0).sum(axis=1) <= 2).all()
    W = mel_ops.linear_to_mel_weight_matrix_host(40, 257, 16000, 0.0, 8000.0)
    assert np.count_nonzero(W) == 464                                  # lidbox's non-endpoint linspace
    for L in (1, 2, 400, 401, 1024):
        w = np.zeros(L, np.float32)
        nv.check(nv.lib.lidbox_hann_window(L, w.ctypes.data))
        assert np.abs(w - fo.hann_window(L)).max() < 1e-6


def test_invalid_arguments_are_reported_not_thrown(nv):
    assert nv.lib.lidbox_mel_weight_matrix(0, 257, 16000, 0.0, 8000.0, None) == -1
    assert "lidbox_mel_weight_matrix" in nv.last_error()
    with pytest.raises(ValueError):
        nv.check(nv.lib.lidbox_hann_window(0, None))
    assert nv.lib.lidbox_gemm_tn_workspace(0, 5, 5) == 0
    assert nv.lib.lidbox_gemm_tn_workspace(50688, 200, 512) > 0
    assert nv.lib.lidbox_gemm_rows_workspace(256, 512, 3000) > 0          # small-M dense layer splits along K
    assert nv.lib.lidbox_gemm_rows_workspace(50688, 512, 200) == 0        # plenty of tiles: no split


def test_product_path_fails_loudly_without_gpu(nv):
    """no CPU fallback: CPU tensors / missing device raise instead of silently computing elsewhere"""
    import lidbox_amd.features as F
    from lidbox_amd.data import tf_utils
    from lidbox_amd.features import audio
    x = torch.zeros(2, 1000)
    with pytest.raises(Exception) as e:
        F.cmvn(torch.zeros(2, 5, 3))
    assert "HIP" in str(e.value) or "cuda" in str(e.value).lower()
    with pytest.raises(Exception):
        audio.linear_to_mel(torch.zeros(1, 4, 257), 16000)
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            tf_utils.extract_features(x, [16000, 16000], "logmelspectrogram")
        from lidbox_amd.models import xvector
        with pytest.raises(Exception):
            xvector.create((198, 40), 4)
    # nothing under lidbox_amd/ imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lidbox_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_shard_bounds_partition():
    from lidbox_amd.train import shard_bounds
    for B in (1, 7, 256, 2048, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(2048, 3, 8) == (768, 1024)


def test_synthetic_batch_is_deterministic():
    from lidbox_amd.testutil import synthetic_batch
    a, ya = synthetic_batch(5, 4)
    b, yb = synthetic_batch(5, 4)
    assert np.array_equal(a, b) and np.array_equal(ya, yb)
    assert a.shape == (5, 32000) and a.dtype == np.float32
    assert np.allclose(np.abs(a).max(axis=1), 10 ** (-3 / 20), rtol=1e-6)     # -3 dBFS peak


_DP_WORKER = textwrap.dedent("""
    import os, sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from lidbox_amd.train import GradSync, init_distributed, shard_bounds

    rank, world, _ = init_distributed(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    torch.manual_seed(0)
    n = 1003
    # every rank computes the same per-utterance "gradients"; each keeps the SUM over its shard scaled by
    # 1/B_local (what a rank's backward produces for a mean-over-local-batch loss)
    B = 10
    per_utt = torch.randn(B, n, dtype=torch.float64)
    lo, hi = shard_bounds(B, rank, world)
    flat = per_utt[lo:hi].sum(0) / (hi - lo)
    sync = GradSync(flat, [0, 250, n])                   # two buckets, high one first
    assert sync.active and sync.world == 2 and sync.num_buckets == 2
    sync.launch(1)
    sync.launch(0)
    sync.wait()
    averaged = flat * sync.grad_scale
    expect = per_utt.mean(0)                             # single-device mean over the global batch
    err = float((averaged - expect).abs().max())
    assert err < 1e-12, err
    # a second step on the same buffer (buffers are reused every step)
    flat.copy_(per_utt[lo:hi].sum(0) / (hi - lo))
    sync.launch(1); sync.launch(0); sync.wait()
    assert float((flat * sync.grad_scale - expect).abs().max()) < 1e-12
    # uneven shards (11 utterances -> 6 + 5): every rank scales by 1 / GLOBAL batch (what Trainer._loss_scale does),
    # the all-reduced sum then is the global mean's gradient with no further scaling
    B2 = 11
    per2 = torch.randn(B2, n, dtype=torch.float64)
    lo2, hi2 = shard_bounds(B2, rank, world)
    assert (hi2 - lo2) == (6 if rank == 0 else 5)
    cnt = torch.tensor([hi2 - lo2]); dist.all_reduce(cnt)
    assert int(cnt) == B2
    flat2 = per2[lo2:hi2].sum(0) / int(cnt)
    s2 = GradSync(flat2, [0, 250, n])
    s2.launch(1); s2.launch(0); s2.wait()
    assert float((flat2 - per2.mean(0)).abs().max()) < 1e-12
    # bf16 wire format (Trainer(grad_wire_dtype="bfloat16")): every bucket is rounded once, summed as bf16 by the
    # collective, widened back into the fp32 buffer -- exactly bf16(bf16(g0) + bf16(g1)), within 2^-7 of the fp32 sum
    per3 = torch.randn(B, n, dtype=torch.float32)
    mine = (per3[lo:hi].sum(0) / B).contiguous()
    other = (torch.cat([per3[:lo], per3[hi:]]).sum(0) / B).contiguous()
    flat3 = mine.clone()
    s3 = GradSync(flat3, [0, 252, n], wire_dtype="bfloat16")
    assert s3.wire_bytes == 2 * n and GradSync(flat3.clone(), [0, n]).wire_bytes == 4 * n
    s3.launch(1); s3.launch(0); s3.wait()
    expect3 = (mine.to(torch.bfloat16).float() + other.to(torch.bfloat16).float()).to(torch.bfloat16).float()
    assert flat3.dtype == torch.float32 and torch.equal(flat3, expect3)
    ref3 = per3.mean(0)
    assert float((flat3 - ref3).abs().max()) <= 2.0 ** -7 * float(ref3.abs().max()) + 1e-6
    # C_avg-style counters: all-reduce(sum) of integer-valued float counters is exact
    c = torch.full((7,), float(rank + 1))
    dist.all_reduce(c)
    assert float(c[0]) == 3.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_data_parallel_grad_sync_two_gloo_ranks(tmp_path, nv):
    """the N > 1 path on CPU: 2 processes, gloo, bucketed all-reduce == single-device mean gradient"""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "ok" in out


_DP8_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from lidbox_amd.metrics import SparseAverageDetectionCost
    from lidbox_amd.train import GradSync, init_distributed, plan_buckets, shard_bounds
    from oracle import model_np as mo

    rank, world, _ = init_distributed(backend="gloo")
    assert world == 8 and dist.get_backend() == "gloo"

    class Conv:
        def __init__(self, name):
            self.name = name

    class XVector:        # layout of lidbox.models.xvector (4 languages): what plan_buckets needs, no device
        convs = [Conv("frame%%d" %% i) for i in range(1, 6)]
        layout = {"frame1.W": (0, (5, 40, 512)), "frame2.W": (102912, (3, 512, 512)), "frame3.W": (889856, (3, 512, 512)),
                  "frame4.W": (1676800, (1, 512, 512)), "frame5.W": (1939456, (1, 512, 1500))}
        num_flat = 4510176
    bounds, split = plan_buckets(XVector(), 3)
    assert bounds == [0, 102912, 889856, 4510176]
    n = XVector.num_flat

    # BASELINE configs[2] / [4]: 8 ranks, uneven shards (2051 = 3 x 257 + 5 x 256).  A rank's backward leaves
    # sum_{b in shard} g_b / GLOBAL batch in its flat buffer (Trainer._loss_scale); the bucketed all-reduce(sum), launched from the
    # top bucket down as the backward pass does, must leave the global-batch mean gradient on every rank.
    B = 2051
    lo, hi = shard_bounds(B, rank, world)
    assert hi - lo == (257 if rank < 3 else 256)
    g = torch.Generator().manual_seed(1234)
    base = torch.randn(n, generator=g, dtype=torch.float32)                  # the same on every rank
    coef = torch.randn(B, generator=g, dtype=torch.float64)                  # utterance b's gradient = coef[b] * base (rank-1: cheap and exact to reason about)
    mine = (base.double() * (coef[lo:hi].sum() / B)).float()
    expect = base.double() * coef.mean()
    flat = mine.clone()
    sync = GradSync(flat, bounds)
    assert sync.active and sync.world == 8 and sync.num_buckets == 3 and sync.wire_bytes == 4 * n
    for i in (2, 1, 0):
        sync.launch(i)
    sync.wait()
    tol = 8 * 2.0 ** -24 * float(expect.abs().max()) * 4
    assert float((flat.double() - expect).abs().max()) <= tol
    # every rank holds the same bits (the collective's result does not depend on the rank)
    probe = flat[::4099].clone()
    ref = probe.clone(); dist.broadcast(ref, 0)
    assert torch.equal(probe, ref)

    # bf16 wire format at world 8: each rank rounds once, the collective sums in bf16 (world - 1 more roundings), the result is
    # widened to fp32: within (world + 1) half-ulps of bf16 of the fp32 sum, and again identical on every rank
    flat16 = mine.clone()
    s16 = GradSync(flat16, bounds, wire_dtype="bfloat16")
    assert s16.wire_bytes == 2 * n
    for i in (2, 1, 0):
        s16.launch(i)
    s16.wait()
    assert flat16.dtype == torch.float32
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    bound = sum(p.abs().double() for p in parts) * (world + 1) * 2.0 ** -9 + 1e-30
    assert bool(((flat16.double() - expect).abs() <= bound).all())
    ref16 = flat16[::4099].clone(); dist.broadcast(ref16, 0)
    assert torch.equal(flat16[::4099], ref16)
    try:
        GradSync(mine.clone(), bounds, wire_dtype="float16")
        raise SystemExit("an unknown wire format must be rejected")
    except ValueError:
        pass

    # C_avg at configs[4]'s world size: every rank counts its shard (here with the oracle's counting, the HIP kernel needs a GPU),
    # sync_counters() all-reduces the four state tensors, the result equals one process over the whole batch -- exactly
    N, Th, Bm = 10, 13, 803
    rng = np.random.default_rng(7)
    scores = rng.standard_normal((Bm, N)).astype(np.float32)
    labels = rng.integers(0, N, size=Bm)
    th = np.linspace(-2.0, 2.0, Th).astype(np.float32)
    lo, hi = shard_bounds(Bm, rank, world)
    part = mo.SparseAverageDetectionCost(N, th)
    part.update_state(labels[lo:hi], scores[lo:hi])
    m = SparseAverageDetectionCost(N, th, device="cpu")
    for dst, src in zip(m.counters(), (part.tp, part.fn, part.fp_pairs, part.tn_pairs)):
        dst.copy_(torch.from_numpy(src))
    m.sync_counters()
    whole = mo.SparseAverageDetectionCost(N, th)
    whole.update_state(labels, scores)
    for got, want in zip(m.counters(), (whole.tp, whole.fn, whole.fp_pairs, whole.tn_pairs)):
        assert np.array_equal(got.numpy(), want)
    synced = mo.SparseAverageDetectionCost(N, th)
    synced.tp, synced.fn, synced.fp_pairs, synced.tn_pairs = [c.numpy() for c in m.counters()]
    assert synced.result() == whole.result()
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_data_parallel_at_baseline_world_size_eight_gloo_ranks(tmp_path, nv):
    """BASELINE configs[2] / [4] run on 8 GPUs; no box this build saw had more than one.  What can be checked without them: the
    bucket plan of the real x-vector layout, uneven shards, the fp32 and bf16 wire formats and the C_avg counter exchange across
    EIGHT gloo ranks on the host equal the single-process answer."""
    script = tmp_path / "dp8_worker.py"
    script.write_text(_DP8_WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "ok" in out


def test_bench_refuses_a_scaling_run_that_is_not_what_it_claims():
    """bench.py --gpus N prints a line only for N ranks over nccl with the gradient exchange captured in the step's graph (host logic:
    the condition is a function of what the run reports about itself)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok = bench.scaling_run_problem
    assert ok(8, 8, "nccl", True, "in_graph", False) is None
    assert ok(2, 2, "nccl", True, "eager", True) is None                      # --no-graph: the eager exchange is what was asked for
    for bad in ((8, 4, "nccl", True, "in_graph", False),                       # a communicator with fewer ranks
                (8, 8, "gloo", True, "in_graph", False),                       # not RCCL
                (8, 8, "nccl", False, "none", False),                          # no exchange at all
                (8, 8, "nccl", True, "segmented", False),                      # host-launched between graph segments
                (8, 8, "nccl", True, "eager", False)):
        msg = ok(*bad)
        assert msg and "--gpus 8" in msg and "expected 8 ranks" in msg


def test_bucket_plan_is_contiguous_partition():
    """plan_buckets needs no device: use a layout-only stand-in with the x-vector's shapes"""
    from lidbox_amd.train import plan_buckets

    class Conv:
        def __init__(self, name):
            self.name = name

    class FakeModel:
        convs = [Conv("frame%d" % i) for i in range(1, 6)]
        layout = {"frame1.W": (0, (5, 40, 512)), "frame2.W": (102912, (3, 512, 512)), "frame3.W": (889856, (3, 512, 512)),
                  "frame4.W": (1676800, (1, 512, 512)), "frame5.W": (1939456, (1, 512, 1500))}
        num_flat = 4510176
    bounds, split = plan_buckets(FakeModel(), 2)
    assert bounds[0] == 0 and bounds[-1] == 4510176 and len(bounds) == 3
    assert bounds[1] == 889856 and split == 2          # frame1+frame2 form the late (small) bucket
    assert plan_buckets(FakeModel(), 1) == ([0, 4510176], None)
    # three buckets: frame1 | frame2 | everything above; more than the conv boundaries allow degrades gracefully
    assert plan_buckets(FakeModel(), 3) == ([0, 102912, 889856, 4510176], [1, 2])
    b9, s9 = plan_buckets(FakeModel(), 9)
    assert s9 == [1, 2] and b9 == [0, 102912, 889856, 4510176]


def test_host_signal_chunk_plan_matches_oracle(nv):
    """lidbox_signal_chunk_plan (host-only entry point; reference steps.py:586-588,604-614) against the oracle's
    restatement over a grid of lengths / sample rates / chunkings, and the vectorised count used by signal_ops"""
    import ctypes
    from oracle import signal_np as so
    from lidbox_amd.features import signal_ops as sg
    rng = np.random.default_rng(0)
    out = (ctypes.c_long * 4)()
    for sr in (8000, 16000, 22050, 44100):
        for length_ms, step_ms, pad_ms in ((1000, 500, 0), (1000, 500, 400), (2000, 1500, 500), (250, 100, 250),
                                           (10, 25, 10), (3, 1, 0), (30, 30, 29)):
            ns = [0, 1, 2] + [int(v) for v in rng.integers(0, 5 * sr, size=20)]
            L, S, _, _ = so.signal_chunk_plan(0, sr, length_ms, step_ms, pad_ms)
            ns += [L - 1, L, L + 1, L + S - 1, L + S, 3 * L]
            P = int(np.float32(sr) * np.float32(1e-3 * pad_ms))
            counts = sg.chunk_counts(np.array(ns), L, S, P)
            for n, c in zip(ns, counts):
                nv.check(nv.lib.lidbox_signal_chunk_plan(n, sr, length_ms, step_ms, pad_ms, out))
                ref = so.signal_chunk_plan(n, sr, length_ms, step_ms, pad_ms)
                assert tuple(out) == ref, (sr, length_ms, step_ms, pad_ms, n)
                assert c == ref[3]
    with pytest.raises(ValueError):
        nv.check(nv.lib.lidbox_signal_chunk_plan(100, 16000, 0, 10, 0, out))      # zero-length chunks


def test_wav_header_parser_is_host_logic():
    """audio.parse_wav_pcm16 (the host half of read_wav, reference audio.py:17-19) on the reference's WAV fixtures: the samples
    the oracle's reader sees, as int16"""
    import glob
    import os
    from oracle import features_np as fo
    from lidbox_amd.features import audio
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "audio", "*.wav")))
    assert paths
    for p in paths:
        pcm, nch, rate = audio.parse_wav_pcm16(open(p, "rb").read())
        sig, r = fo.read_wav_pcm16(p)
        assert rate == r and pcm.dtype == np.int16
        assert np.array_equal((pcm.astype(np.float32) / np.float32(32768.0)).reshape(-1, nch).mean(axis=1, dtype=np.float32), sig)
    with pytest.raises(ValueError):
        audio.parse_wav_pcm16(b"RIFFxxxxWAVE")


def test_gemm_plans_tuned_table_and_model(nv):
    """lidbox_gemm_plan_query is host logic: shapes listed in csrc/gemm_tuned.h (measured by tools/gemm_sweep.py) return
    the measured decomposition, every other shape the cost model's; the workspace queries cover whatever is chosen."""
    import ctypes
    import re
    src = open(os.path.join(os.path.dirname(nv.__file__), "csrc", "gemm_tuned.h")).read()
    entries = [tuple(int(v) for v in m.groups()) for m in
               re.finditer(r"^\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", src, re.M)]
    assert len(entries) >= 10
    out = (ctypes.c_int * 4)()
    for kind, M, N, K, bm, bn, splits, _, waves in entries:
        nv.check(nv.lib.lidbox_gemm_plan_query(kind, M, N, K, 1 << 30, out))
        assert (out[0], out[1]) == (bm, bn), (kind, M, N, K)
        assert waves in (4, 8) and (kind == 2 or nv.lib.lidbox_gemm_plan_waves(kind, M, N, K, 1 << 30) == waves)
        if kind == 2:
            assert out[2] >= 1 and out[2] * out[3] >= M and (out[2] - 1) * out[3] < M       # slices cover the M rows
            assert nv.lib.lidbox_gemm_tn_workspace(M, K, N) >= (out[2] * K * N + out[2] * N) * 4
        else:
            assert out[2] == splits == 1 or out[2] * out[3] >= K
            if out[2] > 1:
                assert nv.lib.lidbox_gemm_rows_workspace(M, N, K) >= out[2] * M * N * 4
    # an unlisted shape goes through the model: valid tile, whole K covered
    for kind, M, N, K in ((0, 1000, 300, 77), (1, 5000, 64, 640), (2, 3000, 96, 200)):
        assert tuple(e[:4] for e in entries).count((kind, M, N, K)) == 0
        nv.check(nv.lib.lidbox_gemm_plan_query(kind, M, N, K, 1 << 26, out))
        assert out[0] in (64, 128) and out[1] in (64, 128) and out[2] >= 1
        assert out[2] * out[3] >= (M if kind == 2 else K)


def test_stream_k_dispatch_policy_is_host_logic(nv, monkeypatch):
    """which launches go to the persistent stream-K kernels (csrc/gemm_sk.h): forward GEMMs with K >= 1024 and wgrads, above
    a size floor, workspace permitting, unless the measured table (csrc/gemm_tuned.h) holds a faster classic decomposition
    for the shape; LIDBOX_GEMM_SK=0 / LIDBOX_GEMM_SK_ALL=1 move the policy (A/B and test aids)"""
    q = nv.lib.lidbox_gemm_plan_is_stream_k
    big = 1 << 30
    for var in ("LIDBOX_GEMM_SK", "LIDBOX_GEMM_SK_ALL", "LIDBOX_GEMM_SK_GRID", "LIDBOX_GEMM_SK_MIN_FLOP"):
        monkeypatch.delenv(var, raising=False)
    # x-vector layers at bs 256: frame2 forward (K = 1536) yes, frame3 forward no (4.125 tiles per CU: the table
    # lists 64 x 64 tiles for it), K = 512 / 200 forward no, dgrad no
    assert q(0, 25344, 512, 1536, big) == 1 and q(0, 8448, 512, 1536, big) == 0
    monkeypatch.setenv("LIDBOX_GEMM_NO_TUNED", "1")
    assert q(0, 8448, 512, 1536, big) == 1
    monkeypatch.delenv("LIDBOX_GEMM_NO_TUNED")
    assert q(0, 8448, 1500, 512, big) == 0 and q(0, 50688, 512, 200, big) == 0
    assert q(1, 8448, 512, 1500, big) == 0 and q(1, 25344, 1024, 512, big) == 0
    # wgrads (kind 2: K = K1) above 6 GFLOP yes, frame4's 4.4 GFLOP and the dense head no
    assert q(2, 25344, 512, 1536, big) == 1 and q(2, 50688, 512, 200, big) == 1 and q(2, 8448, 1500, 512, big) == 1
    assert q(2, 8448, 512, 512, big) == 0 and q(2, 256, 512, 3000, big) == 0
    # the workspace decides too, and the queries cover what the kernels need (counters + two slabs per workgroup)
    need = nv.lib.lidbox_gemm_rows_workspace(25344, 512, 1536)
    assert need >= 16384 + 768 * 2 * 128 * 128 * 4
    assert q(0, 25344, 512, 1536, need) == 1 and q(0, 25344, 512, 1536, need - 1) == 0
    assert q(2, 25344, 512, 1536, nv.lib.lidbox_gemm_tn_workspace(25344, 1536, 512)) == 1
    monkeypatch.setenv("LIDBOX_GEMM_SK", "0")
    assert q(0, 25344, 512, 1536, big) == 0 and q(2, 25344, 512, 1536, big) == 0
    monkeypatch.delenv("LIDBOX_GEMM_SK")
    monkeypatch.setenv("LIDBOX_GEMM_SK_ALL", "1")
    assert q(1, 25344, 1024, 512, big) == 1 and q(0, 8448, 1500, 512, big) == 1


def test_dgrad_wgrad_pair_policy_is_host_logic(nv, monkeypatch):
    """lidbox_gemm_nt_tn launches a layer's dgrad + wgrad as ONE kernel when both are small 64 x 64 LDS-DMA launches
    (csrc/gemm.hip: pair_plan): the dense head at any batch size the x-vector sees, frame4 at bs 256; chip-filling layers,
    stream-K shapes and LIDBOX_GEMM_NO_PAIR=1 fall back to the two calls"""
    q = nv.lib.lidbox_gemm_plan_is_pair
    big = 1 << 30
    for var in ("LIDBOX_GEMM_NO_PAIR", "LIDBOX_GEMM_PAIR_MAX_BLOCKS", "LIDBOX_GEMM_DMA", "LIDBOX_GEMM_PLAN", "LIDBOX_GEMM_TN_PLAN"):
        monkeypatch.delenv(var, raising=False)
    # (M, Co, N = rows of dX per window, K1): segment1 / segment2 at bs 256, frame4 (33 frames x 256 utterances)
    assert q(256, 512, 3000, 3000, big, big) == 1 and q(256, 512, 512, 512, big, big) == 1
    assert q(8448, 512, 512, 512, big, big) == 1
    # frame3 (dgrad N = 1536: 3168 + 1536 blocks) and frame2 are beyond two rounds of resident workgroups
    assert q(8448, 512, 1536, 1536, big, big) == 0 and q(25344, 512, 1024, 1536, big, big) == 0
    # the wgrad's partial sums need their workspace
    assert q(256, 512, 512, 512, big, 1024) == 0
    monkeypatch.setenv("LIDBOX_GEMM_NO_PAIR", "1")
    assert q(256, 512, 512, 512, big, big) == 0
    monkeypatch.delenv("LIDBOX_GEMM_NO_PAIR")
    monkeypatch.setenv("LIDBOX_GEMM_DMA", "0")
    assert q(256, 512, 512, 512, big, big) == 0


def test_lr_schedules_follow_keras(nv):
    """tf.keras.optimizers.schedules.ExponentialDecay / PiecewiseConstantDecay (reference keras_utils.py:137-139) as host
    functions of the optimizer step (Keras' 0-based `iterations`)"""
    from lidbox_amd.models.keras_utils import _optimizer_from_config, lr_schedule_from_config
    s = lr_schedule_from_config({"cls": "ExponentialDecay", "kwargs": {"initial_learning_rate": 0.01, "decay_steps": 100, "decay_rate": 0.5}})
    assert s(0) == pytest.approx(0.01) and s(100) == pytest.approx(0.005) and s(50) == pytest.approx(0.01 * 0.5 ** 0.5, rel=1e-6)
    st = lr_schedule_from_config({"cls": "ExponentialDecay",
                                  "kwargs": {"initial_learning_rate": 0.01, "decay_steps": 100, "decay_rate": 0.5, "staircase": True}})
    assert st(99) == pytest.approx(0.01) and st(100) == pytest.approx(0.005) and st(250) == pytest.approx(0.0025)
    pw = lr_schedule_from_config({"cls": "PiecewiseConstantDecay", "kwargs": {"boundaries": [10, 20], "values": [1.0, 0.5, 0.1]}})
    assert [pw(i) for i in (0, 10, 11, 20, 21, 1000)] == [1.0, 1.0, 0.5, 0.5, 0.1, 0.1]      # boundaries are inclusive on the left value
    opt = _optimizer_from_config({"cls": "Adam", "kwargs": {"beta_1": 0.8, "lr_scheduler": {"cls": "PiecewiseConstantDecay",
                                                                                          "kwargs": {"boundaries": [1], "values": [0.1, 0.01]}}}})
    assert opt["beta_1"] == 0.8 and opt["lr_schedule"](0) == 0.1 and opt["lr_schedule"](2) == 0.01
    with pytest.raises(ValueError):
        lr_schedule_from_config({"cls": "CosineDecay", "kwargs": {}})


def test_early_stopping_follows_keras_patience_semantics():
    """tf.keras.callbacks.EarlyStopping: stop once `patience` consecutive epochs brought no improvement (wait >= patience)"""
    from lidbox_amd.models.keras_utils import EarlyStopping

    class W:
        stop_training = False

    for patience, expect_epochs in ((0, 2), (1, 2), (2, 3), (3, 4)):
        cb, w = EarlyStopping(monitor="val_loss", patience=patience), W()
        n = 0
        for epoch, v in enumerate([1.0, 1.0, 1.0, 1.0, 1.0, 1.0]):
            cb.on_epoch_end(w, epoch, {"val_loss": v})
            n += 1
            if w.stop_training:
                break
        assert n == expect_epochs, (patience, n)
    cb, w = EarlyStopping(monitor="acc", patience=2, mode="max", min_delta=0.1), W()
    for epoch, v in enumerate([0.5, 0.55, 0.7, 0.75, 0.72]):       # +0.05 is below min_delta, +0.15 resets the wait
        cb.on_epoch_end(w, epoch, {"acc": v})
    assert w.stop_training and cb.best == 0.7
    import pytest
    with pytest.raises(ValueError):
        EarlyStopping(restore_best_weights=True)
    EarlyStopping(restore_best_weights=False, baseline=None, verbose=1)
