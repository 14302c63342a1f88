"""
Data-parallel equivalence on the GPU box: two ranks, each with half of the batch,
bucketed all-reduce + Adam(1/world)  ==  one process training on the whole batch.

The box has ONE GPU and RCCL refuses two ranks on the same device, so the two ranks share cuda:0
and exchange gradients over gloo (which stages CUDA tensors through the host).  Everything else
-- sharding, the three hipGraph segments, side-stream bucket launches, event ordering, Adam's
grad_scale -- is exactly the code path the 8-GPU RCCL run takes.
"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer, init_distributed, shard_bounds
    from lidbox_amd.metrics import SparseAverageDetectionCost

    use_graph = bool(int(sys.argv[1]))
    out_path = sys.argv[2]
    num_buckets = int(sys.argv[3])
    wire = sys.argv[4] if len(sys.argv) > 4 else None
    os.environ["LOCAL_RANK"] = "0"                       # both ranks on the only GPU
    rank, world, _ = init_distributed(backend="gloo")
    torch.cuda.set_device(0)
    B = 12
    sig, y = synthetic_batch(B, num_labels=4, duration_s=0.5)
    lo, hi = shard_bounds(B, rank, world)
    sd = torch.from_numpy(sig[lo:hi]).cuda()
    yd = torch.from_numpy(y[lo:hi].astype(np.int32)).cuda()
    model = xvector.create((48, 40), 4, seed=0)
    plan = audio.get_plan(16000, 400, 160)
    tr = Trainer(model, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=use_graph, num_buckets=num_buckets, grad_wire_dtype=wire)
    assert tr.sync.active == (world > 1) and tr.sync.num_buckets == num_buckets == tr.num_stages
    assert tr.sync.wire_bytes == model.num_flat * (2 if wire == "bfloat16" else 4)
    losses = []
    for _ in range(3):
        losses.append(float(tr.train_step(sd, yd)))
    torch.cuda.synchronize()
    # global mean loss = mean of the rank means (equal shard sizes)
    l = torch.tensor(losses, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(l)
        l /= world
    # C_avg over the GLOBAL batch: each rank counts its shard, one all-reduce(sum) of the counters (SURVEY 8e)
    metric = SparseAverageDetectionCost(4, np.linspace(-6.0, 0.0, 30))
    feats = plan.run(nv.FEAT_LOGMEL, sd)
    metric.update_state(yd, model(feats, training=False))
    metric.sync_counters()
    cavg, per_th = metric.result(return_per_threshold=True)
    if rank == 0:
        np.savez(out_path, flat=model.flat.cpu().numpy(), losses=l.numpy(), cavg=float(cavg), per_th=per_th.cpu().numpy(),
                 tp=metric.tp.cpu().numpy(), fn=metric.fn.cpu().numpy(), grad_sync=tr.grad_sync_mode)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(script, world, use_graph, out, num_buckets=2, wire=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), str(int(use_graph)), str(out), str(num_buckets)] + ([wire] if wire else []), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, o[-3000:])
    return np.load(out)


@pytest.mark.parametrize("use_graph,num_buckets", [(False, 2), (True, 2), (False, 3), (True, 3)])
def test_two_rank_step_equals_single_process_step(tmp_path, use_graph, num_buckets):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    single = _run(script, 1, use_graph, tmp_path / "single.npz", num_buckets)
    dual = _run(script, 2, use_graph, tmp_path / "dual.npz", num_buckets)
    assert str(dual["grad_sync"]) == ("segmented" if use_graph else "eager") and str(single["grad_sync"]) == "none"
    # same loss trajectory and the same weights after 3 Adam steps, to fp32 summation-order tolerance
    assert np.allclose(single["losses"], dual["losses"], rtol=1e-5, atol=1e-6), (single["losses"], dual["losses"])
    # Adam normalises each update to ~lr, so a weight whose gradient is ~0 can legitimately flip sign on a
    # summation-order difference; everything else must agree far below lr = 1e-3
    diff = np.abs(single["flat"] - dual["flat"])
    assert np.median(diff) <= 1e-6, np.median(diff)
    assert (diff > 2e-4).mean() <= 1e-3, (diff > 2e-4).mean()
    assert diff.max() <= 3 * 2e-3 + 1e-6                       # never more than 3 steps of +-lr apart
    # metric counters: the two shards' counts, all-reduced, cover the global batch exactly (tp + fn = utterances per
    # class at every threshold); the weights differ by ~1e-6, so individual threshold decisions may flip, C_avg barely
    assert np.array_equal(single["tp"] + single["fn"], dual["tp"] + dual["fn"])
    assert (single["tp"] + single["fn"]).sum(axis=0).max() == 12
    assert abs(float(single["cavg"]) - float(dual["cavg"])) <= 0.05
    assert np.abs(single["per_th"] - dual["per_th"]).max() <= 0.1


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_step_with_bf16_gradient_wire_tracks_the_fp32_exchange(tmp_path, use_graph):
    """Trainer(grad_wire_dtype="bfloat16"): the buckets are rounded once (lidbox_f32_to_bf16), summed as bf16 by the collective and
    widened back (lidbox_bf16_to_f32) -- half the bytes on the links.  Against the single-process fp32 step: the loss
    trajectory stays within bf16's relative step, no weight moves by more than the steps' worth of +-lr."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    single = _run(script, 1, use_graph, tmp_path / "single.npz", 3)
    dual = _run(script, 2, use_graph, tmp_path / "dual16.npz", 3, wire="bfloat16")
    assert str(dual["grad_sync"]) == ("segmented" if use_graph else "eager")
    assert np.allclose(single["losses"], dual["losses"], rtol=2e-2, atol=1e-4), (single["losses"], dual["losses"])
    assert not np.array_equal(single["flat"], dual["flat"])              # the wire format did round something
    diff = np.abs(single["flat"] - dual["flat"])
    assert np.median(diff) <= 2e-4 and diff.max() <= 3 * 2e-3 + 1e-6, (np.median(diff), diff.max())


def test_rccl_backend_between_graph_segments_world1():
    """The real RCCL backend (nccl) on the one GPU there is: world_size 1 with the collective path forced on,
    so init_process_group('nccl'), the side-stream all_reduce launches and the three-segment hipGraph replay are
    all exercised exactly as in the N-GPU bench (the reduction itself is the identity at world 1)."""
    import json
    env = dict(os.environ, LIDBOX_FORCE_GRAD_SYNC="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LIDBOX_REQUIRE_INGRAPH_SYNC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--batch", "32", "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line (RCCL's banner belongs on stderr): %r" % lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["value"] > 0
    assert np.isfinite(r["config"]["final_loss"])
    # the capture of the collectives must have SUCCEEDED (LIDBOX_REQUIRE_INGRAPH_SYNC turns the segmented fallback into an
    # error) and the line says which form the timed steps used
    assert r["config"]["grad_sync"] == "in_graph", r["config"]
    assert r["config"]["grad_buckets"] == 3 and r["config"]["allreduce_bytes_per_step"] == 4 * 4510176
    assert r["config"]["allreduce_wire_dtype"] == "float32" and sum(r["config"]["allreduce_bucket_bytes"]) == 4 * 4510176
    assert r["grad_sync_exposed_wait_us"]["steps"] == 8 and r["grad_sync_exposed_wait_us"]["median"] >= 0.0
    assert r["sustained"]["steps"] >= 20 and r["sustained"]["value"] > 0
    # the same with the bf16 wire format: half the bytes, still captured inside the step graph
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    p = subprocess.run(cmd + ["--grad-wire-dtype", "bfloat16", "--sustain-seconds", "0"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert r["config"]["grad_sync"] == "in_graph" and r["config"]["allreduce_bytes_per_step"] == 2 * 4510176
    assert r["config"]["allreduce_wire_dtype"] == "bfloat16" and np.isfinite(r["config"]["final_loss"]) and "sustained" not in r


def test_segmented_sync_is_refused_when_in_graph_is_required():
    """LIDBOX_REQUIRE_INGRAPH_SYNC=1 + a step that cannot capture its collectives (forced segmented form): an error, not a
    silently slower run"""
    env = dict(os.environ, LIDBOX_FORCE_GRAD_SYNC="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LIDBOX_REQUIRE_INGRAPH_SYNC="1",
               LIDBOX_SEGMENTED_SYNC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode != 0 and "LIDBOX_REQUIRE_INGRAPH_SYNC" in (p.stdout + p.stderr)


def test_bench_line_contract_single_process():
    """`python bench.py` (no launcher): one JSON line with the contract's keys, the roofline of the dominant GEMM
    instantiation from live HIP events and the feature kernel's HBM roofline"""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--sustain-seconds", "1.0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_feature", "kernels", "secondary"):
        assert key in r, key
    assert r["config"]["grad_sync"] == "none" and r["config"]["allreduce_bytes_per_step"] == 0
    # the other two single-GPU configurations ride in the same line: configs[3] fp32 and one GPU's shard of configs[4] in bf16
    sec = r["secondary"]
    assert [x["config"]["baseline_config"] for x in sec] == [3, 4] and [x["dtype"] for x in sec] == ["f32", "bf16"]
    for x in sec:
        assert x["value"] > 0 and x["steps"] == 4 and np.isfinite(x["config"]["final_loss"])
        assert abs(x["value"] - x["config"]["per_gpu_batch"] * 1e3 / x["ms_per_step"]) <= 1e-3 * x["value"]
        assert 0 < x["roofline"]["frac"] < 1 and x["roofline"]["bound"] == "mfma"
    assert r["unit"] == "utterances/s" and r["dtype"] == "f32" and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["config"]["global_batch"] == 256 and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and rf["kernel"] in r["kernels"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches_per_step"] >= 1
    ff = r["roofline_feature"]
    assert ff["bound"] == "hbm" and ff["bytes_per_launch"] == 256 * 159680 and 0 < ff["frac"] < 1
    assert abs(r["value"] - 256 * 1e3 / r["ms_per_step"]) <= 1e-3 * r["value"]
    # the sustained block (same captured steps, >= --sustain-seconds) and the label noise that keeps the loss off zero
    su = r["sustained"]
    assert su["seconds"] >= 1.0 and su["steps"] >= 20 and abs(su["value"] - 256 * 1e3 / su["ms_per_step"]) <= 1e-3 * su["value"]
    assert 0.5 * r["value"] <= su["value"] <= 2.0 * r["value"]
    assert r["config"]["label_noise"] == 0.35 and r["config"]["final_loss"] > 0.3 and r["config"]["first_loss"] > 0.3
    assert all("sustained" in x and x["sustained"]["value"] > 0 for x in sec) and sec[0]["config"]["label_noise"] == 0.35


def test_uneven_shards_equal_the_global_batch_step(tmp_path):
    """13 utterances over two ranks (7 + 6): every rank scales its loss gradient by 1 / GLOBAL batch (Trainer._loss_scale),
    so the all-reduced sum is the single-process gradient (mean of per-rank means over-weights the small shard)"""
    script = tmp_path / "worker13.py"
    script.write_text((_WORKER % {"root": ROOT}).replace("B = 12", "B = 13"))
    single = _run(script, 1, True, tmp_path / "single.npz", 2)
    dual = _run(script, 2, True, tmp_path / "dual.npz", 2)
    diff = np.abs(single["flat"] - dual["flat"])
    assert np.median(diff) <= 1e-6 and (diff > 2e-4).mean() <= 1e-3 and diff.max() <= 3 * 2e-3 + 1e-6
    assert np.array_equal(single["tp"] + single["fn"], dual["tp"] + dual["fn"])


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the box (RCCL refuses two ranks on one device)")


@needs_two_gpus
def test_two_rccl_ranks_on_two_gpus_bench_line():
    """lights up on a multi-GPU box: `python bench.py --gpus 2` with NO launcher re-executes itself under
    torch.distributed.run, two ranks all-reduce over RCCL / xGMI between the hipGraph segments, one JSON line comes back"""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LIDBOX_REQUIRE_INGRAPH_SYNC="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "64",
           "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 128 and r["config"]["parallelism"] == "dp2" and r["value"] > 0
    assert r["config"]["grad_sync"] == "in_graph" and r["rank_ms_per_step"]["min"] <= r["rank_ms_per_step"]["max"]


@needs_two_gpus
def test_two_rccl_ranks_step_equals_single_process_step(tmp_path):
    """the equivalence test above with the real backend: one rank per GPU, RCCL all-reduce"""
    script = tmp_path / "worker_rccl.py"
    script.write_text((_WORKER % {"root": ROOT})
                      .replace('os.environ["LOCAL_RANK"] = "0"', 'os.environ["LOCAL_RANK"] = os.environ["RANK"]')
                      .replace('init_distributed(backend="gloo")', 'init_distributed(backend="nccl")')
                      .replace("torch.cuda.set_device(0)", "torch.cuda.set_device(rank)"))
    single = _run(script, 1, True, tmp_path / "single.npz", 3)
    dual = _run(script, 2, True, tmp_path / "dual.npz", 3)
    assert np.allclose(single["losses"], dual["losses"], rtol=1e-5, atol=1e-6)
    diff = np.abs(single["flat"] - dual["flat"])
    assert np.median(diff) <= 1e-6 and diff.max() <= 3 * 2e-3 + 1e-6
    assert str(dual["grad_sync"]) == "in_graph" and str(single["grad_sync"]) == "none"


_WORKER_TWO_SHAPES = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer, init_distributed

    use_graph = bool(int(sys.argv[1]))
    out_path = sys.argv[2]
    os.environ["LOCAL_RANK"] = "0"
    rank, world, _ = init_distributed(backend="gloo")
    torch.cuda.set_device(0)
    sig, y = synthetic_batch(13, num_labels=4, duration_s=0.5)
    # step A: 13 utterances as 7 + 6; step B: the first 11 as 7 + 4 -- rank 0's shard size does NOT change, the global batch does
    cuts = {1: [(0, 13), (0, 11)], 2: [((0, 7), (7, 13)), ((0, 7), (7, 11))]}[world]
    model = xvector.create((48, 40), 4, seed=0)
    plan = audio.get_plan(16000, 400, 160)
    tr = Trainer(model, feature=dict(plan=plan, kind=nv.FEAT_LOGMEL), use_graph=use_graph, num_buckets=2)
    losses = []
    for step, total in ((0, 13), (1, 11), (0, 13), (1, 11)):
        lo, hi = cuts[step] if world == 1 else cuts[step][rank]
        sd = torch.from_numpy(sig[lo:hi]).cuda()
        yd = torch.from_numpy(y[lo:hi].astype(np.int32)).cuda()
        losses.append(float(tr.train_step(sd, yd, global_batch=total)) * (hi - lo) / total)     # this rank's share of the global mean
    torch.cuda.synchronize()
    l = torch.tensor(losses, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(l)
    if rank == 0:
        np.savez(out_path, flat=model.flat.cpu().numpy(), losses=l.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
""")


@pytest.mark.parametrize("use_graph", [False, True])
def test_global_batch_changes_while_one_ranks_shard_does_not(tmp_path, use_graph):
    """the loss scale must follow the GLOBAL batch of every step, also when this rank's own shard size stays the
    same (a short last batch that only shortens the last rank's shard).  Two ranks run 7 + 6, then 7 + 4 utterances, twice,
    passing `global_batch=`; a single process runs 13, then 11: same loss trajectory and weights (the scale is part of the
    captured step's key, so the graph path keeps one graph per global batch)."""
    script = tmp_path / "worker_two_shapes.py"
    script.write_text(_WORKER_TWO_SHAPES % {"root": ROOT})
    single = _run(script, 1, use_graph, tmp_path / "single.npz")
    dual = _run(script, 2, use_graph, tmp_path / "dual.npz")
    assert np.allclose(single["losses"], dual["losses"], rtol=2e-5, atol=1e-6), (single["losses"], dual["losses"])
    diff = np.abs(single["flat"] - dual["flat"])
    assert np.median(diff) <= 1e-6 and (diff > 2e-4).mean() <= 2e-3 and diff.max() <= 4 * 2e-3 + 1e-6


_WORKER_SHARD_CHANGE = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, %(root)r)
    from lidbox_amd import _native as nv
    from lidbox_amd.features import audio
    from lidbox_amd.models import xvector
    from lidbox_amd.testutil import synthetic_batch
    from lidbox_amd.train import Trainer, init_distributed
    os.environ["LOCAL_RANK"] = "0"
    rank, world, _ = init_distributed(backend="gloo")
    torch.cuda.set_device(0)
    sig, y = synthetic_batch(6, num_labels=4, duration_s=0.5)
    sd, yd = torch.from_numpy(sig).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()
    model = xvector.create((48, 40), 4, seed=0)
    tr = Trainer(model, feature=dict(plan=audio.get_plan(16000, 400, 160), kind=nv.FEAT_LOGMEL), use_graph=False)
    assert tr.sync.active and tr.global_batch_of(5) == 5
    tr.train_step(sd, yd)                                  # first step: shard sizes summed once, on every rank
    tr.train_step(sd[:4].contiguous(), yd[:4].contiguous(), global_batch=4)      # explicit: fine
    try:
        tr.train_step(sd[:4].contiguous(), yd[:4].contiguous())                  # a changed shard size without it: refused
    except ValueError as e:
        assert "global_batch" in str(e)
        print("REFUSED")
    tr.train_step(sd, yd)                                  # the first size still works
    torch.cuda.synchronize()
""")


def test_changed_shard_size_without_global_batch_is_refused(tmp_path):
    """under data parallelism a collective that only SOME ranks enter (the old per-new-local-size exchange) would
    pair with the other ranks' gradient all-reduce; the shard sizes are summed once at the first step, and a later change
    of this rank's size without `global_batch=` raises instead"""
    script = tmp_path / "worker_shard_change.py"
    script.write_text(_WORKER_SHARD_CHANGE % {"root": ROOT})
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), LIDBOX_FORCE_GRAD_SYNC="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "REFUSED" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
