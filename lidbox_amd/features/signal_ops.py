"""
Ragged-batch driver for the signal kernels of csrc/signal.hip: energy VAD, VAD application,
fixed-length chunking, peak normalisation, RMS and the SNR mixer for MANY variable-length signals per launch.
`lidbox_amd.features.audio` (single-signal functions with the reference's names) and
`lidbox_amd.data.steps` (dataset steps) are thin layers over this module.

A RaggedSignals holds the samples of B utterances in one device buffer plus `starts` / `lengths` (int64, host
and device copies); every utterance starts on a 16-byte boundary so the kernels' vector paths apply.  Sizes
that depend on the data (speech frames per utterance) are read back once per step -- the same synchronisation
point the reference's eager tensors have.
"""
import numpy as np
import torch

from .. import _native as nv

ALIGN = 4            # samples (16 bytes)


def _device(device=None):
    if not torch.cuda.is_available():
        raise nv.LidboxHipError("lidbox_amd signal ops need a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _aligned_starts(lengths):
    padded = (np.asarray(lengths, np.int64) + ALIGN - 1) // ALIGN * ALIGN
    starts = np.concatenate(([0], np.cumsum(padded)))
    return starts[:-1].astype(np.int64), int(starts[-1])


def _csr(counts, device):
    off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    return off, torch.from_numpy(off).to(device)


class RaggedSignals:
    """utterance b = flat[starts[b] : starts[b] + lengths[b]]"""

    def __init__(self, flat, starts, lengths):
        self.flat = nv.require_gpu_tensor(flat, "flat", torch.float32)
        self.starts_host = np.asarray(starts, np.int64).reshape(-1)
        self.lengths_host = np.asarray(lengths, np.int64).reshape(-1)
        if len(self.starts_host) != len(self.lengths_host) or (self.lengths_host < 0).any():
            raise ValueError("starts and lengths must have one non-negative entry per utterance")
        if len(self.starts_host) and int((self.starts_host + self.lengths_host).max()) > self.flat.numel():
            raise ValueError("an utterance runs past the end of the sample buffer")
        dev = self.flat.device
        self.starts = torch.from_numpy(self.starts_host).to(dev)
        self.lengths = torch.from_numpy(self.lengths_host).to(dev)

    @classmethod
    def from_list(cls, signals, device=None):
        """1-D signals (tensors or arrays, any device) -> one device buffer, 16-byte aligned starts"""
        dev = _device(device)
        sigs = [torch.as_tensor(s, dtype=torch.float32).reshape(-1) for s in signals]
        lengths = np.array([int(s.numel()) for s in sigs], np.int64)
        starts, total = _aligned_starts(lengths)
        flat = torch.zeros(max(total, ALIGN), dtype=torch.float32, device=dev)
        for s, st in zip(sigs, starts):
            if s.numel():
                flat[int(st):int(st) + s.numel()].copy_(s)
        return cls(flat, starts, lengths)

    @classmethod
    def from_dense(cls, signals):
        """[B, N] device tensor (rows become utterances; no copy when N is a multiple of 4)"""
        x = nv.require_gpu_tensor(signals, "signals", torch.float32)
        if x.dim() != 2:
            raise ValueError("signals must be [B, N]")
        B, N = x.shape
        if N % ALIGN == 0 and x.is_contiguous():
            return cls(x.reshape(-1) if x.numel() else torch.zeros(ALIGN, device=x.device), np.arange(B) * N, np.full(B, N))
        return cls.from_list(list(x), device=x.device)

    @property
    def B(self):
        return len(self.lengths_host)

    def split(self, flat=None):
        flat = self.flat if flat is None else flat
        return [flat[int(s):int(s) + int(n)] for s, n in zip(self.starts_host, self.lengths_host)]


def frame_rms(r, frame_len):
    """RMS of every non-overlapping frame (audio.py:314-317) -> (rms [total_frames], frame CSR host / device)"""
    frame_len = int(frame_len)
    if frame_len < 1:
        raise ValueError("frame length must be at least one sample")
    fo_h, fo_d = _csr(r.lengths_host // frame_len, r.flat.device)
    total = int(fo_h[-1])
    out = torch.empty(total, dtype=torch.float32, device=r.flat.device)
    with torch.cuda.device(r.flat.device):
        nv.check(nv.lib.lidbox_frame_rms(nv.ptr(r.flat), nv.ptr(r.starts), nv.ptr(fo_d), r.B, total, frame_len,
                                         nv.ptr(out), nv.current_stream()))
    return out, fo_h, fo_d


def vad_decisions(r, frame_len, min_non_speech_frames=0, strength=0.05, min_rms_threshold=1e-3):
    """audio.py:308-329 for every utterance of the batch -> dict(decisions uint8 [total_frames], slots, counts,
    frame_offsets (CSR), thresholds, rms)"""
    rms, fo_h, fo_d = frame_rms(r, frame_len)
    dev = r.flat.device
    total = int(fo_h[-1])
    dec = torch.empty(total, dtype=torch.uint8, device=dev)
    slots = torch.empty(total, dtype=torch.int32, device=dev)
    counts = torch.zeros(r.B, dtype=torch.int32, device=dev)
    thr = torch.empty(r.B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nv.check(nv.lib.lidbox_vad_decisions(nv.ptr(rms), nv.ptr(fo_d), r.B, total, float(strength),
                                             float(min_rms_threshold), int(min_non_speech_frames), nv.ptr(dec),
                                             nv.ptr(slots), nv.ptr(counts), nv.ptr(thr), nv.current_stream()))
    return dict(decisions=dec, slots=slots, counts=counts, frame_offsets_host=fo_h, frame_offsets=fo_d,
                frame_len=int(frame_len), rms=rms, thresholds=thr)


def split_frames(vad, x):
    """per-utterance views of a per-frame tensor (decisions, rms, ...)"""
    fo = vad["frame_offsets_host"]
    return [x[int(fo[b]):int(fo[b + 1])] for b in range(len(fo) - 1)]


def apply_vad(r, vad):
    """steps.py:191-198: keep the speech frames of every utterance -> new RaggedSignals"""
    dev = r.flat.device
    L = vad["frame_len"]
    counts = vad["counts"].cpu().numpy().astype(np.int64)          # the one host read-back of this step
    lengths = counts * L
    starts, total = _aligned_starts(lengths)
    out = torch.zeros(max(total, ALIGN), dtype=torch.float32, device=dev)
    starts_d = torch.from_numpy(starts).to(dev)
    nframes = int(vad["frame_offsets_host"][-1])
    with torch.cuda.device(dev):
        nv.check(nv.lib.lidbox_apply_vad(nv.ptr(r.flat), nv.ptr(r.starts), nv.ptr(vad["frame_offsets"]),
                                         nv.ptr(vad["decisions"]), nv.ptr(vad["slots"]), nv.ptr(starts_d), r.B, nframes,
                                         L, nv.ptr(out), nv.current_stream()))
    return RaggedSignals(out, starts, lengths)


def chunk_plan(num_samples, sample_rate, length_ms, step_ms, max_pad_ms=0):
    """steps.py:586-588,604-614 in the reference's float32: (chunk_length, chunk_step, padded length, chunks)"""
    out = (nv.C.c_long * 4)()
    nv.check(nv.lib.lidbox_signal_chunk_plan(int(num_samples), int(sample_rate), int(length_ms), int(step_ms),
                                             int(max_pad_ms), out))
    return tuple(int(v) for v in out)


def chunk_counts(lengths, L, S, max_pad):
    """steps.py:607-614 for an array of signal lengths (pure integer arithmetic; L, S, max_pad in samples)"""
    n = np.asarray(lengths, np.int64)
    full = np.maximum(0, 1 + (n - L) // S)                       # numpy // floors like tf's
    last = n - full * S
    n = np.where((last < L) & (L <= last + max_pad), n + L - last, n)
    return np.where(n >= L, 1 + (n - L) // S, 0).astype(np.int64)


def signal_chunks(r, sample_rate, length_ms, step_ms, max_pad_ms=0):
    """steps.py:600-614 for the whole batch -> (chunks [total_chunks, L] dense, chunks per utterance)"""
    dev = r.flat.device
    L, S = chunk_plan(0, sample_rate, length_ms, step_ms, max_pad_ms)[:2]
    P = int(np.float32(sample_rate) * np.float32(1e-3 * max_pad_ms))                        # float32 cast chain of :588,606
    nch = chunk_counts(r.lengths_host, L, S, P)
    co_h, co_d = _csr(nch, dev)
    total = int(co_h[-1])
    out = torch.empty((total, L), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nv.check(nv.lib.lidbox_signal_chunks(nv.ptr(r.flat), nv.ptr(r.starts), nv.ptr(r.lengths), nv.ptr(co_d), r.B,
                                             total, L, S, nv.ptr(out), nv.current_stream()))
    return out, nch


def peak_normalize(r, dBFS=0.0):
    """audio.py:57-59 per utterance -> RaggedSignals with the same layout (alignment gaps are left unspecified)"""
    out = torch.empty_like(r.flat)
    with torch.cuda.device(r.flat.device):
        max_len = int(r.lengths_host.max()) if r.B else 0
        aligned = int(r.B == 0 or bool((r.starts_host % ALIGN == 0).all()))
        nv.check(nv.lib.lidbox_peak_normalize_max(nv.ptr(r.flat), nv.ptr(r.starts), nv.ptr(r.lengths), r.B, float(dBFS),
                                                  max_len, aligned, nv.ptr(out), nv.current_stream()))
    return RaggedSignals(out, r.starts_host, r.lengths_host)


def signal_rms(r):
    """audio.py:266-270 per utterance -> [B]"""
    out = torch.empty(r.B, dtype=torch.float32, device=r.flat.device)
    with torch.cuda.device(r.flat.device):
        nv.check(nv.lib.lidbox_signal_rms(nv.ptr(r.flat), nv.ptr(r.starts), nv.ptr(r.lengths), r.B, nv.ptr(out),
                                          nv.current_stream()))
    return out


def snr_mixer(clean, noise, snr):
    """audio.py:128-148 on dense batches: clean, noise [B, N] (or [N]), snr [B] (or scalar) ->
    (clean_norm, noisenewlevel, noisyspeech)"""
    c = nv.require_gpu_tensor(clean, "clean", torch.float32)
    z = nv.require_gpu_tensor(noise, "noise", torch.float32)
    single = c.dim() == 1
    c2 = c.reshape(1, -1) if single else c
    z2 = z.reshape(1, -1) if z.dim() == 1 else z
    if c2.shape != z2.shape:
        raise ValueError("mismatching length for signals clean and noise given to snr mixer")        # audio.py:132
    c2, z2 = c2.contiguous(), z2.contiguous()
    B, N = c2.shape
    s = torch.as_tensor(snr, dtype=torch.float32).reshape(-1).to(c2.device)
    if s.numel() == 1 and B > 1:
        s = s.expand(B)
    s = s.contiguous()
    if s.numel() != B:
        raise ValueError("snr must be a scalar or one value per signal")
    outs = [torch.empty_like(c2) for _ in range(3)]
    if B and N:
        with torch.cuda.device(c2.device):
            nv.check(nv.lib.lidbox_snr_mixer(nv.ptr(c2), nv.ptr(z2), nv.ptr(s), B, N, nv.ptr(outs[0]), nv.ptr(outs[1]),
                                             nv.ptr(outs[2]), nv.current_stream()))
    return tuple(o.reshape(-1) if single else o for o in outs)
