"""
Counterpart of lidbox/features/audio.py for the hot path: the same function names, argument
names and defaults as the reference (cited per function, file:line relative to the lidbox
checkout), over torch tensors on the HIP device.  All arithmetic runs in liblidbox_hip.so.
The energy VAD, silence removal, peak normalisation, RMS and SNR-mixer helpers run on the
ragged-batch kernels of csrc/signal.hip through `signal_ops`; 16-bit PCM is scaled and channel-averaged on the device
(`pcm16_to_float`, `read_wav`: the RIFF header is parsed on the host, the samples cross PCIe as int16); MP3 decoding,
resampling, WebRTC VAD and the random FIR augmentation stay out of scope (host libraries).
"""
import math
import threading

import torch

from .. import _native as nv
from . import mel_ops, signal_ops

_plans = {}
_plan_lock = threading.Lock()


class FeaturePlan:
    """Immutable device tables for one (sample_rate, framing, fft, mel, mfcc) configuration."""

    def __init__(self, sample_rate, frame_length, frame_step, fft_length, power, num_mel_bins, fmin, fmax,
                 coef_begin, coef_end):
        import ctypes
        h = ctypes.c_void_p()
        nv.check(nv.lib.lidbox_feat_plan_create(int(sample_rate), int(frame_length), int(frame_step),
                                                int(fft_length), float(power), int(num_mel_bins),
                                                float(fmin), float(fmax), int(coef_begin), int(coef_end),
                                                ctypes.byref(h)))
        self.handle = h
        self.frame_length, self.frame_step, self.fft_length = int(frame_length), int(frame_step), int(fft_length)
        self.power = float(power)
        self.fused = bool(nv.lib.lidbox_feat_plan_is_fused(h, 2, None, 0))      # the fused kernels need no workspace

    def channels(self, kind):
        return nv.lib.lidbox_feat_plan_channels(self.handle, kind)

    def num_frames(self, num_samples):
        return nv.lib.lidbox_num_frames(int(num_samples), self.frame_length, self.frame_step)

    def run(self, kind, signals, out=None, out_batch_stride=0, out16=None, nonfinite=None):
        """signals [B,N] on the HIP device -> [B,T,C].  float32, or int16: 16-bit mono PCM read in place (value / 32768 as
        tf.audio.decode_wav scales it, reference audio.py:17-23; bit-identical to pcm16_to_float followed by the float call, half
        the bytes read).  out16: optional bfloat16 tensor laid out like `out`, receives bf16(out) from the same call (the shadow the
        bf16-storage Conv1D path reads).  nonfinite: optional int32 device tensor of one element the kernel ORs 1 into when a value
        it wrote is NaN / Inf (the caller zeroes it and reads it when it wants the answer: `all_finite`)."""
        if not isinstance(signals, torch.Tensor) or signals.dtype not in (torch.float32, torch.int16):
            nv.require_gpu_tensor(signals, "signals", torch.float32)      # raises the usual message
        nv.require_gpu_tensor(signals, "signals", signals.dtype)
        if signals.dim() != 2:
            raise ValueError("Input signals for feature extraction must be batches of mono signals "
                             "without channels, i.e. of shape [B, N]")
        if signals.stride(1) != 1:
            signals = signals.contiguous()
        B, N = signals.shape
        T, C = self.num_frames(N), self.channels(kind)
        if out is None:
            out = torch.empty((B, T, C), dtype=torch.float32, device=signals.device)
        if B == 0 or T == 0:
            return out
        stride = signals.stride(0) if B > 1 else N      # a size-1 dim may carry any stride
        pcm = signals.dtype == torch.int16
        if pcm and not self._pcm_in_place(signals, stride):
            signals = pcm16_to_float(signals.contiguous().reshape(-1), 1, device=signals.device).reshape(B, N)      # one conversion pass, then the float path
            stride, pcm = N, False
        wbytes = 0 if (pcm or self.fused) else nv.lib.lidbox_extract_features_workspace(self.handle, kind, B, N, nv.ptr(signals), stride)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=signals.device) if wbytes else None
        if out16 is not None:
            nv.require_gpu_tensor(out16, "out16", torch.bfloat16)
        flag_ptr = None
        if nonfinite is not None:
            flag_ptr = nonfinite.ptr if isinstance(nonfinite, FiniteFlag) else nv.ptr(nv.require_gpu_tensor(nonfinite, "nonfinite", torch.int32))

        def call():
            nv.check(nv.lib.lidbox_extract_features_fwd_ex(self.handle, kind, nv.ptr(signals), nv.SRC_PCM16 if pcm else nv.SRC_F32, B, N,
                                                           stride, nv.ptr(out), int(out_batch_stride), nv.ptr(out16), flag_ptr,
                                                           nv.ptr(ws), wbytes, nv.current_stream()))
        if signals.device.index == torch.cuda.current_device():
            call()
        else:
            with torch.cuda.device(signals.device):
                call()
        return out

    def _pcm_in_place(self, signals, stride):
        """what lidbox_extract_features_fwd_ex asks of a 16-bit source (include/lidbox_hip.h): the fused fft_length-512 kernel with
        power 2, 8-byte aligned rows, row stride / frame_length / frame_step multiples of 4"""
        return (self.fft_length == 512 and self.power == 2.0 and self.frame_length <= 512 and signals.data_ptr() % 8 == 0
                and stride % 4 == 0 and self.frame_length % 4 == 0 and self.frame_step % 4 == 0
                and bool(nv.lib.lidbox_feat_plan_is_fused(self.handle, 2, nv.ptr(signals), stride)))


def all_finite(flag):
    """True when no call since the flag was zeroed wrote a NaN / Inf (one 4-byte read; synchronises the stream)."""
    if isinstance(flag, FiniteFlag):
        return flag.ok()
    return int(flag.item()) == 0


class FiniteFlag:
    """The word the feature kernels OR "a stored value was not finite" into, in PINNED HOST memory the device writes directly (a
    kernel touches it only when it has found such a value): asking for the answer is a stream synchronisation and a host load, no
    device-to-host copy and no allocation per call.  One per (thread, device), re-armed after it has fired."""

    _local = threading.local()

    def __init__(self):
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        import ctypes
        self.ptr = ctypes.c_void_p(self.host.data_ptr())      # hipHostMalloc'd: the same address is valid on the device

    @classmethod
    def get(cls, device):
        flags = getattr(cls._local, "flags", None)
        if flags is None:
            flags = cls._local.flags = {}
        key = torch.device(device).index
        f = flags.get(key)
        if f is None:
            f = flags[key] = cls()
        return f

    def ok(self):
        torch.cuda.current_stream().synchronize()      # (polling an event instead was measured: no faster)
        if int(self.host[0]) == 0:
            return True
        self.host[0] = 0
        return False


def get_plan(sample_rate, frame_length, frame_step, fft_length=512, power=2.0, num_mel_bins=40, fmin=0.0,
             fmax=8000.0, coef_begin=1, coef_end=13, device=None):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (dev, int(sample_rate), int(frame_length), int(frame_step), int(fft_length), float(power),
           int(num_mel_bins), float(fmin), float(fmax), int(coef_begin), int(coef_end))
    with _plan_lock:
        plan = _plans.get(key)
        if plan is None:
            with torch.cuda.device(dev):
                plan = FeaturePlan(*key[1:])
            _plans[key] = plan
    return plan


def _scalar(x):
    return x.item() if isinstance(x, torch.Tensor) else x


def ms_to_frames(sample_rate, ms):
    """reference lidbox/features/audio.py:185-189 (float32 arithmetic, truncation)."""
    return nv.lib.lidbox_ms_to_frames(int(_scalar(sample_rate)), int(_scalar(ms)))


def spectrograms(signals, sample_rate, frame_length_ms=25, frame_step_ms=10, power=2.0, fft_length=512):
    """reference lidbox/features/audio.py:219-230: |tf.signal.stft|^power, [B,N] -> [B,T,fft_length//2+1]."""
    sample_rate = int(_scalar(sample_rate))
    frame_length = ms_to_frames(sample_rate, frame_length_ms)
    frame_step = ms_to_frames(sample_rate, frame_step_ms)
    plan = get_plan(sample_rate, frame_length, frame_step, int(fft_length), float(power),
                    device=signals.device if isinstance(signals, torch.Tensor) else None)
    return plan.run(nv.FEAT_SPECTROGRAM, signals)


def linear_to_mel(spectrograms, sample_rate, num_mel_bins=40, fmin=0.0, fmax=8000.0):
    """reference lidbox/features/audio.py:247-261: tensordot(S [B,T,F], W [F,M], 1) on the MFMA GEMM."""
    S = nv.require_gpu_tensor(spectrograms, "spectrograms", torch.float32)
    if S.dim() != 3:
        raise ValueError("spectrograms must be [B, T, F]")
    S = S.contiguous()
    B, T, F = S.shape
    W = mel_ops.linear_to_mel_weight_matrix(num_mel_bins=num_mel_bins, num_spectrogram_bins=F,
                                            sample_rate=int(_scalar(sample_rate)), lower_edge_hertz=fmin,
                                            upper_edge_hertz=fmax, device=S.device)
    out = torch.empty((B, T, int(num_mel_bins)), dtype=torch.float32, device=S.device)
    if B * T == 0:
        return out
    A = nv.Rows(S.data_ptr(), 0, F, 1, B * T)
    Cd = nv.Rows(out.data_ptr(), 0, int(num_mel_bins), 1, B * T)
    with torch.cuda.device(S.device):
        nv.check(nv.lib.lidbox_gemm_nn(A, nv.ptr(W), int(num_mel_bins), Cd, F, int(num_mel_bins), nv.EPI_NONE,
                                       None, None, 0, nv.current_stream()))
    return out


def _minmax(x):
    scratch = torch.empty(2048 + 2, dtype=torch.float32, device=x.device)
    mm = scratch[2048:]
    with torch.cuda.device(x.device):
        nv.check(nv.lib.lidbox_minmax(nv.ptr(x), x.numel(), nv.ptr(mm), nv.ptr(scratch), nv.current_stream()))
    return mm


def log10(x):
    """reference lidbox/features/audio.py:162-164: ln(x) / ln(10), float32 (lidbox_log10_fwd)."""
    x = nv.require_gpu_tensor(x, "x", torch.float32).contiguous()
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            nv.check(nv.lib.lidbox_log10_fwd(nv.ptr(x), x.numel(), nv.ptr(out), nv.current_stream()))
    return out


def power_to_db(S, amin=1e-10, top_db=80.0):
    """reference lidbox/features/audio.py:167-174 (factor 20, batch-global max)."""
    S = nv.require_gpu_tensor(S, "S", torch.float32).contiguous()
    out = torch.empty_like(S)
    if S.numel() == 0:
        return out
    mm = _minmax(S)
    with torch.cuda.device(S.device):
        nv.check(nv.lib.lidbox_power_to_db_fwd(nv.ptr(S), S.numel(), nv.ptr(mm), float(amin), float(top_db),
                                               nv.ptr(out), nv.current_stream()))
    return out


def db_to_power(S):
    """reference lidbox/features/audio.py:177-181."""
    return torch.pow(10.0, S / 20.0)


def fft_frequencies(sample_rate, n_fft):
    """reference lidbox/features/audio.py:151-159 (host constant)."""
    return torch.linspace(0.0, float(int(sample_rate) // 2), 1 + int(n_fft) // 2)


# ------------------------------------------------------------------ 16-bit PCM ingest (reference audio.py:17-23)
def pcm16_to_float(pcm, channels=1, device=None):
    """tf.audio.decode_wav's sample arithmetic + the channel average of reference lidbox/features/audio.py:20-22 on the device:
    `pcm` = int16 samples, frames x channels interleaved (torch tensor on any device, or anything numpy can view as int16);
    returns the float32 signal [frames] on the HIP device, bit-identical to (pcm / 32768).mean(axis=1) in float32."""
    import numpy as np
    channels = int(channels)
    if not isinstance(pcm, torch.Tensor):
        pcm = torch.from_numpy(np.ascontiguousarray(np.asarray(pcm, dtype=np.int16)))
    if pcm.dtype != torch.int16:
        raise TypeError("pcm must be int16, got %s" % pcm.dtype)
    dev = torch.device(device) if device is not None else (pcm.device if pcm.is_cuda else torch.device("cuda"))
    pcm = pcm.reshape(-1).to(dev, non_blocking=True).contiguous()
    if channels < 1 or pcm.numel() % channels:
        raise ValueError("%d samples are not whole frames of %d channels" % (pcm.numel(), channels))
    frames = pcm.numel() // channels
    out = torch.empty(frames, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nv.check(nv.lib.lidbox_pcm16_to_f32(nv.ptr(pcm), frames, channels, nv.ptr(out), nv.current_stream()))
    return out


def parse_wav_pcm16(data):
    """RIFF/WAVE container -> (int16 samples as a numpy view of `data`, channels, sample_rate).  16-bit PCM only, like
    tf.audio.decode_wav (format tag 1, or WAVE_FORMAT_EXTENSIBLE with the PCM subformat); chunks other than `fmt ` / `data` are
    skipped; a `data` size that runs past the end of the file is cut to whole frames."""
    import struct
    import numpy as np
    mv = memoryview(data)
    if len(mv) < 12 or bytes(mv[0:4]) != b"RIFF" or bytes(mv[8:12]) != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt = 12, None
    while pos + 8 <= len(mv):
        cid, size = bytes(mv[pos:pos + 4]), struct.unpack_from("<I", mv, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            if size < 16:
                raise ValueError("short fmt chunk")
            tag, nch, rate, _, _, bits = struct.unpack_from("<HHIIHH", mv, body)
            if tag == 0xFFFE and size >= 26:
                tag = struct.unpack_from("<H", mv, body + 24)[0]
            fmt = (tag, nch, rate, bits)
        elif cid == b"data":
            if fmt is None:
                raise ValueError("data chunk before fmt chunk")
            tag, nch, rate, bits = fmt
            if tag != 1 or bits != 16 or nch < 1:
                raise ValueError("only 16-bit PCM is supported (format tag %d, %d bits)" % (tag, bits))
            size = min(size, len(mv) - body)
            size -= size % (2 * nch)
            return np.frombuffer(mv[body:body + size], dtype="<i2"), nch, rate
        pos = body + size + (size & 1)
    raise ValueError("no data chunk")


def read_wav(path, device=None):
    """reference lidbox/features/audio.py:17-23: (signal [frames] float32 on the HIP device, sample_rate).  The file is read and
    its header parsed on the host; scaling and the channel average run in lidbox_pcm16_to_f32."""
    if isinstance(path, bytes):
        path = path.decode("utf-8")
    with open(path, "rb") as f:
        data = f.read()
    pcm, nch, rate = parse_wav_pcm16(data)
    return pcm16_to_float(torch.from_numpy(pcm.copy()), nch, device=device), rate


# ------------------------------------------------------------------ signal helpers
def dBFS_to_linear(level):
    """reference lidbox/features/audio.py:49-51"""
    return 10.0 ** (float(_scalar(level)) / 20.0)


def _one(signal):
    x = nv.require_gpu_tensor(signal, "signal", torch.float32)
    if x.dim() != 1:
        raise ValueError("signal must be rank 1")
    return signal_ops.RaggedSignals.from_list([x], device=x.device)


def peak_normalize(signal, dBFS=0):
    """reference lidbox/features/audio.py:57-59"""
    r = signal_ops.peak_normalize(_one(signal), float(_scalar(dBFS)))
    return r.split()[0].clone()


def root_mean_square(x, axis=-1):
    """reference lidbox/features/audio.py:266-270 for a rank-2 input reduced over its last axis"""
    x = nv.require_gpu_tensor(x, "x", torch.float32)
    if x.dim() != 2 or axis not in (-1, 1):
        raise ValueError("root_mean_square expects [rows, samples] reduced over the last axis")
    return signal_ops.signal_rms(signal_ops.RaggedSignals.from_dense(x))


def framewise_rms_energy_vad_decisions(signal, sample_rate, frame_step_ms, min_non_speech_ms=0, strength=0.05,
                                       min_rms_threshold=1e-3, time_axis=0):
    """reference lidbox/features/audio.py:308-329 -> bool [num_frames]"""
    if time_axis != 0:
        raise ValueError("only time_axis=0 (rank-1 signals) is supported")
    sample_rate = int(_scalar(sample_rate))
    frame_step = ms_to_frames(sample_rate, frame_step_ms)
    min_frames = int(ms_to_frames(sample_rate, min_non_speech_ms) / frame_step)                    # :325
    vad = signal_ops.vad_decisions(_one(signal), frame_step, min_frames, strength, min_rms_threshold)
    return vad["decisions"].to(torch.bool)


def remove_silence(signal, rate, window_ms=10, min_non_speech_ms=300):
    """reference lidbox/features/audio.py:337-353"""
    rate = int(_scalar(rate))
    window_frames = (int(window_ms) * rate) // 1000                                                # :341
    frame_step = ms_to_frames(rate, window_ms)
    if frame_step != window_frames:
        raise ValueError("window_ms * rate must be a whole number of samples")     # the reference would fail at frames[vad_1]
    r = _one(signal)
    min_frames = int(ms_to_frames(rate, min_non_speech_ms) / frame_step)
    vad = signal_ops.vad_decisions(r, frame_step, min_frames, strength=0.1)                        # :344-349
    return signal_ops.apply_vad(r, vad).split()[0].clone()


def snr_mixer(clean, noise, snr):
    """reference lidbox/features/audio.py:128-148 (rank-1 signals; `signal_ops.snr_mixer` takes batches)"""
    return signal_ops.snr_mixer(clean, noise, snr)
