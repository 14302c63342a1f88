"""
GPU parity of the signal steps, the embedding-extraction step (8f.2) and the scoring utilities
(8f.4) against oracle/signal_np.py and the properties the reference's own tests hold
(reference tests/test_features_audio.py:59-66,157-191).

Decisions, slots, chunk layouts and gathered samples are exact (bit-equal); RMS / normalisation / mixer values
are fp32 vs the float64 oracle at rel 1e-5.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import features_np as fo
from oracle import signal_np as so

pytestmark = pytest.mark.gpu
AUDIO = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "audio", "*.wav")))


def _dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype))).cuda()


def _ragged_batch(rng, B, lo=0, hi=9000, silence=True):
    sigs = []
    for b in range(B):
        n = int(rng.integers(lo, hi))
        s = rng.standard_normal(n) * 0.2
        if silence and n > 800:                           # stretches of near silence of assorted lengths
            for _ in range(int(rng.integers(1, 5))):
                a = int(rng.integers(0, n - 1))
                s[a:a + int(rng.integers(50, 3000))] *= 1e-4
        sigs.append(s.astype(np.float32))
    return sigs


@pytest.mark.parametrize("frame_len,min_frames,strength", [(160, 0, 0.05), (160, 30, 0.1), (400, 3, 0.5), (7, 5, 0.3), (1, 2, 0.9)])
def test_vad_decisions_and_apply_match_oracle(frame_len, min_frames, strength):
    from lidbox_amd.features import signal_ops as sg
    rng = np.random.default_rng(frame_len * 31 + min_frames)
    sigs = _ragged_batch(rng, 23) + [np.zeros(0, np.float32), np.zeros(frame_len - 1, np.float32), np.zeros(5 * frame_len, np.float32)]
    r = sg.RaggedSignals.from_list(sigs)
    vad = sg.vad_decisions(r, frame_len, min_frames, strength)
    got = [d.cpu().numpy().astype(bool) for d in sg.split_frames(vad, vad["decisions"])]
    rms = [x.cpu().numpy() for x in sg.split_frames(vad, vad["rms"])]
    voiced = sg.apply_vad(r, vad)
    out = [v.cpu().numpy() for v in voiced.split()]
    for b, s in enumerate(sigs):
        frames = so.frame_nonoverlapping(s.astype(np.float64), frame_len)
        ref_rms = so.root_mean_square(frames, axis=1) if len(frames) else np.zeros(0)
        assert rms[b].shape == ref_rms.shape
        if len(ref_rms):
            assert np.abs(rms[b] - ref_rms).max() <= 1e-5 * max(1e-12, ref_rms.max())
            thr = strength * max(1e-3, ref_rms.mean())
            ref = so.invert_too_short_consecutive_false(ref_rms > thr, min_frames)
            safe = np.abs(ref_rms - thr) > 1e-5 * thr      # frames whose fp32 / fp64 comparison cannot differ
            if safe.all():
                assert (got[b] == ref).all(), b
            else:
                assert (got[b][safe] == (ref_rms > thr)[safe]).all() or min_frames > 0
        # applying the kernel's own decisions gathers exactly those frames, in order
        ref_out = frames[got[b]].reshape(-1).astype(np.float32) if len(frames) else np.zeros(0, np.float32)
        assert out[b].shape == ref_out.shape and (out[b] == ref_out).all(), b


def test_vad_reference_properties_on_fixtures():
    """reference tests/test_features_audio.py:175-191 through the reference-named functions"""
    from lidbox_amd.features import audio
    assert AUDIO
    for path in AUDIO:
        s, r = fo.read_wav_pcm16(path)
        x = _dev(s)
        vad = audio.framewise_rms_energy_vad_decisions(x, r, 25)
        assert vad.dtype == torch.bool and vad.shape == (len(s) // 400,) and bool(vad.all())
        ref = so.framewise_rms_energy_vad_decisions(s, r, 25)
        assert (vad.cpu().numpy() == ref).all()
        s1 = audio.remove_silence(x, r)
        assert s1.shape == x.shape and not torch.isnan(s1).any() and torch.equal(s1, x)
    z = torch.zeros(3 * 16000, device="cuda")
    assert not bool(audio.framewise_rms_energy_vad_decisions(z, 16000, 25).any())
    assert audio.remove_silence(z, 16000).numel() == 0


def test_remove_silence_matches_oracle_on_gappy_signal():
    from lidbox_amd.features import audio
    rng = np.random.default_rng(5)
    s = (rng.standard_normal(48000) * 0.3).astype(np.float32)
    s[4000:12000] *= 1e-5            # 0.5 s of silence: dropped (>= 300 ms)
    s[20000:22000] *= 1e-5           # 125 ms of silence: kept (shorter than min_non_speech_ms)
    got = audio.remove_silence(_dev(s), 16000).cpu().numpy()
    ref = so.remove_silence(s, 16000)
    assert got.shape == ref.shape and 0 < got.size < s.size and (got == ref).all()


def test_rms_peak_normalize_and_reference_bounds():
    """reference tests/test_features_audio.py:59-66,157-163"""
    from lidbox_amd.features import audio
    rng = np.random.default_rng(6)
    for _ in range(10):
        x = rng.normal(0, 5, size=rng.integers(1, 10, size=2))
        rms1 = np.sqrt(np.mean(np.square(np.abs(x)), axis=-1))
        assert np.abs(rms1 - audio.root_mean_square(_dev(x), axis=-1).cpu().numpy()).max() < 1e-5
    for path in AUDIO[:2]:
        s, r = fo.read_wav_pcm16(path)
        s1 = s + rng.normal(0, 10, s.shape)
        for level in (0, -3, -9):
            s2 = audio.peak_normalize(_dev(s1), dBFS=level).cpu().numpy()
            assert not np.isnan(s2).any()
            assert np.max(np.abs(s2)) <= audio.dBFS_to_linear(level) * (1 + 1e-6)
            assert np.abs(s2 - so.peak_normalize(s1, level)).max() <= 1e-5 * np.abs(s2).max()


@pytest.mark.parametrize("length_ms,step_ms,pad_ms", [(1000, 500, 0), (1000, 500, 400), (250, 100, 250), (10, 25, 10), (3, 1, 0)])
def test_signal_chunks_match_oracle(length_ms, step_ms, pad_ms):
    from lidbox_amd.features import signal_ops as sg
    rng = np.random.default_rng(length_ms + step_ms)
    lens = [0, 1, 15999, 16000, 16001, 23999, 24000, 40007, 8000, 7999] + [int(v) for v in rng.integers(0, 50000, size=12)]
    sigs = [(np.arange(n) % 977 + 1).astype(np.float32) * (1 + i) for i, n in enumerate(lens)]
    r = sg.RaggedSignals.from_list(sigs)
    chunks, nch = sg.signal_chunks(r, 16000, length_ms, step_ms, pad_ms)
    chunks = chunks.cpu().numpy()
    c0 = 0
    for i, s in enumerate(sigs):
        ref = so.create_signal_chunks(s, 16000, length_ms, step_ms, pad_ms)
        assert sg.chunk_plan(len(s), 16000, length_ms, step_ms, pad_ms) == so.signal_chunk_plan(len(s), 16000, length_ms, step_ms, pad_ms)
        assert nch[i] == ref.shape[0], (i, len(s))
        assert (chunks[c0:c0 + nch[i]] == ref).all(), i
        c0 += int(nch[i])
    assert c0 == chunks.shape[0]


@pytest.mark.parametrize("N", [32000, 4001, 16000, 40000, 4])      # register-resident (<= 32768, N % 4 == 0) and three-pass paths
def test_snr_mixer_matches_oracle(N):
    from lidbox_amd.features import audio, signal_ops as sg
    rng = np.random.default_rng(N)
    B = 5
    clean = rng.standard_normal((B, N)) * rng.uniform(0.01, 1.0, size=(B, 1))
    noise = rng.standard_normal((B, N)) * rng.uniform(0.001, 0.3, size=(B, 1))
    snr = np.array([-5.0, 0.0, 3.0, 12.5, 30.0])
    got = [g.cpu().numpy() for g in sg.snr_mixer(_dev(clean), _dev(noise), _dev(snr))]
    for b in range(B):
        ref = so.snr_mixer(clean[b], noise[b], snr[b])
        for g, rf in zip(got, ref):
            assert np.abs(g[b] - rf).max() <= 2e-5 * np.abs(rf).max()
    one = audio.snr_mixer(_dev(clean[1]), _dev(noise[1]), 0.0)
    assert one[2].shape == (N,) and torch.equal(one[2], torch.from_numpy(got[2][1]).cuda())
    with pytest.raises(ValueError):                       # audio.py:132
        audio.snr_mixer(_dev(clean[0]), _dev(noise[0][:-1]), 0.0)


def test_dataset_steps_vad_chunks_features_embeddings():
    """compute_rms_vad -> apply_vad -> create_signal_chunks -> extract_features -> extract_embeddings on a small
    ragged dataset, each step against the oracle (reference steps.py:183-200,417-432,579-632,674-705)"""
    from lidbox_amd.data import steps
    from lidbox_amd.models import xvector
    rng = np.random.default_rng(9)
    sigs = _ragged_batch(rng, 7, lo=20000, hi=60000)
    ds = [dict(id="utt%d" % i, signal=s, sample_rate=16000, duration=len(s) / 16000, target=i % 3) for i, s in enumerate(sigs)]
    with_vad = list(steps.compute_rms_vad(ds, strength=0.1, vad_frame_length_ms=10, min_non_speech_length_ms=100, launch_batch=3))
    assert [x["id"] for x in with_vad] == [x["id"] for x in ds]
    voiced = list(steps.apply_vad(with_vad, launch_batch=4))
    for x, v, s in zip(with_vad, voiced, sigs):
        assert "vad_is_speech" not in v and "vad_frame_length_ms" not in v and v["target"] == x["target"]
        ref = so.apply_vad(s, 16000, 10, x["vad_is_speech"].cpu().numpy())
        assert (v["signal"].cpu().numpy() == ref).all()
    chunks = list(steps.create_signal_chunks(voiced, 1000, 500, max_pad_ms=300, launch_batch=5))
    k = 0
    for v in voiced:
        ref = so.create_signal_chunks(v["signal"].cpu().numpy(), 16000, 1000, 500, 300)
        for c in range(ref.shape[0]):
            x = chunks[k]
            assert x["id"] == "%s-%06d" % (v["id"], c + 1) and abs(x["duration"] - 1.0) < 1e-6
            assert (x["signal"].cpu().numpy() == ref[c]).all()
            k += 1
    assert k == len(chunks) and k > 0
    feats = list(steps.extract_features(chunks, {"type": "logmelspectrogram", "batch_size": 4}))
    assert feats[0]["input"].shape == (98, 40)
    m1, m2 = xvector.create((98, 40), 3, seed=1), xvector.create((98, 40), 3, seed=2)
    ex = [xvector.as_embedding_extractor(m1), m2]
    out = list(steps.extract_embeddings(feats, {"extractors": ex, "batch_size": 3}))
    assert len(out) == len(feats)
    X = torch.stack([f["input"] for f in feats])
    ref = torch.cat([m1.embed(X), m2.embed(X)], dim=1)
    got = torch.stack([o["embedding"] for o in out])
    assert got.shape == (len(feats), 1024)
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=1)
    assert float(cos.min()) >= 0.9999                   # batch composition only changes GEMM tiling
    batched = list(steps.extract_embeddings(feats, {"extractors": ex, "batch_size": 4, "no_unbatch": True}))
    assert batched[0]["embedding"].shape == (4, 1024) and len(batched[0]["id"]) == 4


def test_merge_chunk_predictions_and_classification_report():
    """reference util.py:41-57 and :60-105 (C_avg on the device, the rest through sklearn as in the reference)"""
    import sklearn.metrics
    from lidbox_amd import util
    from oracle import model_np as mo
    rng = np.random.default_rng(11)
    ids, preds = [], []
    for u, n in (("b-utt", 3), ("a-utt", 1), ("c-x-utt", 5)):
        for c in range(n):
            ids.append("%s-%06d" % (u, c + 1))
            preds.append(rng.standard_normal(4).astype(np.float32))
    df = util.predictions_to_dataframe(ids, preds)
    merged = util.merge_chunk_predictions(df)
    assert list(merged.index) == ["a-utt", "b-utt", "c-x-utt"]
    for u in merged.index:
        ref = np.stack([p for i, p in zip(ids, preds) if util.chunk_parent_id(i) == u]).mean(axis=0)
        assert np.abs(merged.loc[u].prediction - ref).max() < 1e-6
    custom = util.merge_chunk_predictions(df, merge_rows_fn=lambda v: np.stack(v).max(axis=0))
    assert np.allclose(custom.loc["b-utt"].prediction, np.stack(preds[:3]).max(axis=0))
    # classification report
    N, n = 4, 200
    true = rng.integers(0, N, size=n)
    pred = rng.standard_normal((n, N)).astype(np.float32)
    pred[np.arange(n), true] += 1.5
    l2t = {"lang%d" % i: i for i in range(N)}
    rep = util.classification_report(true, pred, l2t)
    cavg = mo.SparseAverageDetectionCost(N, np.linspace(pred.min(), pred.max(), 100))
    cavg.update_state(true, pred)
    assert abs(rep["avg_detection_cost"] - cavg.result()) < 1e-6
    assert (rep["confusion_matrix"] == sklearn.metrics.confusion_matrix(true, pred.argmax(1))).all()
    assert 0.0 <= rep["avg_equal_error_rate"] <= 0.5 and "equal_error_rate" in rep["lang2"]
    assert abs(rep["accuracy"] - (pred.argmax(1) == true).mean()) < 1e-12


@pytest.mark.parametrize("frames,channels", [(0, 1), (1, 1), (7, 1), (48000, 1), (100003, 1), (48000, 2), (1001, 3), (513, 6)])
def test_pcm16_ingest_is_bit_exact(frames, channels):
    """lidbox_pcm16_to_f32 (reference audio.py:17-23: decode_wav's int16 / 32768, then the channel mean) against the float32
    arithmetic restated in numpy -- every sample value incl. -32768 / 32767, odd lengths, unaligned views, several channels"""
    from lidbox_amd.features import audio
    rng = np.random.default_rng(frames + channels)
    pcm = rng.integers(-32768, 32768, size=(frames, channels)).astype(np.int16)
    if frames >= 2:
        pcm[0, :] = -32768
        pcm[1, :] = 32767
    ref = (pcm.astype(np.float32) / np.float32(32768.0)).mean(axis=1, dtype=np.float32) if frames else np.zeros(0, np.float32)
    got = audio.pcm16_to_float(pcm, channels)
    assert got.dtype == torch.float32 and got.shape == (frames,)
    assert np.array_equal(got.cpu().numpy(), ref)
    if channels == 1 and frames > 16:                          # a view that starts 2 bytes off a 16-byte boundary
        buf = torch.from_numpy(np.concatenate([[0], pcm[:, 0]]).astype(np.int16)).cuda()
        assert np.array_equal(audio.pcm16_to_float(buf[1:], 1).cpu().numpy(), ref)


def test_read_wav_matches_oracle_on_reference_fixtures(tmp_path):
    """audio.read_wav (host RIFF parse + device ingest) == the oracle's decode_wav restatement on the reference's own WAV files, a
    stereo file written here, and a file with an extra chunk ahead of `data`"""
    import struct
    import wave
    from lidbox_amd.features import audio
    for p in AUDIO:
        s, r = audio.read_wav(p)
        so_, ro = fo.read_wav_pcm16(p)
        assert r == ro and np.array_equal(s.cpu().numpy(), so_)
    rng = np.random.default_rng(5)
    st = rng.integers(-32768, 32768, size=(12345, 2)).astype("<i2")
    p2 = str(tmp_path / "stereo.wav")
    with wave.open(p2, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(22050); f.writeframes(st.tobytes())
    s, r = audio.read_wav(p2.encode("utf-8"))
    so_, ro = fo.read_wav_pcm16(p2)
    assert r == ro == 22050 and np.array_equal(s.cpu().numpy(), so_)
    raw = open(p2, "rb").read()
    extra = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"             # odd-sized chunk + its pad byte
    i = raw.index(b"data")
    p3 = str(tmp_path / "list.wav")
    open(p3, "wb").write(raw[:i] + extra + raw[i:])
    s3, _ = audio.read_wav(p3)
    assert torch.equal(s3, s)
    with pytest.raises(ValueError):
        audio.parse_wav_pcm16(b"RIFF\x00\x00\x00\x00WAVEjunk")
    with pytest.raises(ValueError):
        audio.parse_wav_pcm16(b"not a wav file at all")
