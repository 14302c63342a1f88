"""
Generates the committed golden vectors under tests/golden/ from the ORACLE (oracle/), float64
math cast to float32.  The reference itself cannot run here (TensorFlow is not importable), so
these pin the build against regressions of the restatement, not against TensorFlow output --
"parity unpinned" in oracle/__init__.py applies.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import features_np as fo  # noqa: E402
from oracle import model_np as mo     # noqa: E402
from lidbox_amd.testutil import synthetic_batch  # noqa: E402

WAVS = ["noisy_100hz_sine.wav", "noisy_200hz_sine.wav", "noisy_300hz_sine.wav", "noisy_400hz_sine.wav",
        "noise.wav"]


def main():
    # 1. the reference's own WAV fixtures -> log-mel, MFCC(1:13)+CMVN
    sigs = np.stack([fo.read_wav_pcm16(os.path.join(HERE, "audio", w))[0] for w in WAVS])
    sr = [16000] * len(sigs)
    logmel = fo.extract_features(sigs, sr, "logmelspectrogram")
    mfcc = fo.extract_features(sigs, sr, "mfcc", window_norm_kwargs=dict(window_len=-1, normalize_variance=True))
    np.savez_compressed(os.path.join(HERE, "features_wav.npz"), logmel=logmel.astype(np.float32),
                        mfcc_cmvn=mfcc.astype(np.float32))
    # 2. seeded synthetic batch (SURVEY 8d recipe) -> log-mel -> x-vector log-probs / embedding / loss / grads
    sig, y = synthetic_batch(4, num_labels=4)
    x = fo.extract_features(sig, [16000] * 4, "logmelspectrogram")
    p = {k: v.astype(np.float64) for k, v in mo.xvector_init(40, 4, seed=0).items()}
    loss, g, logp = mo.xvector_loss_and_grads(p, x, y)
    emb = mo.xvector_fwd(p, x, embedding=True)
    np.savez_compressed(os.path.join(HERE, "xvector_synth.npz"), labels=y, logmel=x.astype(np.float32),
                        logp=logp.astype(np.float32), embedding=emb.astype(np.float32),
                        loss=np.float32(loss),
                        grad_norms=np.array([np.linalg.norm(g[k]) for k in sorted(g)], np.float32),
                        grad_names=np.array(sorted(g)))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
