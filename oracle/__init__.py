"""
oracle/ -- CPU restatement of the lidbox hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under lidbox_amd/ imports it; the product path fails
loudly when the HIP extension is missing instead of falling back to this code.

PARITY UNPINNED (values): the reference's arithmetic lives in TensorFlow
(`tensorflow ~= 2.3.0`, /root/reference/requirements-test.txt:4), which is not
importable in the build container and has no wheel for this Python.  The
reference's own tests pin shape laws, sanity bounds and three exact/numeric
identities on this path (ms_to_frames, log10, feature_scaling) -- those are all
restated in tests/test_oracle.py -- but no log-mel / MFCC / x-vector / loss /
C_avg VALUES.  What pins the restatement instead:
  * the published TF-op semantics, restated function by function below with the
    reference call site (file:line) each one follows;
  * closed-form known answers derived from the reference's own demo data
    (losses.py:61-97, metrics.py:127-151): see tests/test_oracle.py;
  * independent cross-checks against numpy.fft / scipy.fft.dct /
    torch.nn.functional.conv1d / torch autograd (oracle/torch_ref.py);
  * the reference's own 16 kHz WAV fixtures (tests/golden/audio/*.wav).

Modules:
  features_np  numpy (float64 truth, or float32 op-order-faithful) feature path
  model_np     numpy x-vector / CNN forward + hand-derived backward, Adam, losses, C_avg
  torch_ref    torch-CPU fp32 restatement (autograd) -- gradient cross-check and
               the timed cpu_baseline "port"
"""
