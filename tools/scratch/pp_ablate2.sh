for L in ppab7 ppab23 ppab55; do
  export LIDBOX_HIP_LIB=$PWD/tools/ab/lib$L.so
  echo "== $L"; BF16S_VARIANTS="256,256,2" python tools/bf16s_variants.py 256 2>&1 | grep -v "^call\|^sum" | awk '{printf "%s %s %s   ", $1,$2,$NF} END {print ""}'
done
unset LIDBOX_HIP_LIB
python - <<'PY'
# an empty kernel launched the same way, for the launch floor
import torch, time
x = torch.zeros(1, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): x.add_(1.0)
e1.record(); torch.cuda.synchronize(); print("tiny torch kernel back-to-back us", e0.elapsed_time(e1) / 50 * 1e3)
PY
