#!/bin/bash
# correctness of every LDS-DMA variant of the bf16-storage rows kernel, then per-call timings of one bf16 step
for v in 64,64,2 64,64,3 128,64,2 64,128,2 128,128,2 128,128,3; do
  echo "== tests $v"; LIDBOX_GEMM16S_DMA=$v python -m pytest tests/test_gemm_bf16_gpu.py -x -q -m gpu -k "storage_gemm_nt or storage_gemm_implicit or shadow_only" 2>&1 | tail -2
done
for v in 0 64,64,2 64,64,3 64,64,4 128,64,2 128,64,3 64,128,2 128,128,2 128,128,3; do
  echo "== step_calls $v"; LIDBOX_GEMM16S_DMA=$v python tools/step_calls.py bfloat16 2>/dev/null | grep -E "bf16s_nt|total"
done
