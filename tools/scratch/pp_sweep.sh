python -m pytest tests/test_gemm_bf16_gpu.py -m gpu -q -k "storage_gemm_nt_and_shadow or shadow_only" 2>&1 | tail -3
BF16S_VARIANTS="policy;256,256,2;256,256,1;256,128,2;256,128,1" python tools/bf16s_variants.py 256 2>&1
echo "== bs512"; BF16S_VARIANTS="policy;256,256,2;256,128,2" python tools/bf16s_variants.py 512 2>&1
