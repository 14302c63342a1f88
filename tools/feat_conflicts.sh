for L in featnew feat_NOMEL feat_NOEXCH; do
  echo "== $L"
  LIDBOX_HIP_LIB=$PWD/tools/ab_ship/lib$L.so bash tools/pmc_one.sh cf "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" -- python tools/feat_one.py 2048
  grep feat512 gpurun_out/pmc_cf.txt | sed 's/^[^ ]* *//'
done
