for L in base ppab1 ppab2 ppab4 ppab3; do
  if [ $L = base ]; then unset LIDBOX_HIP_LIB; else export LIDBOX_HIP_LIB=$PWD/tools/ab/lib$L.so; fi
  echo "== $L"; BF16S_VARIANTS="256,256,2" python tools/bf16s_variants.py 256 2>&1 | grep -v "^call\|^sum" | awk '{printf "%s %s %s   ", $1,$2,$NF} END {print ""}'
done
