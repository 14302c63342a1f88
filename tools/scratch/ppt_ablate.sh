# ablation of the ping-pong wgrad tile: base, then LBX_PPT_ABLATE builds (tools/ab/libppt<f>.so)
for L in base ppt1 ppt64 ppt32 ppt2 ppt4 ppt6 ppt16; do
  if [ $L = base ]; then unset LIDBOX_HIP_LIB; else export LIDBOX_HIP_LIB=$PWD/tools/ab/lib$L.so; fi
  echo "== $L"; TN_LAYERS=frame2,frame3 TN_MODES=1 python tools/scratch/tn_pp_time.py ${1:-512} 2>&1 | grep "frame2\|frame3" | sed 's/ (ws[^)]*)//g; s/|dW.*//'
done
