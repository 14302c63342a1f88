# usage (GPU box): bash tools/feat_counters.sh <tag>   -> gpurun_out/<tag>_feature_counters.txt
# SQ counters of the log-mel feature kernel alone (tools/feat_one.py), one rocprofv3 --pmc pass per counter group, B = 256 and 2048.
T=${1:-r06}
OUT=gpurun_out/${T}_feature_counters.txt
: > $OUT
for B in 256 2048; do
  echo "== log-mel, B = $B (per launch means; SQ_* cycle counters are in quad-cycles per the guide)" >> $OUT
  i=0
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    bash tools/pmc_one.sh fc$i "$G" -- python tools/feat_one.py $B
    grep "feat512_stream_kernel" gpurun_out/pmc_fc$i.txt | sed 's/^[^ ]* *//' >> $OUT
    rm -f gpurun_out/pmc_fc$i.txt
  done
done
cat $OUT
