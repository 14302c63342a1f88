# usage (on the MI355X box): bash tools/profile_round.sh r02   -> gpurun_out/<tag>_* (copy what is to be judged into profiles/)
set -x
T=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-timing --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
cp $(ls gpurun_out/p_stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
HASH=$(python -c "import bench; print(bench.source_hash())")      # bench.py quotes a summary only next to the kernel sources it was taken from
echo $HASH > gpurun_out/${T}_kernel_stats.csv.hash
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats16 -- python bench.py --compute-dtype bfloat16 --steps 100 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-timing --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
cp $(ls gpurun_out/p_stats16/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats_bf16.csv
echo $HASH > gpurun_out/${T}_kernel_stats_bf16.csv.hash
# configs[3] / [4]: stats + the kernel sequence of one step (before the bench lines: their roofline.frac quotes these summaries)
for C in 3 4; do bash tools/prof_stats.sh ${T}c$C --steps 100 --config $C --sustain-seconds 0 > /dev/null 2>&1; python tools/step_seq.py gpurun_out/${T}c${C}_kernel_trace.csv > gpurun_out/${T}_step_seq_config$C.txt; cp gpurun_out/${T}c${C}_kernel_stats.csv gpurun_out/${T}_kernel_stats_config$C.csv; echo $HASH > gpurun_out/${T}_kernel_stats_config$C.csv.hash; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
python tools/traffic_from_pmc.py gpurun_out/p_fetch gpurun_out/p_write gpurun_out/${T}_traffic.json | head -8
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch16 -- python bench.py --compute-dtype bfloat16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write16 -- python bench.py --compute-dtype bfloat16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
python tools/traffic_from_pmc.py gpurun_out/p_fetch16 gpurun_out/p_write16 gpurun_out/${T}_traffic_bf16.json | head -8
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_mfma -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/p_mfma > gpurun_out/${T}_pmc_mfma_raw.txt
python tools/mfma_util_from_pmc.py gpurun_out/${T}_pmc_mfma_raw.txt gpurun_out/${T}_mfma_util.json | head -8
for C in 3 4; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetchc$C -- python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_writec$C -- python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-feature-api --sustain-seconds 0 > /dev/null 2>&1
  python tools/traffic_from_pmc.py gpurun_out/p_fetchc$C gpurun_out/p_writec$C gpurun_out/${T}_traffic_config$C.json | head -4
  rm -rf gpurun_out/p_fetchc$C gpurun_out/p_writec$C
done
python tools/bench_configs.py > gpurun_out/${T}_configs.jsonl 2>/dev/null
python tools/step_calls.py float32 > gpurun_out/${T}_step_calls_fp32.txt 2>/dev/null
python tools/step_calls.py bfloat16 > gpurun_out/${T}_step_calls_bf16.txt 2>/dev/null
python tools/bench_bf16s.py 256 > gpurun_out/${T}_bf16_storage_gemm.txt 2>/dev/null
python tools/bench_features.py > gpurun_out/${T}_bench_features.txt 2>/dev/null
bash tools/feat_counters.sh ${T} > /dev/null 2>&1
# the bench lines last: their `roofline.traffic` quotes the PMC summaries of THIS build (matched by source hash), which have to
# sit in profiles/ when bench.py runs (on the box; the copies that are committed come from gpurun_out/)
: > gpurun_out/${T}_bench.err
cp gpurun_out/${T}_traffic.json gpurun_out/${T}_traffic_bf16.json gpurun_out/${T}_traffic_config3.json gpurun_out/${T}_traffic_config4.json profiles/
cp gpurun_out/${T}_kernel_stats*.csv gpurun_out/${T}_kernel_stats*.csv.hash profiles/
python bench.py > gpurun_out/${T}_bench.json 2>> gpurun_out/${T}_bench.err
python bench.py --compute-dtype bfloat16 > gpurun_out/${T}_bench_bf16.json 2>> gpurun_out/${T}_bench.err
python bench.py --config 3 > gpurun_out/${T}_bench_config3.json 2>> gpurun_out/${T}_bench.err
python bench.py --config 4 > gpurun_out/${T}_bench_config4.json 2>> gpurun_out/${T}_bench.err
bash tools/prof_stats.sh ${T}s32 --steps 100 --sustain-seconds 0 > /dev/null 2>&1; python tools/step_seq.py gpurun_out/${T}s32_kernel_trace.csv > gpurun_out/${T}_step_seq_fp32.txt
bash tools/prof_stats.sh ${T}s16 --steps 100 --compute-dtype bfloat16 --sustain-seconds 0 > /dev/null 2>&1; python tools/step_seq.py gpurun_out/${T}s16_kernel_trace.csv > gpurun_out/${T}_step_seq_bf16.txt
rm -f gpurun_out/${T}s32_* gpurun_out/${T}s16_* gpurun_out/${T}c3_* gpurun_out/${T}c4_*
rm -rf gpurun_out/p_stats gpurun_out/p_stats16 gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_fetch16 gpurun_out/p_write16 gpurun_out/p_mfma
cut -c1-600 gpurun_out/${T}_bench.json; cut -c1-300 gpurun_out/${T}_bench_bf16.json
