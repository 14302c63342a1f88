set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err
python bench.py --compute-dtype bfloat16 > gpurun_out/r01_bench_bf16.json 2>> gpurun_out/r01_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(ls gpurun_out/p_stats/*/*kernel_stats.csv | head -1) gpurun_out/r01_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats16 -- python bench.py --compute-dtype bfloat16 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(ls gpurun_out/p_stats16/*/*kernel_stats.csv | head -1) gpurun_out/r01_kernel_stats_bf16.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/traffic_from_pmc.py gpurun_out/p_fetch gpurun_out/p_write gpurun_out/r01_traffic.json | head -8
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_mfma -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/p_mfma > gpurun_out/r01_pmc_mfma_raw.txt
python tools/mfma_util_from_pmc.py gpurun_out/r01_pmc_mfma_raw.txt gpurun_out/r01_mfma_util.json | head -8
python tools/bench_configs.py > gpurun_out/r01_configs.jsonl 2>/dev/null
rm -rf gpurun_out/p_stats gpurun_out/p_stats16 gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_mfma
cut -c1-600 gpurun_out/r01_bench.json; cut -c1-300 gpurun_out/r01_bench_bf16.json
