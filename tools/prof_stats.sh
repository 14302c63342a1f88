# usage (GPU box): bash tools/prof_stats.sh <tag> [bench.py args...]  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of one bench run)
T=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/p_$T/*/*kernel_trace.csv | head -1) gpurun_out/${T}_kernel_trace.csv; rm -rf gpurun_out/p_$T
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_$T -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-timing --no-feature-api "$@" > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cp $(ls gpurun_out/p_$T/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
python -c "import bench; print(bench.source_hash())" > gpurun_out/${T}_kernel_stats.csv.hash      # bench.py quotes the summary only next to the same kernel sources
cp $(ls gpurun_out/p_$T/*/*kernel_trace.csv | head -1) gpurun_out/${T}_kernel_trace.csv; rm -rf gpurun_out/p_$T
python tools/stats_table.py gpurun_out/${T}_kernel_stats.csv | head -30
