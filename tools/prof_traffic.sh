# usage (GPU box): bash tools/prof_traffic.sh <tag> [bench.py args...] -> gpurun_out/<tag>_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes)
T=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf_$T gpurun_out/pw_$T
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf_$T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pw_$T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing "$@" > /dev/null 2>&1
python tools/traffic_from_pmc.py gpurun_out/pf_$T gpurun_out/pw_$T gpurun_out/${T}_traffic.json | head -16
rm -rf gpurun_out/pf_$T gpurun_out/pw_$T
