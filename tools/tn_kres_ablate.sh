# usage (GPU box): bash tools/tn_kres_ablate.sh [variants...]   kernel-only durations (rocprofv3) of frame1's bf16 wgrad: the tree's library, then
# tools/ab_ship/libtkr<N>.so (python tools/ab_build.py tkrN gemm_bf16.hip -DLBX_TKR_ABLATE=N with LIDBOX_AB_DIR=ab_ship)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_tkr -- python tools/tn_kres_time.py 512 > /dev/null 2>&1
  python tools/stats_table.py $(ls gpurun_out/p_tkr/*/*kernel_stats.csv | head -1) | grep -i "tn_kres"; rm -rf gpurun_out/p_tkr
}
echo "== tree"; run
for v in "$@"; do echo "== ablate $v"; LIDBOX_HIP_LIB=$PWD/tools/ab_ship/libtkr$v.so run; done
