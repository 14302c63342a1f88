# shader clock during the bf16 ping-pong tiles: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / kernel duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_clk
BF16S_VARIANTS="256,256,2" rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/_clk -- python tools/bf16s_variants.py 512 > /dev/null 2>&1
python tools/pmc_dispatches.py gpurun_out/_clk gemm16s | cut -c1-200
rm -rf gpurun_out/_clk2
TN_LAYERS=frame2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/_clk2 -- python tools/scratch/tn_pp_time.py 512 > /dev/null 2>&1
python tools/pmc_dispatches.py gpurun_out/_clk2 gemm16s | cut -c1-200
rm -rf gpurun_out/_clk gpurun_out/_clk2
