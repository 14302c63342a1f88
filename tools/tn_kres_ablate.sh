cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_tkr -- python tools/tn_kres_time.py 512 > /dev/null 2>&1
python tools/stats_table.py $(ls gpurun_out/p_tkr/*/*kernel_stats.csv | head -1) | grep -i "tn_k\|reduce\|tn_pp" ; rm -rf gpurun_out/p_tkr
for v in 1 3; do echo "== ablate $v"; LIDBOX_HIP_LIB=$PWD/tools/ab_ship/libtkr$v.so python tools/tn_kres_time.py 512 2>&1 | grep "K1-resident"; done
