cd $GRAFT_REPO_ROOT
O=gpurun_out/c20; mkdir -p $O
for i in 1 2; do
timeout 200 python tools/gemm_bench.py 1 2>&1 | grep TOTAL >> $O/ab.log
EGV_GEMM_DBG=7 timeout 200 python tools/gemm_bench.py 1 2>&1 | grep TOTAL >> $O/ab.log
done
EGV_GEMM_DBG=7 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_prio.log 2>&1
