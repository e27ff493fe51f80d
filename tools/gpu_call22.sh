cd $GRAFT_REPO_ROOT
O=gpurun_out/c22; mkdir -p $O
for d in 0 0x1000 0x2000 0x3000; do
echo "== diag $d" >> $O/diag.log
TRACE_DIAG=$d timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -E "main loop|span" >> $O/diag.log
done
