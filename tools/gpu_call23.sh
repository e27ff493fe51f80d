cd $GRAFT_REPO_ROOT
O=gpurun_out/c23; mkdir -p $O
for d in 0 0x4000 0x8000; do
echo "== diag $d" >> $O/diag.log
TRACE_DIAG=$d timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -E "main loop|span" >> $O/diag.log
TRACE_DIAG=$d timeout 100 python tools/gemm_trace.py 25120 2304 768 2>&1 | grep -E "main loop|span" >> $O/diag.log
TRACE_DIAG=$d timeout 100 python tools/gemm_trace.py 2304 768 25120 tn 9 2>&1 | grep -E "main loop|span" >> $O/diag.log
done
