cd $GRAFT_REPO_ROOT
O=gpurun_out/c24; mkdir -p $O
( EGV_GEMM_DBG=65536 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -4 ) > $O/pytest_m1.log 2>&1
( EGV_GEMM_DBG=131072 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -4 ) > $O/pytest_m2.log 2>&1
for i in 1 2; do
for d in 0 65536 131072; do
echo "== mode $d" >> $O/bench.log
EGV_GEMM_DBG=$d timeout 200 python tools/gemm_bench.py 1 2>&1 | grep -E "fc2   fwd|qkv   fwd|qkv wgrad|TOTAL" >> $O/bench.log
done
done
for d in 0 0x10000 0x20000; do
echo "== diag $d" >> $O/diag.log
TRACE_DIAG=$d timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -E "main loop|span" >> $O/diag.log
done
