cd $GRAFT_REPO_ROOT
O=gpurun_out/c28; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -6 ) > $O/pytest_gemm.log 2>&1
if grep -q "passed" $O/pytest_gemm.log && ! grep -q "failed" $O/pytest_gemm.log; then
  timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -E "main loop|span|prologue|epilogue" > $O/trace_fc2.log
  timeout 100 python tools/gemm_trace.py 25120 2304 768 2>&1 | grep -E "main loop|span" > $O/trace_qkv.log
  timeout 100 python tools/gemm_trace.py 2304 768 25120 tn 9 2>&1 | grep -E "main loop|span" > $O/trace_tn.log
  timeout 200 python tools/gemm_bench.py > $O/gemm.log 2>&1
  ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest_all.log 2>&1
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>/dev/null
fi
