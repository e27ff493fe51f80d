# usage: bash tools/gpu_quick.sh <tag>: GEMM tests + tile traces + gemm microbench (about 1 GPU-minute)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3 ) > $O/pytest_gemm.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -E "main loop|span|prologue|epilogue" > $O/trace_fc2.log
timeout 100 python tools/gemm_trace.py 25120 2304 768 2>&1 | grep -E "main loop|span" > $O/trace_qkv.log
timeout 100 python tools/gemm_trace.py 2304 768 25120 tn 9 2>&1 | grep -E "main loop|span" > $O/trace_tn.log
timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $O/gemm.log
cat $O/pytest_gemm.log $O/trace_fc2.log $O/trace_qkv.log $O/trace_tn.log; grep -E "passes=1|TOTAL" $O/gemm.log
