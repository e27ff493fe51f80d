cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c6; mkdir -p $O
timeout 100 python tools/gemm_trace.py 25120 768 768 > $O/trace_proj.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 768 3072 > $O/trace_fc2.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 2304 768 > $O/trace_qkv.log 2>&1
