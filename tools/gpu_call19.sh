cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c19; mkdir -p $O
( EGV_GEMM_KERNEL=6 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -15 ) > $O/pytest_duo.log 2>&1
EGV_GEMM_KERNEL=6 timeout 200 python tools/gemm_bench.py > $O/gemm_duo.log 2>&1
timeout 200 python tools/gemm_bench.py > $O/gemm_big.log 2>&1
EGV_GEMM_KERNEL=6 timeout 100 python tools/gemm_trace.py 25120 768 768 > $O/trace_duo_proj.log 2>&1
EGV_GEMM_KERNEL=6 timeout 100 python tools/gemm_trace.py 25120 768 3072 > $O/trace_duo_fc2.log 2>&1
