cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c7; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 200 python tools/gemm_bench.py > $O/gemm.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 768 768 > $O/trace_proj.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 2304 768 > $O/trace_qkv.log 2>&1
bash tools/gpu_prof.sh c7 bf16
