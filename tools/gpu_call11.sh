cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c11; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 100 python tools/gemm_trace.py 2304 768 25120 tn 9 > $O/trace_tn_qkv.log 2>&1
timeout 100 python tools/gemm_trace.py 768 768 25120 tn 28 > $O/trace_tn_proj.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 768 3072 > $O/trace_fc2.log 2>&1
bash tools/gpu_prof.sh c11 bf16
