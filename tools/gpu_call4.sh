cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c4; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 200 python tools/gemm_bench.py 1 > $O/gemm_base.log 2>&1
EGV_DESYNC=1 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_desync1.log 2>&1
EGV_DESYNC=2 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_desync2.log 2>&1
bash tools/gpu_prof.sh c4 bf16
