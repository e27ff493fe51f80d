cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c5; mkdir -p $O
EGV_DESYNC=100 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_nostore.log 2>&1
BENCH_OUT=bf16 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_bf16out.log 2>&1
