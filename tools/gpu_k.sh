# usage: bash tools/gpu_k.sh <tag>: GEMM tests, gemm microbench with slab vs atomic split-K, short bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -5 ) > $O/pytest_gemm.log 2>&1
timeout 200 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_slab.log
BENCH_SPLITK=atomic timeout 200 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_atomic.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err
cat $O/pytest_gemm.log; paste <(cut -c1-75 $O/gemm_slab.log) <(cut -c46-75 $O/gemm_atomic.log); tail -c 1500 $O/bench.json
