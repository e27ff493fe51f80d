cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c1
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c1/pytest.log 2>&1
for v in 1 2; do EGV_GEMM_KERNEL=$v timeout 200 python tools/gemm_bench.py > gpurun_out/c1/gemm_v$v.log 2>&1; done
EGV_GEMM_KERNEL=2 EGV_PINGPONG=1 timeout 200 python tools/gemm_bench.py > gpurun_out/c1/gemm_v2pp.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16 > gpurun_out/c1/bench_bf16.json 2> gpurun_out/c1/bench_bf16.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision mixed > gpurun_out/c1/bench_mixed.json 2> gpurun_out/c1/bench_mixed.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1/prof_bf16 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --precision bf16 > $GRAFT_REPO_ROOT/gpurun_out/c1/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/c1/prof_bf16 -name "*.db" -delete; find gpurun_out/c1/prof_bf16 -name "*kernel_trace*" -delete
nproc > gpurun_out/c1/nproc.txt
