cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c2; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
EGV_GEMM_KERNEL=3 timeout 300 python tools/gemm_bench.py > $O/gemm_auto.log 2>&1
EGV_GEMM_KERNEL=4 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_mf4.log 2>&1
EGV_GEMM_KERNEL=14 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_mf4nc4.log 2>&1
EGV_GEMM_KERNEL=5 timeout 200 python tools/gemm_bench.py 1 > $O/gemm_mf5.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision mixed > $O/bench_mixed.json 2> $O/bench_mixed.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --precision bf16 > $GRAFT_REPO_ROOT/$O/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_bf16 -name "*stats*.csv" -exec cp {} $O/ \;
ls -R /tmp/prof_bf16 > $O/prof_ls.txt 2>&1
