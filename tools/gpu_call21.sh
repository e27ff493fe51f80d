cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c21; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 200 python tools/gemm_bench.py 3 > $O/gemm_x3_fused.log 2>&1
EGV_X3_FUSED=0 timeout 200 python tools/gemm_bench.py 3 > $O/gemm_x3_seg.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --precision mixed > $O/bench_mixed_fused.json 2>/dev/null
EGV_X3_FUSED=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --precision mixed > $O/bench_mixed_seg.json 2>/dev/null
