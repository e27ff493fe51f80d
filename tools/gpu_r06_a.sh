# round 6, session a: baseline of the round-5 tree + the new one-GPU W > 1 exchange test + time attention with the wave barriers
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_gradsync.py tests/test_gpu_ops.py tests/test_gpu_block.py -m gpu -q 2>&1 | grep -v "amdgpu\|^$" | tail -30 ) > $O/pytest_subset.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/pytest_subset.txt; cut -c1-400 $O/bench_default.json
