# round-2 call C: full GPU test suite + default bench (new boundary / dist / trainer tests included)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -150 ) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -70 $O/pytest_gpu.log; cat $O/bench_default.json; tail -3 $O/bench_default.err
