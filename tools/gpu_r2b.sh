# round-2 call B: full GPU test suite (new tests print their measured errors) + default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -120 ) > $O/pytest_gpu.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -60 $O/pytest_gpu.log; cat $O/bench_default.json; tail -3 $O/bench_default.err
