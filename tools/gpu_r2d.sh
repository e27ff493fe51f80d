# round-2 call D: GPU tests (all) + bench + fill attribution
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -250 ) > $O/pytest_gpu.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python tools/find_fills.py bf16 2>&1 | grep -v amdgpu | head -70 > $O/find_fills.txt
grep -c . $O/pytest_gpu.log; tail -40 $O/pytest_gpu.log; cat $O/bench_default.json | cut -c1-600; head -40 $O/find_fills.txt
