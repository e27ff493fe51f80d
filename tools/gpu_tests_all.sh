# the whole GPU suite + smoke + the default bench line: bash tools/gpu_tests_all.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -400 ) > $O/pytest_gpu.txt 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -8 ) > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
grep -n "passed\|failed\|FAILED\|Error" $O/pytest_gpu.txt | tail -30; cat $O/smoke.log; cut -c1-600 $O/bench_default.json
