# usage: bash tools/gpu_final2.sh <tag> : GPU tests + smoke + the default bench line on the final code
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -200 ) > $O/pytest_gpu.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/pytest_gpu.txt; cat $O/smoke.log; cut -c1-330 $O/bench_default.json
