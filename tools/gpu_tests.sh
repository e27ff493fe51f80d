# the whole GPU test suite + smoke: bash tools/gpu_tests.sh <tag> [pytest args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; shift
( timeout 2400 python -m pytest tests -m gpu -q -s "$@" 2>&1 | grep -v "amdgpu\|^$" | tail -400 ) > $O/pytest_gpu.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) >> $O/pytest_gpu.txt 2>&1
grep -n "passed\|failed\|Error\|smoke" $O/pytest_gpu.txt | tail -20
