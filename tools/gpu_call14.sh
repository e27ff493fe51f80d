cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c14; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
( time timeout 900 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
