# isolated attention timings under env variants: bash tools/gpu_attn_ab.sh <tag> "<env1>" "<env2>" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; shift
( timeout 600 python -m pytest tests -m gpu -q -x -k "attention or attn" 2>&1 | tail -3 ) > $O/pytest_attn.txt 2>&1
cat $O/pytest_attn.txt
for rep in 1 2; do
  for e in "$@"; do
    echo "== [$e] rep $rep" >> $O/attn.txt
    ( env $e timeout 200 python tools/attn_time.py 2>&1 | grep -v amdgpu ) >> $O/attn.txt
  done
done
cat $O/attn.txt
