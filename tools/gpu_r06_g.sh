cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for S in 0 2; do EGV_X2_SEG=$S timeout 300 python tools/x2seg_bench.py 2>&1 | grep EGV_X2; done; done | tee $O/x2seg_bench.txt
( timeout 300 python -m pytest tests/test_gpu_f16x2.py -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_f16x2_seg.txt
