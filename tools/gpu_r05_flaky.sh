cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for rep in 1 2 3 4 5 6; do
  echo "rep=$rep" >> $O/flaky.txt
  ( timeout 300 python -m pytest tests/test_gpu_block.py -q -s -k "fp16_modes or equal_the_per_kernel_path" 2>&1 | grep "passed\|failed\|AssertionError: \|block calls vs per-kernel\|FAILED" | cut -c1-700 ) >> $O/flaky.txt 2>&1
done
cat $O/flaky.txt
