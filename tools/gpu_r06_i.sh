cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_f16x2.py tests/test_gpu_f16bwd.py -m gpu -q 2>&1 | grep -v "amdgpu\|^$" | tail -12 ) > $O/pytest_subset.txt 2>&1
tail -5 $O/pytest_subset.txt
for B in f16; do
  rm -rf /tmp/prof_$B
  ( cd /tmp && EGV_X2_BWD=$B timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$B -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err --wgrad-side 0 --text-side 0 ) > $O/prof_$B.log 2>&1
  f=$(find /tmp/prof_$B -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_f16mix_$B.csv >> $O/prof_$B.log 2>&1
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err > $O/bench.json 2>$O/bench.err; cut -c1-300 $O/bench.json
