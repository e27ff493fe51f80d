# round 6, session e: kernel statistics (timed steps only, one stream) of the bf16 and the fp16 backward
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for B in bf16 f16; do
  rm -rf /tmp/prof_$B
  ( cd /tmp && EGV_X2_BWD=$B timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$B -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 --text-side 0 ) > $O/prof_$B.log 2>&1
  f=$(find /tmp/prof_$B -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_f16mix_$B.csv >> $O/prof_$B.log 2>&1
done
head -50 $O/kernel_stats_timed_f16mix_f16.csv
