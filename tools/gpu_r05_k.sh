cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
rm -rf /tmp/prof_k
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 --text-side 0 ) > $O/prof.log 2>&1
f=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_f16mix.csv >> $O/prof.log 2>&1
head -60 $O/kernel_stats_timed_f16mix.csv | cut -c1-190
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg 2>/dev/null | cut -c1-300
