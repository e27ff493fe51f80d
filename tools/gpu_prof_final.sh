# usage: bash tools/gpu_prof_final.sh <tag> : rocprofv3 kernel trace of the default bench command, reduced to the timed steps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
rm -rf /tmp/prof_fin
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fin -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode ) > $O/prof.log 2>&1
f=$(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_mixed.csv >> $O/prof.log 2>&1
head -2 $O/kernel_stats_timed_mixed.csv | cut -c1-200; grep -o '"value": [0-9.]*' $O/prof.log | head -1
