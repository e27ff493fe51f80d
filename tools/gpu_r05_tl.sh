cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
rm -rf /tmp/prof_tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg ) > $O/prof_tl.log 2>&1
python tools/timeline.py $(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1) ${2:-27} ${3:-40} > $O/stream_timeline_tail.txt 2>&1
head -5 $O/stream_timeline_tail.txt; grep -c "" $O/stream_timeline_tail.txt
