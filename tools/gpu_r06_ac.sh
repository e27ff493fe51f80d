cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06ac; mkdir -p $O
LEGS="--no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err --no-guard --no-traffic"
rm -rf /tmp/prof_tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 $LEGS ) > $O/prof_tl.log 2>&1
f=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 0 40 > $O/timeline_full.txt 2>&1
head -6 $O/timeline_full.txt
