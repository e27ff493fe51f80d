# kernel-level evidence for BASELINE configs 4 (T = 16, B = 16) and 5 (ViT-L/14, B = 16): bash tools/gpu_cfg45_profile.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for C in "config4 --frames 16 --batch 16" "config5 --arch large_patch14_224 --batch 16"; do
  set -- $C; name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --wgrad-side 0 --text-side 0 "$@" ) > $O/prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/${name}_kernel_stats_timed_mixed.csv >> $O/prof_$name.log 2>&1
  head -25 $O/${name}_kernel_stats_timed_mixed.csv
done
