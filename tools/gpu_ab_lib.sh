# same-box A/B of library variants: bash tools/gpu_ab_lib.sh <tag> "<pytest -k expr or empty>" <suffix> [<suffix> ...]
# (suffix "" = the product library egovlp_amd/libegovlp_hip.so; others: egovlp_amd/libegovlp_hip_<suffix>.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; K="$2"; shift; shift
if [ -n "$K" ]; then
  ( timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | grep -v "amdgpu\|^$" | tail -40 ) > $O/pytest_subset.txt 2>&1
  tail -3 $O/pytest_subset.txt
fi
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2 3; do
  for sfx in "$@"; do
    L=egovlp_amd/libegovlp_hip.so; [ "$sfx" != "main" ] && L=egovlp_amd/libegovlp_hip_$sfx.so
    echo -n "lib=$sfx rep=$rep " >> $O/ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" ) >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
rm -rf /tmp/prof_ab
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 ) > $O/prof.log 2>&1
f=$(find /tmp/prof_ab -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_mixed.csv >> $O/prof.log 2>&1
head -30 $O/kernel_stats_timed_mixed.csv | cut -c1-150
