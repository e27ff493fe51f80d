# usage: bash tools/gpu_prof.sh <tag> <precision>   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; PREC=${2:-bf16}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --precision $PREC > $O/bench_$PREC.json 2> $O/bench_$PREC.err
cd /tmp && rm -rf /tmp/prof_$PREC && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$PREC -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fast-mode --precision $PREC > $O/prof_$PREC.log 2>&1
find /tmp/prof_$PREC -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_$PREC.csv \;
ls -R /tmp/prof_$PREC > $O/prof_ls.txt 2>&1
