# usage: bash tools/gpu_dist_prof.sh <tag> : rocprofv3 kernel trace of the world-size-1 RCCL run (Bf16GradSync) reduced to the timed steps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
rm -rf /tmp/prof_dist
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dist -o p -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --force-dist --gemm-grid 248 ) > $O/prof.log 2>&1
for f in $(find /tmp/prof_dist -name "*kernel_trace.csv"); do echo $f $(wc -l < $f); done >> $O/prof.log
f=$(find /tmp/prof_dist -name "*kernel_trace.csv" -size +100k | head -1)
[ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_sync.csv >> $O/prof.log 2>&1
head -3 $O/kernel_stats_timed_sync.csv | cut -c1-200; tail -5 $O/prof.log | cut -c1-300
