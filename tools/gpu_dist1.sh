# usage: bash tools/gpu_dist1.sh <tag> : cost of the data-parallel machinery at world size 1 under RCCL (hooks, pack / unpack, self all-reduce)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
C="bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing"
P=29630
show() { echo $1 $(grep -o '"value": [0-9.]*' $O/$1.json | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/$1.json) $(grep -o '"ms_per_step": [0-9.]*' $O/$1.json | head -1); }
run() { name=$1; shift; P=$((P+1)); timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P $C --force-dist "$@" > $O/$name.json 2>> $O/err; show $name; }
for rep in 1 2; do
timeout 300 python $C --wgrad-side 0 > $O/plain$rep.json 2> $O/err; show plain$rep
timeout 300 python $C --wgrad-side 1 > $O/wgrad_side$rep.json 2>> $O/err; show wgrad_side$rep
done
run torchrun_sync_poll_w0 --gemm-grid 248 --wgrad-side 0
run torchrun_sync_poll_w1 --gemm-grid 248 --wgrad-side 1
