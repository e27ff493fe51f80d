# world-size-1 RCCL A/B of bench.py environments: bash tools/gpu_forcedist_ab.sh "<ENV...>" "<ENV...>" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in "$@"; do
  echo -n "force-dist [$e] rep=$rep: "
  env $e timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$rep bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --force-dist 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['comm']['grad_sync'], d['comm']['grad_sync_exposed_ms_mean'])"
done; done
