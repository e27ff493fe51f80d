# usage: bash tools/gpu_grid.sh <tag> : the 1-GPU cost of capping the persistent GEMM grid (what data-parallel runs do to leave CUs to RCCL)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for g in 256 248; do
  v=$(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --gemm-grid $g 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "gemm_grid=$g rep$rep $v"
done; done > $O/grid.txt
cat $O/grid.txt
