cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for sk in 0 256 4352 1280 65792; do echo "EGV_PLANE_SKEW=$sk"; EGV_PLANE_SKEW=$sk timeout 200 python tools/attn_time.py 2>&1 | grep "passes=3"; EGV_PLANE_SKEW=$sk timeout 200 python tools/gemm_bench.py 3 1 2>&1 | grep -E "qkv   fwd|fc1   fwd|TOTAL" | cut -c1-100; done > $O/skew.txt
cat $O/skew.txt
