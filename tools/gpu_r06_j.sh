cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for c in 30 10 5; do echo "HEAVY_COMP=$c"; EGV_HEAVY_COMP=$c timeout 600 python tools/heavy_check.py heavy 2>&1 | grep -v amdgpu; done | tee $O/heavy_check.txt
