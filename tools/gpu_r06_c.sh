cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for W in space time; do
( EGV_EXP_ATT3=$W timeout 600 python tools/f16bwd_check.py 4 2>&1 | grep -v "amdgpu\|Warning\|detach\|return te\|worst" | grep -A1 "per-kernel  " ) > $O/f16bwd_check_att3_$W.txt 2>&1
echo $W; cat $O/f16bwd_check_att3_$W.txt
done
