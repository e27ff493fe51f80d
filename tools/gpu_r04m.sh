cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O
rm -f $O/attn_time_v4_dbg.txt
for l in "" _tmf_dbg1 _tmf_dbg2 _tmf_dbg3; do
  ( EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so ATT_B=16 ATT_T=16 timeout 300 python tools/attn_time.py 2>&1 | grep "time attention fwd" | sed "s/^/T16 B16 lib$l: /" ) >> $O/attn_time_v4_dbg.txt 2>&1
done
cat $O/attn_time_v4_dbg.txt
