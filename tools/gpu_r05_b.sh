# round 5, GPU call B: time-attention row stores (A/B against the direct 8-byte stores), per-instance PMC counters, the full GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" 2>&1 | grep -v "amdgpu\|^$" | tail -5 ) > $O/pytest_attn.txt 2>&1
tail -2 $O/pytest_attn.txt
for L in main tmfold; do
  F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
  for rep in 1 2; do
    echo "lib=$L rep=$rep" >> $O/attn_time_ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 200 python tools/attn_time.py 2>&1 | grep "time attention" ) >> $O/attn_time_ab.txt
    ( ATT_T=16 ATT_B=16 EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 200 python tools/attn_time.py 2>&1 | grep "time attention" | sed 's/^/T16 /' ) >> $O/attn_time_ab.txt
  done
done
cat $O/attn_time_ab.txt
bash tools/gpu_ab_lib.sh $1 "" main tmfold > $O/ab_lib.log 2>&1
cat $O/ab.txt
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  n=$(echo $G | cut -d' ' -f1)
  rm -rf /tmp/gp_$n
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/gp_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py run $O/gemm_pmc_order.json ) > $O/gemm_pmc_$n.log 2>&1
  f=$(find /tmp/gp_$n -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/gemm_pmc_$n.csv
done
python tools/gemm_pmc.py parse $O/gemm_pmc_order.json $O/gemm_pmc_summary.txt $O/gemm_pmc_*.csv > $O/gemm_pmc_parse.log 2>&1
cat $O/gemm_pmc_summary.txt | cut -c1-220
head -3 $O/gemm_pmc_FETCH_SIZE.csv > $O/gemm_pmc_csv_head.txt 2>/dev/null
rm -f $O/gemm_pmc_*.csv
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -400 ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
