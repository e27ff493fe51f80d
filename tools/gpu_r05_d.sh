# round 5, GPU call D: TN (wgrad) work-id -> XCD mapping, A/B against the per-slice remap (libegovlp_hip_tnold.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16x2.py -q -x -k "gemm or wgrad or tn" 2>&1 | grep -v "amdgpu\|^$" | tail -4 ) > $O/pytest_gemm.txt 2>&1
tail -2 $O/pytest_gemm.txt
for L in main tnold; do
  F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
  for rep in 1 2; do
    echo "lib=$L rep=$rep" >> $O/gemm_bench_ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 300 python tools/gemm_bench.py 1 1 2>&1 | grep "wgrad\|TOTAL" ) >> $O/gemm_bench_ab.txt
  done
done
cat $O/gemm_bench_ab.txt
bash tools/gpu_ab_lib.sh $1 "" main tnold > $O/ab_lib.log 2>&1
cat $O/ab.txt
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $G | cut -d' ' -f1)
  rm -rf /tmp/gp_$n
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/gp_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py run $O/gemm_pmc_order.json ) > $O/gemm_pmc_$n.log 2>&1
  f=$(find /tmp/gp_$n -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/gemm_pmc_$n.csv
done
python tools/gemm_pmc.py parse $O/gemm_pmc_order.json $O/gemm_pmc_summary.txt $O/gemm_pmc_*.csv > $O/gemm_pmc_parse.log 2>&1
grep "wgrad" $O/gemm_pmc_summary.txt | cut -c1-220
rm -f $O/gemm_pmc_*.csv
