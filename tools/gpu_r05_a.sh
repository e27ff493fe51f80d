# round 5, GPU call A: the single-fp16-product forward ('f16mix') -- new kernel tests, shape timings, same-box A/B of the precision
# policies, per-instance PMC counters of the GEMMs the step runs, one default bench line.   bash tools/gpu_r05_a.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_block.py -q -s -x 2>&1 | grep -v "amdgpu\|^$" | tail -60 ) > $O/pytest_new.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_model.py -q -s -x -k "golden_in_the_f16x2_mode" 2>&1 | grep -v "amdgpu\|^$" | tail -20 ) > $O/pytest_golden.txt 2>&1
tail -3 $O/pytest_new.txt; tail -4 $O/pytest_golden.txt
( timeout 300 python tools/gemm_bench.py 2 1 2>&1 | grep -v amdgpu ) > $O/gemm_bench_f16x2.txt
( BENCH_SINGLE=7 timeout 300 python tools/gemm_bench.py 2 1 2>&1 | grep -v amdgpu ) > $O/gemm_bench_f16_single.txt
cat $O/gemm_bench_f16_single.txt | head -8
bash tools/gpu_ab_env.sh $1 2 "EGOVLP_PRECISION=f16x2" "EGOVLP_PRECISION=f16mix" "EGOVLP_PRECISION=f16mix EGV_F16_SINGLE=fc2:0,fc1:0,qkv:0" "EGOVLP_PRECISION=f16mix EGV_F16_SINGLE=fc2:3,fc1:3" > /dev/null 2>&1
cat $O/ab.txt
# per-instance counters: one rocprofv3 pass per counter group over the same sequence of launches
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  n=$(echo $G | cut -d' ' -f1)
  rm -rf /tmp/gp_$n
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/gp_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py run $O/gemm_pmc_order.json ) > $O/gemm_pmc_$n.log 2>&1
  f=$(find /tmp/gp_$n -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/gemm_pmc_$n.csv
done
python tools/gemm_pmc.py parse $O/gemm_pmc_order.json $O/gemm_pmc_summary.txt $O/gemm_pmc_*.csv > $O/gemm_pmc_parse.log 2>&1
cat $O/gemm_pmc_summary.txt | cut -c1-200
rm -f $O/gemm_pmc_*.csv
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
