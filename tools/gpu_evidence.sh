# round evidence on one MI355X: bash tools/gpu_evidence.sh <tag>  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -200 ) > $O/pytest_gpu.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --text-dropout 0 --no-cpu-baseline > $O/bench_nodropout.json 2>> $O/bench_default.err
for PREC in mixed bf16; do
  rm -rf /tmp/prof_$PREC
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$PREC -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --precision $PREC ) > $O/prof_$PREC.log 2>&1
  f=$(find /tmp/prof_$PREC -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_$PREC.csv >> $O/prof_$PREC.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fast-mode ) > $O/pmc_$C.log 2>&1
done
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/gemm_traffic.json > $O/pmc_traffic.log 2>&1
bash tools/gpu_pmc.sh $1 > $O/pmc.log 2>&1
timeout 300 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_bench_mixed.txt
timeout 300 python tools/gemm_bench.py 1 1 2>&1 | grep -v amdgpu > $O/gemm_bench_bf16.txt
timeout 100 python tools/gemm_trace.py 25120 2304 768 2>&1 | grep -v amdgpu > $O/tile_timeline_qkv_fwd.txt
timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -v amdgpu > $O/tile_timeline_fc2_fwd.txt
tail -5 $O/pytest_gpu.txt; cat $O/smoke.log; cut -c1-400 $O/bench_default.json; cat $O/pmc_traffic.log; head -3 $O/kernel_stats_timed_mixed.csv
