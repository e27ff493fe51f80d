# round evidence on one MI355X: bash tools/gpu_evidence.sh <tag>  -> gpurun_out/<tag>/   (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -300 ) > $O/pytest_gpu.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_on_1gpu_box.txt 2>&1; echo "exit code $?" >> $O/bench_gpus2_on_1gpu_box.txt
for PREC in f16mix f16x2 bf16; do
  rm -rf /tmp/prof_$PREC
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$PREC -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 --text-side 0 --precision $PREC ) > $O/prof_$PREC.log 2>&1
  f=$(find /tmp/prof_$PREC -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_$PREC.csv >> $O/prof_$PREC.log 2>&1
done
# the default three-stream configuration: stream-level timeline of one step
rm -rf /tmp/prof_tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg ) > $O/prof_tl.log 2>&1
python tools/timeline.py $(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1) > $O/stream_timeline.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C /tmp/cal_$C
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg ) > $O/pmc_$C.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/cal_$C -o p -- python $GRAFT_REPO_ROOT/tools/traffic_calib.py run ) > $O/calib_$C.log 2>&1
done
python tools/traffic_calib.py parse $(find /tmp/cal_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/cal_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/traffic_calibration.json > $O/calib_parse.log 2>&1
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/gemm_traffic.json > $O/pmc_traffic.log 2>&1
# per-instance counters of the GEMMs the step runs: one rocprofv3 pass per counter group over the same sequence of launches
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  n=$(echo $G | cut -d' ' -f1)
  rm -rf /tmp/gp_$n
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/gp_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py run $O/gemm_pmc_order.json ) > $O/gemm_pmc_$n.log 2>&1
  f=$(find /tmp/gp_$n -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/gemm_pmc_$n.csv
done
python tools/gemm_pmc.py parse $O/gemm_pmc_order.json $O/gemm_pmc_per_instance.txt $O/gemm_pmc_*.csv > $O/gemm_pmc_parse.log 2>&1
rm -f $O/gemm_pmc_*.csv
timeout 300 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_bench_mixed.txt; timeout 300 python tools/gemm_bench.py 2 1 2>&1 | grep -v amdgpu > $O/gemm_bench_f16x2.txt
BENCH_SINGLE=7 timeout 300 python tools/gemm_bench.py 2 1 2>&1 | grep -v amdgpu > $O/gemm_bench_f16_single.txt
timeout 300 python tools/hipblaslt_probe.py 2>&1 | grep -v amdgpu > $O/vendor_gemm_probe.txt
timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu > $O/attention_isolated.txt
timeout 300 python tools/host_overhead.py f16mix 2>&1 | grep -v amdgpu > $O/host_overhead.txt
timeout 600 python bench.py --steps 20 --warmup 5 --frames 16 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config4_T16_B16.json 2>> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config5_vitl14_B16.json 2>> $O/bench_default.err
# kernel statistics of BASELINE configs 4 (T = 16, B = 16) and 5 (ViT-L/14, B = 16)
for C in "config4 --frames 16 --batch 16" "config5 --arch large_patch14_224 --batch 16"; do
  set -- $C; name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 --text-side 0 "$@" ) > $O/prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/${name}_kernel_stats_timed_f16mix.csv >> $O/prof_$name.log 2>&1
done
# the data-parallel code path (process group, RCCL streams, gradient exchange, 248-workgroup grid) at world size 1
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --force-dist 2>&1 | grep "^{" ) > $O/bench_force_dist_w1.json
timeout 300 python tools/block_diag.py 2>&1 | grep -v amdgpu | cut -c1-400 > $O/block_calls_run_to_run.txt
tail -4 $O/pytest_gpu.txt; cat $O/smoke.log; cut -c1-300 $O/bench_default.json; tail -3 $O/bench_gpus2_on_1gpu_box.txt; cat $O/pmc_traffic.log; head -3 $O/kernel_stats_timed_f16mix.csv; head -6 $O/stream_timeline.txt
