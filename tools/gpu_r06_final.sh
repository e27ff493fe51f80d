# round 6 evidence on one MI355X: bash tools/gpu_r06_final.sh <tag>  -> gpurun_out/<tag>/   (what is to be judged is copied into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -420 ) > $O/pytest_gpu.txt 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -8 ) > $O/smoke.log 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_on_1gpu_box.txt 2>&1; echo "exit code $?" >> $O/bench_gpus2_on_1gpu_box.txt
LEGS="--no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err --no-guard --no-traffic"
# kernel statistics of the timed steps, one stream: the benchmarked mode, round 5's pairing (bf16 backward), plain bf16
for PREC in "f16mix f16" "f16mix bf16" "bf16 bf16"; do
  set -- $PREC; F=$1; Bk=$2
  rm -rf /tmp/prof_$F$Bk
  ( cd /tmp && EGV_X2_BWD=$Bk timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$F$Bk -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 $LEGS --wgrad-side 0 --text-side 0 --precision $F ) > $O/prof_${F}_$Bk.log 2>&1
  f=$(find /tmp/prof_$F$Bk -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_${F}_$Bk.csv >> $O/prof_${F}_$Bk.log 2>&1
done
# the default three-stream configuration: stream-level timeline of one step
rm -rf /tmp/prof_tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 $LEGS ) > $O/prof_tl.log 2>&1
python tools/timeline.py $(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1) > $O/stream_timeline.txt 2>&1
f=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_stats.py $f 3 $O/kernel_stats_timed_f16mix_f16_three_streams.csv >> $O/prof_tl.log 2>&1
# per-instance counters of the GEMMs the step runs (fp16 backward): one rocprofv3 pass per counter group over the same launches
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  n=$(echo $G | cut -d' ' -f1)
  rm -rf /tmp/gp_$n
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/gp_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py run $O/gemm_pmc_order.json ) > $O/gemm_pmc_$n.log 2>&1
  f=$(find /tmp/gp_$n -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/gemm_pmc_$n.csv
done
python tools/gemm_pmc.py parse $O/gemm_pmc_order.json $O/gemm_pmc_per_instance.txt $O/gemm_pmc_*.csv > $O/gemm_pmc_parse.log 2>&1
rm -f $O/gemm_pmc_*.csv
timeout 300 python tools/gemm_bench.py 2 4 2>&1 | grep -v amdgpu > $O/gemm_bench_f16x2_f16bwd.txt
BENCH_SINGLE=15 timeout 300 python tools/gemm_bench.py 2 4 2>&1 | grep -v amdgpu > $O/gemm_bench_f16_single_f16bwd.txt
timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu > $O/attention_isolated.txt
timeout 300 python tools/host_overhead.py f16mix 2>&1 | grep -v amdgpu > $O/host_overhead.txt
timeout 900 python bench.py --steps 20 --warmup 5 --frames 16 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config4_T16_B16.json 2>> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config5_vitl14_B16.json 2>> $O/bench_default.err
# same-box interleaved A/B of the backward pairing at the headline size
TAG=$(basename $O); bash tools/gpu_ab_env2.sh $TAG/ab_bwd 2 "EGV_X2_BWD=bf16" "EGV_X2_BWD=f16" > /dev/null 2>&1; cp $O/ab_bwd/ab.txt $O/ab_backward_pairing.txt
# the data-parallel code path (process group, RCCL streams, gradient exchange, 248-workgroup grid) at world size 1
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 $LEGS --force-dist 2>&1 | grep "^{" ) > $O/bench_force_dist_w1.json
tail -4 $O/pytest_gpu.txt; cat $O/smoke.log | tail -3; cut -c1-300 $O/bench_default.json; tail -3 $O/bench_gpus2_on_1gpu_box.txt; head -3 $O/kernel_stats_timed_f16mix_f16.csv; head -6 $O/stream_timeline.txt; cat $O/ab_backward_pairing.txt
echo done
