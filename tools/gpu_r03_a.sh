# round 3, call A: validate the ExecContext refactor + new tests, baseline bench, two env A/Bs, counter calibration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) > $O/smoke.log 2>&1
cat $O/smoke.log
( timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "amdgpu\|^$" | tail -300 ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-600 $O/bench_default.json; tail -3 $O/bench_default.err
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory"
for rep in 1 2; do
  for v in "EGV_TEXT_LAST=1" "EGV_TEXT_LAST=0" "EGV_LN_DY_PLANES=1"; do
    echo "== $v rep $rep" >> $O/ab.txt
    ( env $v timeout 300 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" ) >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/cal_$C -o p -- python $GRAFT_REPO_ROOT/tools/traffic_calib.py run ) > $O/calib_$C.log 2>&1
done
python tools/traffic_calib.py parse $(find /tmp/cal_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/cal_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/traffic_calibration.json > $O/calib_parse.log 2>&1
tail -30 $O/calib_parse.log
