cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -400 ) > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) > $O/smoke.log 2>&1
cat $O/smoke.log | cut -c1-300
timeout 900 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err
wc -l $O/bench_noflags.json; python -c "
import json; d=json.loads(open('$O/bench_noflags.json').readline()); print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], 'dp', d['dp_policy_at_world_size_1'].get('value', d['dp_policy_at_world_size_1']), 'roof', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); print(d['value'], d['ms_per_step'], 'dp', d['dp_policy_at_world_size_1'].get('value'), 'roof', d['roofline']['frac'], 'h2d', d['with_h2d_uint8']['value'], 'fast', d['fast_mode_bf16']['value'], 'grad', d['grad_rel_err']['max'])"
