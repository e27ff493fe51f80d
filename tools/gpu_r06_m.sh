cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
SECONDS=0; timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench wall seconds: $SECONDS"; tail -5 $O/bench_default.err | grep -i "error\|Traceback"; python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r.get('traffic'), r.get('algorithmic_bytes_per_launch'), r.get('traffic_over_algorithmic'), r.get('traffic_missing_because'), d.get('grad_rel_err',{}).get('max'), d.get('precision_guard',{}).get('tried'))"
