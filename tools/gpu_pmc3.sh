# L2 behaviour of gemm_big on the two forward shapes: bash tools/gpu_pmc3.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; ctrs=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o p -- "$@" ) > $O/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc_$name.csv
}
for shape in "25120 768 3072" "25120 2304 768"; do
  s=$(echo $shape | tr ' ' 'x')
  run l2a_$s "TCC_HIT_sum TCC_MISS_sum" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
  run l2b_$s "TCC_REQ_sum TCC_READ_sum" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
  run l2c_$s "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
done
python - <<PY
import csv, glob, os, collections
out = open("$O/pmc_l2_summary.txt", "w")
for f in sorted(glob.glob("$O/pmc_l2*.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        if "gemm_big" not in r.get("Kernel_Name", ""): continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[2:] if len(v) > 3 else v
        out.write("%s %s mean_per_launch %.6g n=%d\n" % (os.path.basename(f), c, sum(v) / len(v), len(v)))
out.close()
print(open("$O/pmc_l2_summary.txt").read())
PY
tail -2 $O/pmc_l2a_25120x768x3072.log
