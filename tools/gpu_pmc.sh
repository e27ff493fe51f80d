# PMC counters for the dominant kernel (gemm_big) on two hot-path shapes; one rocprofv3 pass per counter group
# (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with sys/hip traces).
# usage: bash tools/gpu_pmc.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
run() { # name, counters..., then -- cmd
  name=$1; shift; ctrs=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o p -- "$@" ) > $O/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc_$name.csv
}
for shape in "25120 768 3072" "25120 2304 768"; do
  s=$(echo $shape | tr ' ' 'x')
  run fetch_$s "FETCH_SIZE" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
  run write_$s "WRITE_SIZE" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
  run mfma_$s "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
  run lds_$s "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $shape 6
done
python - <<PY
import csv, glob, os, collections
out = open("$O/pmc_summary.txt", "w")
for f in sorted(glob.glob("$O/pmc_*.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        k = r.get("Kernel_Name", "")
        if "gemm_big" not in k: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[2:] if len(v) > 3 else v     # skip warm-up dispatches
        out.write("%s %s mean_per_launch %.6g n=%d\n" % (os.path.basename(f), c, sum(v) / len(v), len(v)))
out.close()
print(open("$O/pmc_summary.txt").read())
PY
