# wave-state PMC counters of gemm_big on fc2-forward: where do the wave-cycles go?  usage: bash tools/gpu_pmc2.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( cd /tmp && timeout 120 rocprofv3 -L 2>&1 | grep -oE "\b(SQ|SQC|TCP|TA|TD)_[A-Z0-9_]+" | sort -u > $O/counters_avail.txt )
run() { name=$1; shift; ctrs=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o p -- "$@" ) > $O/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc_$name.csv
}
S="25120 768 3072"
run wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $S 6
run act1 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $S 6
run act2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $S 6
run ifetch "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $S 6
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS" python $GRAFT_REPO_ROOT/tools/gemm_one.py 1 $S 6
python - <<PY
import csv, glob, os, collections
out = open("$O/pmc_summary.txt", "w")
for f in sorted(glob.glob("$O/pmc_*.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        if "gemm_big" not in r.get("Kernel_Name", ""): continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[2:] if len(v) > 3 else v
        out.write("%s %s mean_per_launch %.6g n=%d\n" % (os.path.basename(f), c, sum(v) / len(v), len(v)))
out.close()
print(open("$O/pmc_summary.txt").read())
PY
grep -E "WAIT|ACTIVE|BARRIER|IFETCH|INST_CYCLES|LEVEL" $O/counters_avail.txt | tr '\n' ' '
tail -3 $O/pmc_wait.log
