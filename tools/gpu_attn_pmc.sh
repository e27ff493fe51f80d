# PMC counters of the attention kernels (isolated launches of tools/attn_time.py); one rocprofv3 pass per counter group
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; ctrs=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o p -- "$@" ) > $O/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc_$name.csv
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" python $GRAFT_REPO_ROOT/tools/attn_time.py
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" python $GRAFT_REPO_ROOT/tools/attn_time.py
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" python $GRAFT_REPO_ROOT/tools/attn_time.py
run d "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" python $GRAFT_REPO_ROOT/tools/attn_time.py
run e "GRBM_GUI_ACTIVE FETCH_SIZE" python $GRAFT_REPO_ROOT/tools/attn_time.py
run f "WRITE_SIZE" python $GRAFT_REPO_ROOT/tools/attn_time.py
python - <<PY
import csv, glob, os, collections
out = open("$O/pmc_summary.txt", "w")
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("$O/pmc_*.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        k = r.get("Kernel_Name", "")
        if "attn" not in k: continue
        k = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        tab[k][c] = (sum(v) / len(v), len(v))
for k in sorted(tab):
    out.write(k + "\n")
    for c, (m, n) in sorted(tab[k].items()):
        out.write("    %-28s %14.6g  (n=%d)\n" % (c, m, n))
out.close()
print(open("$O/pmc_summary.txt").read())
PY
