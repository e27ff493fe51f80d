# PMC passes over the attention kernels (tools/attn_time.py runs every mode / precision a few times)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
timeout 200 python tools/attn_time.py > $O/attn_time.txt 2>&1
run() { name=$1; shift; ctrs=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/tools/attn_time.py ) > $O/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$name.csv; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run sq2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
python - <<PY
import csv, glob, os, collections
out = open("$O/pmc_attn_summary.txt", "w")
for f in sorted(glob.glob("$O/pmc_*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "attn" not in k: continue
        k = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            out.write("%-40s %-32s mean %.5g n=%d\n" % (k, c, sum(v) / len(v), len(v)))
out.close()
print(open("$O/pmc_attn_summary.txt").read())
PY
cat $O/attn_time.txt
