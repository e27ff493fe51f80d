"""PMC counters for the GEMM instances the benchmarked step ACTUALLY times, with their real epilogues.

    run:    python tools/gemm_pmc.py run <order.json>            (under `rocprofv3 --kernel-trace --pmc <group> ...`, one pass per group)
    parse:  python tools/gemm_pmc.py parse <order.json> <out.txt> <counter_collection.csv> [more csv ...]

`run` enqueues every case of tools/gemm_bench.py (the step's shapes, operand formats, epilogues and output planes: two-product and
single-product forward, dgrad, TN wgrad) REPS times in a fixed order and writes that order; `parse` walks the gemm_big dispatches of
each counter CSV in dispatch order, assigns them to the cases, drops the first half of every case's dispatches (warm-up) and prints
one row per case: template instance, counters per launch, MFMA-busy fraction, fabric bytes (calibrated as profiles/r03_traffic_
calibration.json: FETCH_SIZE reports 2048 B per KiB on the LDS-DMA and 16-byte paths, WRITE_SIZE 1024 / 1019 B) next to SURVEY
8(d)'s bf16-algorithmic bytes of the shape (A + B + C in bf16) and the bytes of the formats the kernel really moves."""
import json
import os
import sys

REPS = 4


def case_list():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gemm_bench as gb
    out = []
    seen = set()
    for single, tag in ((0, "x2"), (15, "x1")):
        for name, m, n, k, p, run in gb.cases(2, int(os.environ.get("PMC_BWD", "4")), single):       # 4: the fp16 backward (round 6); 1: bf16
            key = (name, p)
            if key in seen or name.startswith("text"):
                continue
            seen.add(key)
            out.append((f"{name.strip()} [{'1 x fp16' if p == 4 else ('2 x fp16' if p == 2 else ('3 x bf16' if p == 3 else '1 x bf16'))}]", m, n, k, p, run))
    return out


def run(order_path):
    import torch
    cases = case_list()
    order = []
    for name, m, n, k, p, fn in cases:
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        order.append({"case": name, "M": m, "N": n, "K": k, "passes": p, "reps": REPS})
    with open(order_path, "w") as f:
        json.dump(order, f, indent=1)
    print("ran", len(order), "cases x", REPS)


def parse(order_path, out_path, csvs):
    import csv
    import collections
    order = json.load(open(order_path))
    table = collections.OrderedDict((o["case"], dict(o)) for o in order)
    for path in csvs:
        rows = [r for r in csv.DictReader(open(path)) if "gemm_big_kernel" in r.get("Kernel_Name", "")]
        by_disp = collections.OrderedDict()
        for r in rows:
            d = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])        # summed over counter instances
        disp = [by_disp[k] for k in sorted(by_disp)]
        want = sum(o["reps"] for o in order)
        if len(disp) != want:
            print(f"{path}: {len(disp)} gemm_big dispatches, expected {want}: skipped", file=sys.stderr)
            continue
        i = 0
        for o in order:
            grp = disp[i:i + o["reps"]][o["reps"] // 2:]
            i += o["reps"]
            ent = table[o["case"]]
            ent["instance"] = grp[0]["name"].split("gemm_big_kernel")[1].split("(")[0]
            for c in grp[0]:
                if c != "name":
                    ent[c] = sum(g[c] for g in grp) / len(grp)
    write_table(table, out_path)


def write_table(table, out_path):
    with open(out_path, "w") as f:
        def w(s=""):
            f.write(s + "\n")
            print(s)
        w("# tools/gemm_pmc.py: PMC counters per launch of the gemm_big instances the benchmarked step runs, real epilogues (M = 25 120 tokens);")
        w("# isolated launches on an idle GPU (the wgrads with the full k-slice count of a main-stream launch), mean of the last 2 of 4 launches per case.")
        w("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (the counter adds up")
        w("#   the SIMDs' busy cycles: a 16x16x32 MFMA keeps one SIMD busy for 16 cycles, which reproduces 2 M N K x products / 1024 flop per cycle);")
        w("# MFMA instr = SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16} / 512 (the counter counts 512-flop units; 16x16x32 = 16 384 flop = 32 units... per lane group)")
        w("# fabric bytes = FETCH_SIZE x 2048 B + WRITE_SIZE x 1024 B (calibration: profiles/r03_traffic_calibration.json; Infinity-Cache hits are counted);")
        w("# algo = bf16-algorithmic bytes of the shape (SURVEY 8d: A + B + C as bf16; wgrad: bf16 operands + fp32 C); us = kernel cycles / 2.4 GHz (profiler clock, ~5 % above the un-profiled launch)")
        w(f"{'case':30s} {'instance <MF,TN,EPI,PROD,MIXED>':30s} {'us':>6s} {'MFMA busy':>9s} {'LDS conflict/active':>20s} {'fetch MB':>9s} {'write MB':>9s} {'algo MB':>8s} {'ratio':>6s}")
        for case, e in table.items():
            cyc = e.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0
            busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (1024.0 * cyc)
            lds = "%.3g / %.3g" % (e.get("SQ_LDS_BANK_CONFLICT", float("nan")), e.get("SQ_LDS_IDX_ACTIVE", float("nan")))
            fe = e.get("FETCH_SIZE", float("nan")) * 2048 / 1e6
            wr = e.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
            m, n, k = e["M"], e["N"], e["K"]
            tn = "wgrad" in case
            algo = (2 * (k * m + k * n) + 4 * m * n if tn else 2 * (m * k + n * k + m * n)) / 1e6
            w(f"{case:30s} {e.get('instance', '?'):30s} {cyc / 2400.0:6.1f} {busy:9.3f} {lds:>20s} {fe:9.1f} {wr:9.1f} {algo:8.1f} {(fe + wr) / algo:6.2f}")
    json.dump(table, open(out_path.replace(".txt", ".json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "table":          # re-print a saved summary: python tools/gemm_pmc.py table <summary.json> <out.txt>
        write_table(json.load(open(sys.argv[2])), sys.argv[3])
    else:
        parse(sys.argv[2], sys.argv[3], sys.argv[4:])
