"""HBM traffic of the dominant kernel from rocprofv3 --pmc passes of bench.py (one pass per counter: FETCH_SIZE and WRITE_SIZE do
not fit one pass; never combined with sys / hip traces).  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts a wide coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM): it is doubled here.  WRITE_SIZE is used as reported.
usage: python tools/pmc_traffic.py <fetch counter csv> <write counter csv> <out.json>"""
import csv
import json
import sys


def per_launch(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and "gemm_big_kernel" in r.get("Kernel_Name", ""):
            tot += float(r["Counter_Value"])
            n += 1
    return tot / max(n, 1), n


f, nf = per_launch(sys.argv[1], "FETCH_SIZE")
w, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0), "fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1),
       "launches": [nf, nw],
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1`, mean "
                 "over all gemm_big_kernel launches; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
