"""HBM traffic of the dominant kernel from rocprofv3 --pmc passes of bench.py (one pass per counter: FETCH_SIZE and WRITE_SIZE do
not fit one pass; never combined with sys / hip traces).  FETCH_SIZE / WRITE_SIZE are reported in KiB.  Scale factors are the
CALIBRATED ones (tools/traffic_calib.py -> profiles/r03_traffic_calibration.json: copies of a known 1 GiB on the GEMM's own paths):
a reported FETCH KiB stands for 2048 bytes with LDS-DMA dwordx4 loads and with plain 16-byte loads alike (the gfx950 note of
MI355X_MICROARCH.md, confirmed), a reported WRITE KiB for 1024 bytes (write-back) / 1019 (nt); reads served by the Infinity
Cache ARE counted (a 96 MB buffer read back right after it was written: 100 %), so the number is fabric traffic, an upper bound
of DRAM traffic.
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
                 "over all gemm_big_kernel launches; bytes = 2048 x FETCH_SIZE[KiB] + 1024 x WRITE_SIZE[KiB], the factors "
                 "calibrated on 1 GiB copies through the GEMM's own load / store paths (profiles/r03_traffic_calibration.json)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
