#!/usr/bin/env python3
"""Per-kernel fingerprint of a hipcc --save-temps .s file: was an instance's machine code changed by an edit of the source?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I egovlp_amd/csrc --save-temps -c egovlp_amd/csrc/gemm_big.hip -o /tmp/x.o
    python tools/isa_hash.py gemm_big-hip-amdgcn-amd-amdhsa-gfx950.s [other.s]

One line per kernel: demangled-ish template arguments, instruction count, VGPRs / AGPRs / spills / LDS as the assembler reports them, and a
hash of the instruction stream with the kernel's own (mangled) name removed -- so that adding a template parameter, which renames every
instance, does not by itself change the fingerprint.  With two files: the kernels whose fingerprints differ (matched by the leading template
arguments the two names share)."""
import hashlib
import re
import sys


def kernels(path):
    txt = open(path).read().split("\n")
    out = {}
    i = 0
    while i < len(txt):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", txt[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        body = []
        j = i + 1
        while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
            body.append(txt[j])
            j += 1
        meta = {}
        k = j
        while k < len(txt) and k < j + 80:
            mm = re.match(r"^;\s*(NumVgprs|NumAgprs|ScratchSize|LDSByteSize|Occupancy|NumSgprs|TotalNumVgprs):\s*(\d+)", txt[k])
            if mm:
                meta[mm.group(1)] = int(mm.group(2))
            if re.match(r"^(_Z\w+):", txt[k]):
                break
            k += 1
        ins = []
        for ln in body:
            s = ln.split(";")[0].strip()
            if not s or s.startswith("."):
                if s.startswith(".LBB"):
                    ins.append(re.sub(r"\.LBB\d+_", ".LBB_", s))
                continue
            s = s.replace(name, "SELF")
            s = re.sub(r"\.LBB\d+_", ".LBB_", s)
            ins.append(s)
        h = hashlib.sha1("\n".join(ins).encode()).hexdigest()[:12]
        if "gemm_big_kernel" in name or "kernel" in name:
            out[name] = (len(ins), meta, h)
        i = j
    return out


def targs(name):
    m = re.search(r"kernelI(.*?)EEv", name)
    if not m:
        return name
    return re.sub(r"ELb|ELi", ",", m.group(1)).replace("Lb", "").replace("Li", "").rstrip("E")


def main():
    a = kernels(sys.argv[1])
    for n, (cnt, meta, h) in sorted(a.items()):
        print(f"{targs(n):28s} ins {cnt:6d}  vgpr {meta.get('NumVgprs', -1):3d} agpr {meta.get('NumAgprs', -1):3d} scratch {meta.get('ScratchSize', -1):4d}"
              f" lds {meta.get('LDSByteSize', -1):6d}  {h}")
    if len(sys.argv) > 2:
        b = kernels(sys.argv[2])
        ta = {targs(n): v for n, v in a.items()}
        tb = {targs(n): v for n, v in b.items()}
        print("---- differences (first file's template arguments as prefix of the second's)")
        for k, v in sorted(ta.items()):
            match = [kb for kb in tb if kb == k or kb.startswith(k + ",")]
            same = [kb for kb in match if tb[kb][2] == v[2]]
            print(f"{k:28s} {'same as ' + same[0] if same else 'CHANGED (candidates: ' + ', '.join(match) + ')'}")


if __name__ == "__main__":
    main()
