#!/bin/bash
# Register / scratch / LDS use of every kernel in the product library, from the code-object metadata (no GPU needed).
# usage: tools/kernel_regs.sh [lib.so] [name filter]
LIB=${1:-egovlp_amd/libegovlp_hip.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB >/dev/null 2>&1
python3 - "$LIB" "$T" "${2:-}" <<'PY'
import subprocess, sys, re, os
lib, tmp, filt = sys.argv[1], sys.argv[2], sys.argv[3]
data = open(lib, 'rb').read()
# the device code object is an ELF embedded in the fat binary section: carve every ELF with e_machine = AMDGPU (224)
outs = []
i = 0
while True:
    i = data.find(b'\x7fELF', i)
    if i < 0: break
    if data[i + 18] == 224:
        outs.append(i)
    i += 4
for n, off in enumerate(outs):
    f = os.path.join(tmp, f'co{n}.elf')
    open(f, 'wb').write(data[off:])
    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f], capture_output=True, text=True).stdout
    for blk in txt.split('- .agpr_count:')[1:]:
        g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
        name = g('name')
        if filt and filt not in name: continue
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        print(f"vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {dem[:120]}")
PY
rm -rf $T
