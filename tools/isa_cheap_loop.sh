#!/bin/bash
# development: compiles rt_march.hip to ISA and reports the register footprint of k_march<2,4,false,false,true> and the scalar-spill
# reloads (v_readlane) / instruction counts in the straight-line part of its cheap loop (from the loop's second block to the first
# block behind the refusal branch)
cd "$(dirname "$0")/../raytracing.jl_amd/csrc" || exit 1
OUT=${1:-/tmp/rt_march.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-fast-math -fPIC --cuda-device-only -S rt_march.hip -o "$OUT" 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
L = open(sys.argv[1]).read().split('\n')
name = '_ZN2rt7k_marchILi2ELi4ELb0ELb0ELb1E'
start = next(i for i, l in enumerate(L) if l.startswith(name) and l.rstrip().endswith(':') or (l.startswith(name) and ': ' in l and '@' in l))
end = next(i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end'))
F = L[start:end]
# the cheap loop: the depth-3 loop that holds the ds_add_f64 of the fused tally
k = next(i for i, l in enumerate(F) if 'ds_add_f64' in l)
hdr = None
for i in range(k, 0, -1):
    m = re.search(r'Header=(BB\d+_\d+) Depth=3', F[i])
    if m: hdr = m.group(1); break
a = next(i for i, l in enumerate(F) if l.startswith('.L' + hdr + ':'))
b = next(i for i in range(k, len(F)) if F[i].startswith('.LBB'))
body = [l.split(';')[0].strip() for l in F[a:b]]
body = [l for l in body if l and not l.startswith('.')]
cnt = lambda p: sum(1 for l in body if re.match(p, l))
print('cheap loop %s: %d instructions to the end of the tally block: valu %d (readlane %d, writelane %d), salu %d, vmem %d, lds %d, waitcnt %d'
      % (hdr, len(body), cnt(r'v_'), cnt(r'v_readlane'), cnt(r'v_writelane'), cnt(r's_(?!waitcnt)'), cnt(r'(global|flat|buffer)_'), cnt(r'ds_'), cnt(r's_waitcnt')))
for l in L[end:end + 400]:
    if re.search(r'\.(vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):', l) is None: continue
for i, l in enumerate(L):
    if l.strip().startswith('.name:') and name in l:
        blk = '\n'.join(L[max(0, i - 40):i + 30])
        for key in ('.vgpr_count', '.sgpr_count', '.sgpr_spill_count', '.vgpr_spill_count', '.private_segment_fixed_size'):
            m = re.search(re.escape(key) + r':\s*(\d+)', blk)
            if m: print(key, m.group(1), end='  ')
        print()
        break
PY
