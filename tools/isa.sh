#!/bin/bash
# tools/isa.sh -- device ISA (gfx950) and the per-kernel resource table of libmcl_hip.so's kernels.
#   tools/isa.sh [outdir]   ->  outdir/mcl.s (assembly), outdir/resources.tsv (kernel, sgpr, vgpr, scratch, lds, occupancy)
# Compile only (no GPU needed): the numbers quoted in DESIGN.md / profiles/ come from here.
set -e
OUT=${1:-/tmp/isa}
mkdir -p "$OUT"
cd "$(dirname "$0")/../smarc_navigation_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -fno-slp-vectorize -S -o "$OUT/mcl.s" mcl_api.hip \
  -Rpass-analysis=kernel-resource-usage 2> "$OUT/res.txt" ${ISA_FLAGS}
python3 - "$OUT" <<'PY'
import re, sys, subprocess
out = sys.argv[1]
rows, cur = [], None
for line in open(out + '/res.txt'):
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = dict(name=m.group(1)); rows.append(cur); continue
    for key, pat in (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r' VGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
names = [r['name'] for r in rows]
dem = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE, universal_newlines=True).stdout.split('\n')
with open(out + '/resources.tsv', 'w') as f:
    f.write('kernel\tsgpr\tvgpr\tscratch_B_per_lane\tlds_B\twaves_per_simd\n')
    for r, d in zip(rows, dem):
        d = re.sub(r'\(.*', '', d)
        f.write('%s\t%s\t%s\t%s\t%s\t%s\n' % (d, r.get('sgpr'), r.get('vgpr'), r.get('scratch'), r.get('lds'), r.get('occ')))
print(open(out + '/resources.tsv').read())
PY
