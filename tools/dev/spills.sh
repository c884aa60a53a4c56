#!/bin/bash
# usage: tools/dev/spills.sh [extra hipcc flags] -> source lines of the scratch (spill) traffic of tools/dev/one_kernel.hip
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -gline-tables-only "$@" tools/dev/one_kernel.hip -o /tmp/one.s 2>/dev/null
python3 - <<'PY'
import re,collections
files={}; cur=None; ld=collections.Counter(); st=collections.Counter(); wl=collections.Counter(); inside=False
for l in open('/tmp/one.s'):
    m=re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"',l)
    if m: files[int(m.group(1))]=m.group(2).split('/')[-1]
    if l.startswith('_Z10ppn_kernel'): inside=True
    if inside and l.startswith('.Lfunc_end'): inside=False
    if not inside: continue
    m=re.match(r'\s+\.loc\s+(\d+)\s+(\d+)',l)
    if m: cur=(files.get(int(m.group(1))),int(m.group(2))); continue
    if 'scratch_load' in l: ld[cur]+=1
    if 'scratch_store' in l: st[cur]+=1
    if 'v_writelane' in l or 'v_readlane' in l: wl[cur]+=1
print('scratch loads %d, stores %d; v_readlane/v_writelane %d' % (sum(ld.values()), sum(st.values()), sum(wl.values())))
print('loads :', ld.most_common(14))
print('stores:', st.most_common(10))
print('lanes :', wl.most_common(14))
PY
