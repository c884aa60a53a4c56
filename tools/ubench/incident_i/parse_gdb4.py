"""python parse_gdb4.py <gdb4 output>: per wave, where it stands and whether the node bytes (LDS 186..557) hold anything but 0."""
import re, sys
txt = open(sys.argv[1], errors='replace').read()
a = txt.split('==== all waves')[1].split('==== all lds')[0]
where = {}
cur = None
for line in a.splitlines():
    m = re.match(r'Thread (\d+)', line)
    if m: cur = int(m.group(1)); where[cur] = []; continue
    m = re.match(r'#(\d+)\s+(.*)', line)
    if m and cur is not None: where[cur].append(m.group(2)[:110])
b = txt.split('==== all lds')[1]
lds = {}
cur = None
for line in b.splitlines():
    m = re.match(r'Thread (\d+)', line)
    if m: cur = int(m.group(1)); lds[cur] = []; continue
    m = re.match(r'local#0x[0-9a-f]+:\s+(.*)', line)
    if m and cur is not None: lds[cur] += [int(x, 16) for x in m.group(1).split()]
n_bad = 0
for t in sorted(lds):
    by = lds[t]
    if len(by) < 558: continue
    nz = [(i, by[i]) for i in range(186, 558) if by[i]]
    st_on = sum(by[:186])
    w = ' | '.join(where.get(t, [])[:2])
    print('thread %3d  lines on %3d  nonzero node bytes %3d %s  at %s' % (t, st_on, len(nz), str(nz[:6]) if nz else '', w))
    n_bad += bool(nz)
print('waves with LDS read: %d, with nonzero node bytes: %d' % (sum(len(v) >= 558 for v in lds.values()), n_bad))
