"""Developer aid: print the gfx950 ISA the compiler emitted for a range of source lines of one kernel.

usage: python tools/isa_region.py <file-name-in-csrc> <first-line> <last-line> [kernel-mangled-substring]
(compiles pypownet_amd/csrc/ppn_engine.hip with -gline-tables-only -S into /tmp/isa)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pypownet_amd', 'csrc', 'ppn_engine.hip')
OUT = '/tmp/isa/ppn_g.s'


def main():
    fname, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    kern = sys.argv[4] if len(sys.argv) > 4 else 'ILi2ELi0E'
    os.makedirs('/tmp/isa', exist_ok=True)
    srcs = [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC))]
    if not os.path.exists(OUT) or any(os.path.getmtime(s) > os.path.getmtime(OUT) for s in srcs):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only',
                               '-gline-tables-only', '-S', '-o', OUT, SRC] + os.environ.get('ISA_FLAGS', '').split(),
                              stderr=subprocess.DEVNULL)
    text = open(OUT).read().split('\n')
    files = {}
    for l in text:
        m = re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    inside = False
    cur = None
    keep = []
    for l in text:
        if re.match(r'^_Z10ppn_kernel' + kern + r'.*:', l):
            inside = True
        if inside and l.startswith('.Lfunc_end'):
            inside = False
        if not inside:
            continue
        m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (files.get(int(m.group(1))), int(m.group(2)))
            continue
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.Ltmp') or s.startswith('.cfi'):
            continue
        tag = '*' if (cur and cur[0] == fname and lo <= cur[1] <= hi) else ' '
        keep.append((tag, cur[1] if cur and cur[0] == fname else 0, l))
    # print instructions tagged, with a little context
    idx = [i for i, k in enumerate(keep) if k[0] == '*']
    if not idx:
        print('no instructions found')
        return
    shown = set()
    for i in idx:
        for j in range(max(0, i - 1), min(len(keep), i + 2)):
            shown.add(j)
    last = -2
    for j in sorted(shown):
        if j != last + 1:
            print('   ...')
        print('%s %4d %s' % keep[j])
        last = j


main()
