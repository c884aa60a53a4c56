#!/usr/bin/env python
"""Lint of the kernels' optimised LLVM IR for the hazard of DESIGN 12.9 (tools/ubench/README_gpu_only_failures.md): a cycle that
contains a CONVERGENT operation (readfirstlane, readlane, a barrier, a DPP / permute in inline assembly ...) and that lanes may run a
different number of times -- i.e. a branch inside the cycle whose condition depends on the lane index decides between staying in
the cycle and leaving it, or between two different ways round it.  In a one-wavefront-per-environment kernel every such cycle must
be left and re-entered by all 64 lanes together; `simplifycfg` broke exactly that in the rollout kernel of round 5 by threading
`if (lane == 0)` at the loop's tail into `if (lane == 0)` at its head.

    python tools/dev/ir_lint_convergent.py file.ll [file.ll ...]          # IR from hipcc -O3 --cuda-device-only -emit-llvm -S
    python tools/dev/ir_lint_convergent.py --build [--flags -DPPN_WAVE_FULL_OFF]      # emits the IR of the six kernel translation units first

Divergence is approximated by a taint: the lane index (mbcnt, workitem.id), results of atomics, loads through tainted addresses,
and everything computed from them; readfirstlane / readlane / ballot results are uniform; a phi is tainted if an incoming value is,
or if it joins the two sides of a branch on a tainted condition with different values (if-then and if-then-else shapes only: a
heuristic, not LLVM's uniformity analysis).  Per-lane loops WITHOUT convergent operations (`for (r = lane; r < n;
r += 64)`) are what the kernels are made of and are not reported."""
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
CONVERGENT_CALLS = ('readfirstlane', 'readlane', 's.barrier', 'wave.barrier', 'ds.bpermute', 'ds.permute', 'permlane', 'update.dpp', 'mov.dpp',
                    'ballot', 'wave.reduce', 'set.inactive', 'writelane', 'ds.swizzle', 'icmp.i64', 'fcmp.i64')
UNIFORM_RESULTS = ('readfirstlane', 'readlane', 'ballot', 'wave.reduce', 'icmp.i64', 'fcmp.i64', 's.getreg', 's.memtime', 's.memrealtime')
ASM_CONVERGENT = ('dpp', 'readlane', 'permlane', 'bpermute', 'swizzle', 'readfirstlane', 'row_newbcast', 'row_shr', 'row_bcast')
VAL = re.compile(r'%[\w.]+')


def functions(path):
    name, body = None, []
    with open(path) as f:
        for line in f:
            if line.startswith('define '):
                m = re.search(r'@([\w.$]+)\(', line)
                name, body = (m.group(1) if m else '?'), []
                kernel = 'amdgpu_kernel' in line
            elif name is not None:
                if line.startswith('}'):
                    if kernel:
                        yield name, body
                    name = None
                else:
                    body.append(line.rstrip('\n'))


def parse(body):
    """-> blocks: label -> dict(insts=[(dst or None, text)], succ=[...], cond=%x or None)"""
    blocks, order, cur = {}, [], None
    for line in body:
        m = re.match(r'^([\w.]+):', line)
        if m:
            cur = m.group(1)
        elif cur is None:
            cur = 'entry'      # (the unnamed first block: its label is the next free number; only used as the DFS root)
        if cur not in blocks:
            blocks[cur] = dict(insts=[], succ=[], cond=None)
            order.append(cur)
        if m or not line.startswith('  '):
            continue
        text = line.strip()
        d = re.match(r'^(%[\w.]+) = ', text)
        blocks[cur]['insts'].append((d.group(1) if d else None, text))
        if text.startswith('br '):
            labels = re.findall(r'label %([\w.]+)', text)
            blocks[cur]['succ'] = labels
            c = re.match(r'^br i1 (%[\w.]+),', text)
            blocks[cur]['cond'] = c.group(1) if (c and len(labels) == 2) else None
        elif text.startswith('switch '):
            blocks[cur]['succ'] = re.findall(r'label %([\w.]+)', text)
            c = re.match(r'^switch \w+ (%[\w.]+),', text)
            blocks[cur]['cond'] = c.group(1) if c else None
    return blocks, order


def is_convergent(text):
    if ' asm ' in text:
        m = re.search(r'asm [\w ]*"([^"]*)"', text)
        return bool(m) and any(k in m.group(1) for k in ASM_CONVERGENT)
    if 'call ' not in text:
        return False
    m = re.search(r'@llvm\.amdgcn\.([\w.]+)\(', text)
    return bool(m) and any(m.group(1).startswith(k) for k in CONVERGENT_CALLS)


def taint(blocks, order):
    preds = defaultdict(list)
    for b in order:
        for s in blocks[b]['succ']:
            preds[s].append(b)
    t = {}      # value -> the tainted operand (or the reason) that made it lane-dependent
    changed = True
    while changed:
        changed = False
        for b in order:
            for dst, text in blocks[b]['insts']:
                if dst is None or dst in t:
                    continue
                rhs = text.split(' = ', 1)[1]
                call = re.search(r'@llvm\.amdgcn\.([\w.]+)\(', rhs)
                hot = False
                if call and any(call.group(1).startswith(k) for k in UNIFORM_RESULTS):
                    hot = False
                elif call and (call.group(1).startswith('mbcnt') or call.group(1).startswith('workitem.id')):
                    hot = 'lane index'
                elif rhs.startswith('atomicrmw') or rhs.startswith('cmpxchg'):
                    hot = 'atomic result'
                elif rhs.startswith('phi '):
                    inc = [(v.strip(), p) for v, p in re.findall(r'\[ ([^,]+), %([\w.]+) \]', rhs)]
                    hot = next((v for v, _ in inc if v in t), False)
                    if not hot and len({v for v, _ in inc}) > 1:
                        # joins of a lane-dependent branch (if-then: the branching block is itself a predecessor; if-then-else: two
                        # predecessors with one lane-dependent branching block in front of both) with different values coming in
                        for v, p in inc:
                            c = blocks.get(p, {}).get('cond')
                            if c in t and b in blocks[p]['succ']:
                                others = [s_ for s_ in blocks[p]['succ'] if s_ != b]
                                vals = dict((p_, v_) for v_, p_ in inc)
                                # the predecessors of this block that the OTHER side of the branch can arrive through (without passing
                                # this block): a different value on one of them makes the phi lane-dependent
                                seen, work, reach = set(others), list(others), set()
                                while work and len(seen) < 4000:
                                    x = work.pop()
                                    if x in vals:
                                        reach.add(x)
                                    for s_ in blocks.get(x, {}).get('succ', []):
                                        if s_ != b and s_ not in seen:
                                            seen.add(s_)
                                            work.append(s_)
                                if any(vals[r] != v for r in reach) or len(seen) >= 4000:
                                    hot = 'join of the branch on ' + c
                            for v2, p2 in inc:
                                if p2 != p and v2 != v and len(preds[p]) == 1 and preds[p] == preds[p2] and blocks.get(preds[p][0], {}).get('cond') in t:
                                    hot = 'join of the branch on ' + blocks[preds[p][0]]['cond']
                else:
                    hot = next((v for v in VAL.findall(rhs) if v in t), False)
                if hot:
                    t[dst] = hot
                    changed = True
    return t, preds


def loops(blocks, order, preds):
    """natural loops of the back edges found by an iterative DFS from the first block: header -> set of blocks"""
    root = order[0]
    color, back = {}, []
    stack = [(root, iter(blocks[root]['succ']))]
    color[root] = 1
    while stack:
        b, it = stack[-1]
        for s in it:
            if s not in blocks:
                continue
            if color.get(s, 0) == 0:
                color[s] = 1
                stack.append((s, iter(blocks[s]['succ'])))
                break
            if color[s] == 1:
                back.append((b, s))
        else:
            color[b] = 2
            stack.pop()
    out = defaultdict(set)
    for u, h in back:
        body = {h, u}
        work = [u] if u != h else []
        while work:
            x = work.pop()
            for p in preds[x]:
                if p not in body:
                    body.add(p)
                    work.append(p)
        out[h] |= body
    return out, back


def lint(path):
    findings = []
    n_kernels = n_loops = n_conv_loops = 0
    for name, body in functions(path):
        n_kernels += 1
        blocks, order = parse(body)
        if not order:
            continue
        t, preds = taint(blocks, order)
        lp, back = loops(blocks, order, preds)
        conv_blocks = {b for b in order if any(is_convergent(x) for _, x in blocks[b]['insts'])}
        for h, body_set in lp.items():
            n_loops += 1
            inside = conv_blocks & body_set
            if not inside:
                continue
            n_conv_loops += 1
            for b in body_set:
                c = blocks[b]['cond']
                if c is None or c not in t:
                    continue
                succ = blocks[b]['succ']
                leaves = [s for s in succ if s not in body_set]
                to_header = [s for s in succ if s == h]
                # a lane-dependent choice between leaving and staying, or between the header and another way on
                if (leaves and len(leaves) < len(succ)) or (to_header and len(set(succ)) > 1):
                    chain, v = [], c
                    while v in t and len(chain) < 12:
                        chain.append(v)
                        v = t[v]
                        if v.startswith('join of the branch on '):
                            chain.append('[' + v + ']')
                            v = v[len('join of the branch on '):]
                    chain.append(v)
                    findings.append((name, h, b, c, sorted(inside)[:4], len(body_set), ' <- '.join(chain)))
    return n_kernels, n_loops, n_conv_loops, findings


def build_ir(flags):
    out = os.path.join(ROOT, 'build', 'ir')
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs, files = [], []
    for w in (1, 2, 4):
        for nt in (0, 1):
            f = os.path.join(out, 'ppn_kernels_w%dn%d.ll' % (w, nt))
            files.append(f)
            procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-DPPN_TU_W=%d' % w, '-DPPN_TU_NT=%d' % nt] + flags +
                                          ['--cuda-device-only', '-emit-llvm', '-S', os.path.join(ROOT, 'pypownet_amd', 'csrc', 'ppn_kernel_tu.hip'), '-o', f],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed')
    return files


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith('.ll')]
    if '--build' in args:
        flags = args[args.index('--flags') + 1:] if '--flags' in args else []
        files = build_ir([f for f in flags if not f.endswith('.ll')])
    total = 0
    for f in files:
        nk, nl, nc, found = lint(f)
        # KNOWN AND BENIGN: the polling loop of the step server (K_SERVE = kind 11, a cycle of ~15 blocks).  Its idle-timeout exit reads
        # `if (timed out) { if (lane == 0) store(stop = 2); quit = 1; break; }`: the values that leave the lane-0 triangle are the same on both
        # sides ((4, 1) either way), which the taint cannot see; every lane takes the same exit.
        benign = [x for x in found if re.search(r'ppn_kernelILi\dELi11ELi\dE', x[0]) and x[5] <= 32]
        found = [x for x in found if x not in benign]
        print('%s: %d kernels, %d cycles, %d of them with a convergent operation inside, %d reported%s' % (os.path.basename(f), nk, nl, nc, len(found),
              (' (+ %d known benign: the step server\'s polling loop)' % len(benign)) if benign else ''))
        for name, h, b, c, inside, size, chain in found:
            print('    %s: cycle with header %%%s (%d blocks; convergent operations in %s): block %%%s branches on lane-dependent %s between staying and leaving'
                  % (name, h, size, ', '.join('%' + x for x in inside), b, c))
            if '--why' in sys.argv:
                print('        ' + chain)
        total += len(found)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
