"""Lint over post-register-allocation machine IR (gfx950): vector code in front of a join block's EXEC restore.

Root cause of GPU-only incident (i) (tools/ubench/README_gpu_only_failures.md).  SI_END_CF is lowered to
`$exec = S_OR_B64 $exec, <saved mask>` at the TOP of the block that ends a divergent `if`.  The scalar register allocation may put an
SGPR copy in front of it (harmless: scalar code ignores EXEC).  The vector allocation, which runs later, looks for "the first
instruction behind the block's prologue" to place its own split copies and spills -- and SIInstrInfo::isBasicBlockPrologue does not
count a plain SGPR COPY as prologue, so the scan stops AT that copy and the vector copy lands in front of it, i.e. in front of the EXEC
restore.  It then runs under the narrowed mask of the `then` side: lanes that skipped the branch never save their value, and the reload
(full EXEC, thousands of instructions later) hands them whatever an earlier kernel left in that AGPR / scratch slot.  In incident (i):

    bb.1852:                                               ; join block of an `if` in the Newton factorisation
      $agpr20 = COPY killed $vgpr235                       ; <- inserted by the VGPR allocation: the LDS base of region R, all lanes need it
      $agpr14 = COPY killed $vgpr234
      $sgpr78_sgpr79 = COPY killed $sgpr60_sgpr61          ; <- inserted earlier by the SGPR allocation
      $exec = S_OR_B64 $exec, killed $sgpr0_sgpr1          ; <- the EXEC restore

What is reported: every block of the MIR stopped behind the last register allocation (-stop-after=amdgpu-mark-last-scratch-load: blocks
are not merged yet) in which an EXEC-dependent vector instruction -- anything with `implicit $exec` other than the SGPR <-> VGPR-lane
spill pseudos and the whole-wave-mode spills (expanded with EXEC = -1 around them: prologue by design), and any COPY into a vector register -- stands in front of the block's `$exec = S_OR_B64 $exec, ...` (or, for an else-block, its `S_OR_SAVEEXEC_B64`).

    python tools/dev/mir_lint_exec_restore.py file.mir [...]      # exit status 1 if anything is reported
    python tools/dev/mir_lint_exec_restore.py --build [--keep] [flags]   # emits the MIR of the seven translation units first (80 s on 8 cores;
                                                                         # the files -- half a gigabyte -- are removed again unless --keep)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STOP_AFTER = 'amdgpu-mark-last-scratch-load'
BLOCK = re.compile(r'^  (bb\.\d+[\w.\-]*)')
NAME = re.compile(r'^name:\s+(\S+)')
RESTORE = re.compile(r'\$exec = S_OR_B64 \$exec, |= S_OR_SAVEEXEC_B64 (killed |renamable )*\$sgpr.*implicit-def \$exec')      # SI_END_CF; SI_ELSE (the else-side's mask; `S_OR_SAVEEXEC_B64 -1` is whole-wave mode around a spill, not a join)
SCALAR_LANE_OPS = re.compile(r'SI_SPILL_WWM_\w+|= SI_SPILL_S\d+_RESTORE|SI_SPILL_S\d+_SAVE|SI_RESTORE_S32_FROM_VGPR|SI_SPILL_S32_TO_VGPR|V_READLANE_B32|V_WRITELANE_B32|V_READFIRSTLANE_B32')
VECTOR_COPY = re.compile(r'^\s+(renamable |dead |undef |early-clobber )*(\$(vgpr|agpr)\d+[\w$]*|%\d+(\.\w+)?:(vgpr|agpr|av_|vreg|areg)\w*) = COPY ')


def lint(path):
    reports = []
    func, block, pending, n_restores = None, None, [], 0
    with open(path, errors='replace') as f:
        for n, line in enumerate(f, 1):
            m = NAME.match(line)
            if m:
                func, block, pending = m.group(1), None, []
                continue
            m = BLOCK.match(line)
            if m:
                block, pending = m.group(1), []
                continue
            if block is None or not line.startswith('    '):
                continue
            if RESTORE.search(line):
                n_restores += 1
                for pn, pl in pending or ():      # (a second EXEC switch further down the block has nothing of the head in front of it)
                    reports.append((func, block, pn, pl.strip(), line.strip()))
                pending = None        # only the head of the block is looked at
                continue
            if pending is None:
                continue
            if SCALAR_LANE_OPS.search(line):
                continue
            if 'implicit $exec' in line or VECTOR_COPY.match(line):
                pending.append((n, line))
    return reports, n_restores


def build_mir(extra_flags):
    tag = 'mir' + ''.join(re.sub(r'\W+', '_', f) for f in extra_flags)
    out = os.path.join(ROOT, 'build', tag)
    os.makedirs(out, exist_ok=True)
    src = os.path.join(ROOT, 'pypownet_amd', 'csrc')
    common = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S',
              '-mllvm', '-stop-after=' + STOP_AFTER] + list(extra_flags)
    jobs = [(os.path.join(out, 'engine.mir'), common + ['-DPPN_SPLIT_BUILD', os.path.join(src, 'ppn_engine.hip')])]
    for w in (1, 2, 4):
        for nt in (0, 1):
            jobs.append((os.path.join(out, 'k_w%dn%d.mir' % (w, nt)), common + ['-DPPN_TU_W=%d' % w, '-DPPN_TU_NT=%d' % nt, os.path.join(src, 'ppn_kernel_tu.hip')]))
    procs = [(o, subprocess.Popen(c + ['-o', o], stderr=subprocess.DEVNULL)) for o, c in jobs]
    for o, p in procs:
        if p.wait() != 0:
            raise SystemExit('could not emit %s' % o)
    return [o for o, _ in jobs]


def main(argv):
    built = argv[:1] == ['--build']
    keep = '--keep' in argv
    files = build_mir([a for a in argv[1:] if a != '--keep']) if built else argv
    if not files:
        raise SystemExit(__doc__)
    total = seen = 0
    for path in files:
        reps, n_restores = lint(path)
        total += len(reps)
        seen += n_restores
        for func, block, pn, pl, restore in reps:
            print('%s:%d: %s %s: `%s` in front of `%s`' % (os.path.basename(path), pn, func, block, pl[:150], restore[:80]))
        print('%s: %d EXEC restores looked at, %d vector instruction(s) in front of one' % (os.path.basename(path), n_restores, len(reps)))
        if built and not keep:
            os.remove(path)           # (half a gigabyte for the seven units)
    if seen == 0:                     # (a compiler that prints its machine IR another way must not pass for a clean one)
        print('no EXEC restore recognised in %d file(s): the lint does not understand this MIR' % len(files))
        return 2
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
