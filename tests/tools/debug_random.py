"""Debug aid: lock-step GPU engine vs C oracle under random actions; prints the details of the first divergence."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_env
import engine_checks as ec
from pypownet_amd.engine import Engine

envname, steps, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = sys.argv[4] if len(sys.argv) > 4 else None
case, cfg, chronics = load_env(envname, conf={'solver': 'newton'})
case.ntopo_offset_lines = case.nP + case.nL + 2 * case.nl
eng = Engine(case, cfg, batch, chronics=chronics, _lib_path=lib)
orc = Engine(case, cfg, batch, chronics=chronics, _lib_path=os.path.join(ROOT, 'oracle', '_build', 'liboracle.so'), _lib_prefix='orc_')
rng = np.random.default_rng(1234)
eng.reset(); orc.reset()
for t in range(steps):
    acts = ec.random_actions(case, rng, batch)
    pe = {f: eng.read(f).copy() for f in ('N_ITERS', 'N_SOLVES')}
    po = {f: orc.read(f).copy() for f in ('N_ITERS', 'N_SOLVES')}
    eng.step(acts, auto_reset=False); orc.step(acts, auto_reset=False)
    a, b = eng.read('DONE'), orc.read('DONE')
    bad = np.where(a != b)[0]
    for e in bad:
        print('step', t, 'env', e, 'DONE gpu/orc', a[e], b[e], 'FLAG', eng.read('FLAG')[e], orc.read('FLAG')[e],
              'iters this step gpu/orc', eng.read('N_ITERS')[e] - pe['N_ITERS'][e], orc.read('N_ITERS')[e] - po['N_ITERS'][e],
              'solves', eng.read('N_SOLVES')[e] - pe['N_SOLVES'][e], orc.read('N_SOLVES')[e] - po['N_SOLVES'][e],
              'depth', eng.read('CASCADE_DEPTH')[e], orc.read('CASCADE_DEPTH')[e])
        vm_e, vm_o = eng.read('VM')[e], orc.read('VM')[e]
        print('   max |dVm|', np.nanmax(np.abs(vm_e - vm_o)), 'vm range gpu', np.nanmin(vm_e), np.nanmax(vm_e), 'orc', np.nanmin(vm_o), np.nanmax(vm_o))
    if len(bad):
        import ctypes as C
        out = np.zeros((batch, 16), dtype=np.int64)
        eng._check(eng._lib.ppn_read(eng._h, 100, out.ctypes.data, out.nbytes, 1, 0), 'read prof')
        for e in bad:
            print('   normF history slots A:', out[e, :7].view(np.float64), ' B:', out[e, 7:14].view(np.float64))
            print('   solve counter', out[e, 15], 'rc/iters history (latest last) %x' % out[e, 14], 'exit code %x' % out[e, 13])
            ls_e, ls_o = eng.read('LINES_STATUS')[e], orc.read('LINES_STATUS')[e]
            print('   lines off gpu', np.where(ls_e == 0)[0], 'orc', np.where(ls_o == 0)[0])
            for f in ('PRODS_NODES', 'LOADS_NODES', 'LINES_OR_NODES', 'LINES_EX_NODES'):
                print('  ', f, np.where(eng.read(f)[e] != 0)[0], np.where(orc.read(f)[e] != 0)[0])
        break
    eng.process_game_over(); orc.process_game_over()
print('done', t)
