"""python tests/tools/poison_full_size.py: the differential register-poison check (engine_checks.check_register_poison) at the bench size:
4096 environments x 30 steps of random node splitting / line switching, two- and four-word kernels, both solvers.  (GPU box.)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                    # noqa: E402  (as in the pytest run: the process's HIP runtime is the one torch brings up)
torch.zeros(1, device='cuda')
import engine_checks as ec      # noqa: E402
HIP = None                     # (the product library)
for solver, mab in (('newton', 118), ('fdxb', 118), ('newton', 0), ('fdxb', 0)):
    t = time.time()
    st = ec.check_register_poison(HIP, 'default118', solver, batch=int(os.environ.get('POISON_BATCH', 4096)), steps=30, max_active_buses=mab, seed=77)
    print('4096 x 30, %s, max_active_buses %d: every field agrees between the patterns; %d game overs, %d solves, %.1f s' % (
        solver, mab, st['done'], st['solves'], time.time() - t), flush=True)
