"""N > 1 path on CPU: two gloo ranks each step their shard of a global batch (emulation build of the kernels);
rank 0 gathers the shards and the result must equal a single-process run of the whole batch."""
import os
import subprocess
import sys

import numpy as np

from helpers import ROOT, ENVS
from test_emu_engine import emu_lib  # noqa: F401

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch.distributed as dist
from pypownet_amd.batched import BatchedRunEnv
import harness
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
with harness.library(%(lib)r):
    env = BatchedRunEnv(%(envdir)r, 'level0', 7, rank=rank, world_size=world, device=0,
                        config_overrides={'solver': 'newton'})
env.reset()
for t in range(5):
    obs, done, flag, ill = env.step(np.zeros((env.batch, env.action_length), dtype=np.uint8))
full = env.gather_to_root(obs)
# single-controller mode: the root's actions are scattered, (done, flag, reward) gathered (one substation switched in env 5)
ga = None
if rank == 0:
    ga = np.zeros((7, env.action_length), dtype=np.uint8)
    ga[5, 0] = 1
ctl = env.controller_step(ga)
tot = env.all_reduce_stats([env.engine.read('N_SOLVES').sum(), env.batch])
if rank == 0:
    np.save(%(out)r, full)
    np.save(%(out)r + '.ctl.npy', np.stack([ctl[0].astype(np.float64), ctl[1].astype(np.float64), ctl[2]]))
    assert tot[1] == 7
dist.destroy_process_group()
'''


def test_two_rank_shards_equal_single_process(emu_lib, tmp_path):  # noqa: F811
    from pypownet_amd.batched import BatchedRunEnv, shard_range
    assert [shard_range(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    envdir = os.path.join(ENVS, 'default14')
    out = str(tmp_path / 'gathered.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, envdir=envdir, lib=emu_lib, out=out))
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    gathered = np.load(out)
    import harness
    with harness.library(emu_lib):
        single = BatchedRunEnv(envdir, 'level0', 7, device=0, config_overrides={'solver': 'newton'})
    single.reset()
    for t in range(5):
        obs, done, flag, ill = single.step(np.zeros((7, single.action_length), dtype=np.uint8))
    assert gathered.shape == obs.shape == (7, single.observation_length)
    assert np.array_equal(gathered, obs)
    ga = np.zeros((7, single.action_length), dtype=np.uint8)
    ga[5, 0] = 1
    done, flag, reward = single.controller_step(ga)
    ctl = np.load(out + '.ctl.npy')
    assert np.array_equal(ctl[0].astype(bool), done) and np.array_equal(ctl[1].astype(np.int32), flag)
    assert np.array_equal(ctl[2], reward)
    assert np.array_equal(single.engine.read('PRODS_NODES')[5, :1], [1]) or single.engine.read('ILLEGAL')[5] != 0
