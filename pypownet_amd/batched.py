"""BatchedRunEnv: B independent pypownet environments on one GPU, sharded over the GPUs of a node.

The load-flow step has NO exchange between environments, so the multi-GPU layout is a plain
contiguous split of the global batch: rank r owns environments [r*B/W, (r+1)*B/W) and steps them with its own
engine; there is no collective inside the step loop.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests) is only used by ``gather_to_root`` / ``all_reduce_stats`` when a single
controller wants the results of every shard.
"""
import os

import numpy as np
import yaml

from .case import Case
from .chronic import Chronic
from .engine import Engine


def shard_range(global_batch, rank, world_size):
    """Contiguous split; the first (global_batch % world_size) ranks own one extra environment."""
    base, extra = divmod(int(global_batch), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def local_device(rank):
    """GPU of this process: LOCAL_RANK when a launcher set it (one process per GPU of the node), else the rank modulo the
    number of GPUs of the node (PPN_DEVICES_PER_NODE, default 8)."""
    if 'LOCAL_RANK' in os.environ:
        return int(os.environ['LOCAL_RANK'])
    return int(rank) % max(1, int(os.environ.get('PPN_DEVICES_PER_NODE', '8')))


def default_assignment(env_ids, chronics):
    """environment e plays chronic (e mod n) from row (37 e) mod T."""
    env_ids = np.asarray(env_ids)
    slots = (env_ids % len(chronics)).astype(np.int32)
    T = np.array([c.n_timesteps for c in chronics])[slots]
    return slots, ((env_ids * 37) % T).astype(np.int32)


class BatchedRunEnv(object):
    def __init__(self, parameters_folder, game_level, global_batch, rank=0, world_size=1, device=None,
                 config_overrides=None, thermal_limits=None, **rule_kw):
        level = os.path.join(parameters_folder, game_level)
        grid = os.path.join(level, 'reference_grid.py')
        if not os.path.exists(grid):
            grid = os.path.join(level, 'reference_grid.json')
        self.case = Case.from_file(grid)
        with open(os.path.join(level, 'configuration.yaml')) as f:
            self.conf = yaml.safe_load(f)
        if config_overrides:
            self.conf.update(config_overrides)
        cdir = os.path.join(level, 'chronics')
        self.chronics = [Chronic(os.path.join(cdir, c)) for c in sorted(os.listdir(cdir))
                         if os.path.isdir(os.path.join(cdir, c)) or c.endswith('.npz')]
        self.rank, self.world_size, self.global_batch = rank, world_size, global_batch
        self.first, self.last = shard_range(global_batch, rank, world_size)
        self.batch = self.last - self.first
        self.env_ids = np.arange(self.first, self.last)
        self.engine = Engine(self.case, self.conf, self.batch, device=local_device(rank) if device is None else device,
                             chronics=self.chronics, thermal_limits=thermal_limits, **rule_kw)
        self.action_length = self.case.action_length
        self.observation_length = self.case.observation_length

    def reset(self):
        slots, t0 = default_assignment(self.env_ids, self.chronics)
        self.engine.reset(chronic_slot=slots, t0=t0)
        return self.engine.observations()

    def step(self, actions, auto_reset=True, want_obs=True):
        """actions: uint8 [batch x action_length] of THIS shard.  Returns (obs|None, done, flag, illegal)."""
        self.engine.step(actions, auto_reset=auto_reset)
        done, flag, ill = self.engine.read('DONE'), self.engine.read('FLAG'), self.engine.read('ILLEGAL')
        return (self.engine.observations() if want_obs else None), done.astype(bool), flag, ill

    def search(self, candidate_actions, want_obs=False):
        """Topology-action search (what the reference's search agents do with one ``simulate`` call per candidate,
        pypownet/agent.py:161-325): ``candidate_actions`` uint8 [batch x K x action_length]; every candidate is
        simulated from the current state of its environment in one launch.  Returns (rewards [batch x K], done
        [batch x K], flag [batch x K], obs [batch x K x observation_length] | None)."""
        a = np.ascontiguousarray(candidate_actions, dtype=np.uint8)
        B, K = a.shape[0], a.shape[1]
        assert B == self.batch and a.shape[2] == self.action_length
        env_ids = np.repeat(np.arange(B, dtype=np.int32), K)
        e = self.engine
        e.simulate_candidates(a.reshape(B * K, -1), env_ids)
        obs = e.observations(simulation=2).reshape(B, K, -1) if want_obs else None
        return (e.read('REWARD', simulation=2).sum(axis=1).reshape(B, K), e.read('DONE', simulation=2).astype(bool).reshape(B, K),
                e.read('FLAG', simulation=2).reshape(B, K), obs)

    def rewards(self, do_sum=True, simulation=False):
        """Reward of the last step of every environment, computed on the device with the reference's shipped
        five-component formula (coefficients: Engine.set_reward; default = default14's scaled by the number of
        substations, which is what default118 ships): [batch] sums or [batch x 5] components."""
        r = self.engine.read('REWARD', simulation=simulation)
        return r.sum(axis=1) if do_sum else r

    def simulate(self, actions):
        self.engine.simulate(actions)
        e = self.engine
        return e.observations(simulation=True), e.read('DONE', simulation=True).astype(bool), \
            e.read('FLAG', simulation=True), e.read('ILLEGAL', simulation=True)

    # ---- single-controller helpers (the only collectives of the design) ---------------------------------------
    def gather_to_root(self, array, root=0):
        """Gather a per-environment array of every shard on `root` (variable shard sizes are padded)."""
        import torch
        import torch.distributed as dist
        if self.world_size == 1:
            return np.asarray(array)
        a = np.ascontiguousarray(array)
        sizes = [shard_range(self.global_batch, r, self.world_size) for r in range(self.world_size)]
        mx = max(b - a_ for a_, b in sizes)
        pad = np.zeros((mx,) + a.shape[1:], dtype=a.dtype)
        pad[:a.shape[0]] = a
        t = torch.from_numpy(pad)
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world_size)] if self.rank == root else None
        dist.gather(t, out, dst=root)
        if self.rank != root:
            return None
        return np.concatenate([o.cpu().numpy()[:b - a_] for o, (a_, b) in zip(out, sizes)])

    def scatter_from_root(self, array, root=0):
        """The other half of the single-controller mode: `root` holds a [global_batch x ...] array (the
        actions its policy chose for every environment), each rank receives the rows of its shard."""
        import torch
        import torch.distributed as dist
        if self.world_size == 1:
            return np.asarray(array)
        sizes = [shard_range(self.global_batch, r, self.world_size) for r in range(self.world_size)]
        mx = max(b - a_ for a_, b in sizes)
        meta = [None]
        if self.rank == root:
            a = np.ascontiguousarray(array)
            assert a.shape[0] == self.global_batch
            meta = [(a.shape[1:], a.dtype.str)]
        dist.broadcast_object_list(meta, src=root)
        tail, dt = meta[0]
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        recv = torch.empty((mx,) + tuple(tail), dtype=torch.from_numpy(np.empty(0, dtype=np.dtype(dt))).dtype, device=dev)
        parts = None
        if self.rank == root:
            parts = []
            for a_, b in sizes:
                pad = np.zeros((mx,) + tuple(tail), dtype=np.dtype(dt))
                pad[:b - a_] = a[a_:b]
                parts.append(torch.from_numpy(pad).to(dev))
        dist.scatter(recv, parts, src=root)
        return recv.cpu().numpy()[:self.batch]

    def controller_step(self, global_actions=None, root=0, auto_reset=True):
        """One step of the single-controller mode: scatter the root's [global_batch x action_length] actions, step the
        shard, gather (done, flag, reward) of every environment on the root.  Returns (done, flag, reward) on the root,
        None elsewhere.  Observations stay on their GPU (gather_to_root(env.engine.observations()) fetches them)."""
        acts = self.scatter_from_root(global_actions, root=root)
        self.engine.step(acts, auto_reset=auto_reset)
        e = self.engine
        res = np.stack([e.read('DONE').astype(np.float64), e.read('FLAG').astype(np.float64), e.read('REWARD').sum(axis=1)], axis=1)
        full = self.gather_to_root(res, root=root)
        if full is None:
            return None
        return full[:, 0].astype(bool), full[:, 1].astype(np.int32), full[:, 2]

    def all_reduce_stats(self, values):
        import torch
        import torch.distributed as dist
        if self.world_size == 1:
            return np.asarray(values, dtype=np.float64)
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()
