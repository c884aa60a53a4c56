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
                 config_overrides=None, thermal_limits=None, device_exchange=False, **rule_kw):
        # device_exchange: controller_step runs its scatter / gather on DEVICE tensors (needs the "nccl" = RCCL backend) and the
        # root gets torch CUDA tensors back; False (default): host arrays in, numpy out, whatever process group the surrounding
        # program has set up.  It is a property of THIS environment batch, set the same on every rank (the return
        # type must not depend on the global torch.distributed state).
        self.device_exchange = bool(device_exchange)
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
        self.device = local_device(rank) if device is None else int(device)
        self.engine = Engine(self.case, self.conf, self.batch, device=self.device,
                             chronics=self.chronics, thermal_limits=thermal_limits, **rule_kw)
        self.action_length = self.case.action_length
        self.observation_length = self.case.observation_length

    def reset(self):
        slots, t0 = default_assignment(self.env_ids, self.chronics)
        self.engine.reset(chronic_slot=slots, t0=t0)
        return self.engine.observations()

    # ---- tensors in, tensors out -----------------------------------------------------------------------------------------
    # Every call below takes either host arrays (numpy: the actions cross PCIe, results come back as numpy) or DEVICE tensors
    # -- a torch CUDA tensor, or any object that exports ``__dlpack__`` (cupy, jax, ...), which is imported zero-copy with
    # torch.from_dlpack -- in which case nothing leaves the GPU: the engine reads the actions where they are
    # (``Engine.step_device``) and writes observations / done / flag / reward into torch CUDA tensors that are returned
    # (``Engine.read_into_device``, ``Engine.observations_into_device``); those export ``__dlpack__`` themselves.
    def _as_device_tensor(self, x):
        """None if x is host data; else a contiguous torch CUDA uint8 tensor viewing (not copying) x.  The tensor has to live
        on the engine's GPU: its raw address is handed to the engine."""
        if isinstance(x, np.ndarray) or isinstance(x, (list, tuple)):
            return None
        import torch
        if not torch.is_tensor(x):
            if not hasattr(x, '__dlpack__'):
                return None
            x = torch.from_dlpack(x)
        if not x.is_cuda:
            return None
        if x.device.index != self.device:
            raise ValueError('actions live on cuda:%s, this environment batch on cuda:%d' % (x.device.index, self.device))
        if x.dtype != torch.uint8:
            x = (x != 0).to(torch.uint8)
        return x.contiguous()

    def _device_results(self, want_obs, simulation=False, layout='full', obs_dtype=None, obs=None):
        """``obs``: the observation tensor the step kernel has already filled (ppn_step_observe) -- nothing is gathered then."""
        import torch
        e = self.engine
        rows = e._n_candidates if int(simulation) == 2 else self.batch
        dev = 'cuda:%d' % self.device
        done = torch.empty((rows,), dtype=torch.uint8, device=dev)
        flag = torch.empty((rows,), dtype=torch.int32, device=dev)
        ill = torch.empty((rows,), dtype=torch.int32, device=dev)
        rew = torch.empty((rows, 5), dtype=torch.float64, device=dev)
        e.read_into_device('DONE', done.data_ptr(), done.numel(), simulation=simulation)
        e.read_into_device('FLAG', flag.data_ptr(), 4 * flag.numel(), simulation=simulation)
        e.read_into_device('ILLEGAL', ill.data_ptr(), 4 * ill.numel(), simulation=simulation)
        e.read_into_device('REWARD', rew.data_ptr(), 8 * rew.numel(), simulation=simulation)
        if want_obs and obs is None:
            tdt = torch.float32 if self._is_f32(obs_dtype) else torch.float64
            obs = torch.empty((rows, e.observation_length(layout)), dtype=tdt, device=dev)
            e.observations_into_device(obs.data_ptr(), obs.numel() * obs.element_size(), simulation=simulation, layout=layout,
                                       dtype=np.float32 if tdt == torch.float32 else np.float64)
        return obs, done, flag, ill, rew

    @staticmethod
    def _is_f32(obs_dtype):
        """obs_dtype as given by a caller: None (float64), a numpy dtype / its name ('float32'), or a torch dtype."""
        if obs_dtype is None:
            return False
        try:
            import torch
            if obs_dtype in (torch.float32, torch.float64):
                return obs_dtype == torch.float32
        except ImportError:
            pass
        dt = np.dtype(obs_dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError('obs_dtype must be float32 or float64, got %s' % dt)
        return dt == np.dtype(np.float32)

    def step(self, actions, auto_reset=True, want_obs=True, layout='full', obs_dtype=None):
        """actions: uint8 [batch x action_length] of THIS shard, host array or device tensor (see above).
        Returns (obs|None, done, flag, illegal) -- numpy for host actions, torch CUDA tensors for device actions (the engine's
        stream is synchronised before they are returned)."""
        t = self._as_device_tensor(actions)
        if t is None:
            self.engine.step(actions, auto_reset=auto_reset)
            done, flag, ill = self.engine.read('DONE'), self.engine.read('FLAG'), self.engine.read('ILLEGAL')
            obs = None
            if want_obs:
                obs = self.engine.observations(layout=layout, dtype=np.float32 if self._is_f32(obs_dtype) else np.float64)
            return obs, done.astype(bool), flag, ill
        assert tuple(t.shape) == (self.batch, self.action_length)
        self._sync_torch(t)
        obs = None
        if want_obs and int(auto_reset) in (0, 1):
            # what RunEnv.step returns in ONE launch: every environment's workgroup writes its observation row behind its step
            import torch
            tdt = torch.float32 if self._is_f32(obs_dtype) else torch.float64
            obs = torch.empty((self.batch, self.engine.observation_length(layout)), dtype=tdt, device='cuda:%d' % self.device)
            self.engine.step_observe_device(t.data_ptr(), obs.data_ptr(), obs.numel() * obs.element_size(), auto_reset=bool(auto_reset),
                                            layout=layout, dtype=np.float32 if tdt == torch.float32 else np.float64)
        else:
            self.engine.step_device(t.data_ptr(), auto_reset=auto_reset)
        obs, done, flag, ill, _ = self._device_results(want_obs, layout=layout, obs_dtype=obs_dtype, obs=obs)
        self.engine.wait()         # (the engine stream only: restarts owed by a deferred auto-reset stay owed)
        return obs, done.bool(), flag, ill

    @staticmethod
    def _sync_torch(t):
        """The engine launches on its own HIP stream: work queued on torch's current stream that produces `t` must be done."""
        import torch
        torch.cuda.current_stream(t.device).synchronize()

    def rollout(self, actions, n_steps=None, auto_reset=True):
        """Open-loop rollout (include/ppn.h, ppn_rollout): ``actions`` uint8 ``[n_steps x batch x action_length]`` (one matrix per
        step) or ``[batch x action_length]`` replayed ``n_steps`` times (the do-nothing agent of the reference's Runner), host
        array or device tensor; every environment plays its steps back to back in ONE launch.  Returns (cumulative reward since
        reset [batch], done / flag of the LAST step, steps executed since reset [batch])."""
        t = self._as_device_tensor(actions)
        e = self.engine
        if t is None:
            e.rollout(actions, n_steps=n_steps, auto_reset=auto_reset)
        else:
            per_step = t.dim() == 3
            if per_step:
                n_steps = t.shape[0] if n_steps is None else n_steps
                assert tuple(t.shape[1:]) == (self.batch, self.action_length)
            else:
                assert n_steps is not None and tuple(t.shape) == (self.batch, self.action_length)
            self._sync_torch(t)
            e.rollout_device(t.data_ptr(), n_steps, per_step_actions=per_step, auto_reset=auto_reset)
        return e.read('RETURN'), e.read('DONE').astype(bool), e.read('FLAG'), e.read('N_STEPS')

    def policy_actions(self, policy='line_relief', params=(1.0,)):
        """The built-in device policy's choice for the CURRENT state of every environment of this shard (include/ppn.h,
        ppn_policy_actions) as a torch CUDA uint8 tensor [batch x action_length] -- ready for ``step``."""
        import torch
        out = torch.empty((self.batch, self.action_length), dtype=torch.uint8, device='cuda:%d' % self.device)
        self._sync_torch(out)
        self.engine.policy_actions(policy, list(params), out.data_ptr())
        self.engine.wait()
        return out

    def rollout_policy(self, n_steps, policy='line_relief', params=(1.0,)):
        """``n_steps`` closed-loop steps of a built-in device policy in ONE launch, every environment on its own clock (include/ppn.h,
        ppn_rollout_policy): same trajectories as ``step(policy_actions())`` round by round with the fused restart.  Returns
        (cumulative reward since reset [batch], done / flag of the LAST step, steps executed since reset [batch])."""
        e = self.engine
        e.rollout_policy(policy, list(params), int(n_steps))
        return e.read('RETURN'), e.read('DONE').astype(bool), e.read('FLAG'), e.read('N_STEPS')

    # ---- asynchronous session: send / recv (include/ppn.h, ppn_async_start) ----------------------------------------------
    # EnvPool-style boundary for policies that live outside the engine: every environment is stepped again as soon as the policy has
    # decided for it; nobody waits for the longest cascade of the batch.  Device tensors only (the policy is a torch module on this
    # GPU); Engine.send / Engine.recv take host arrays.
    def async_start(self, layout='full', obs_dtype=None, workgroups=0, idle_timeout_ms=0):
        """Starts the resident step server.  The observation rows ([batch x observation_length(layout)], float64 or float32) and the
        report rows ([batch x 3]: done, flag, reward sum) live in torch CUDA tensors of this object: ``self.async_obs`` / ``self.async_report``;
        row e is rewritten by every step of environment e and is stable from the recv that returned e to the send that sends it again."""
        import torch
        dev = 'cuda:%d' % self.device
        tdt = torch.float32 if self._is_f32(obs_dtype) else torch.float64
        self.async_obs = torch.zeros((self.batch, self.engine.observation_length(layout)), dtype=tdt, device=dev)
        self.async_report = torch.zeros((self.batch, 3), dtype=torch.float64, device=dev)
        self._async_ids = torch.zeros((self.batch,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(self.device)      # (the zero fills ran on torch's stream; the server runs on its own)
        self.engine.async_start(self.async_obs.data_ptr(), self.async_obs.numel() * self.async_obs.element_size(), self.async_report.data_ptr(),
                                layout=layout, dtype=np.float32 if tdt == torch.float32 else np.float64, workgroups=workgroups,
                                idle_timeout_ms=idle_timeout_ms)
        self._async_stream = torch.cuda.ExternalStream(self.engine.async_stream_ptr(), device=dev)

    def async_stream(self):
        """The session's HIP stream as a torch stream: run the policy under ``torch.cuda.stream(env.async_stream())`` and every send
        is ordered behind the kernels that produced its actions, every gather of a recv behind the copy of its ids -- no host
        synchronisation anywhere in the loop."""
        return self._async_stream

    def send(self, env_ids, actions, rows_by_env=False):
        """env_ids: int tensor / array [n]; actions: uint8 CUDA tensor [n x action_length] (row i for env_ids[i]) or, rows_by_env,
        [batch x action_length] indexed by environment.  The tensor must have been produced on ``async_stream()`` (or that stream must
        wait for its producer) and must not be overwritten before the stream has passed this call."""
        import torch
        ids = env_ids.cpu().numpy() if torch.is_tensor(env_ids) else np.asarray(env_ids)
        t = self._as_device_tensor(actions)
        if t is None:
            self.engine.send(ids, actions, rows_by_env=rows_by_env)
            return
        assert tuple(t.shape) == ((self.batch if rows_by_env else len(ids)), self.action_length)
        cur = torch.cuda.current_stream(t.device)
        if cur.cuda_stream != self._async_stream.cuda_stream:
            self._async_stream.wait_stream(cur)      # the actions were produced on another stream: the session's stream waits for it
        self._keep_alive = t                          # (a temporary made by .contiguous() must outlive the enqueue kernel)
        self.engine.send_device(ids, t.data_ptr(), rows_by_env=rows_by_env)

    def recv(self, min_ready=1, max_n=None, timeout_ms=-1, gather=True):
        """-> (ids [n] int32 CUDA tensor, observation rows [n x len], report rows [n x 3]) of >= min_ready environments whose step is
        complete (gather=False: (ids, None, None) -- index ``async_obs`` / ``async_report`` yourself).  The gathers are queued on the
        session's stream behind the copy of the ids; the host does not wait for them."""
        import torch
        ids_h = self.engine.recv(min_ready=min_ready, max_n=max_n, timeout_ms=timeout_ms, ids_device_ptr=self._async_ids.data_ptr())
        n = len(ids_h)
        with torch.cuda.stream(self._async_stream):
            ids = self._async_ids[:n].clone()        # (the staging tensor is overwritten by the next recv)
            if not gather:
                return ids, None, None
            ix = ids.long()
            return ids, self.async_obs.index_select(0, ix), self.async_report.index_select(0, ix)

    def async_stop(self):
        self.engine.async_stop()

    def search(self, candidate_actions, want_obs=False):
        """Topology-action search (what the reference's search agents do with one ``simulate`` call per candidate,
        pypownet/agent.py:161-325): ``candidate_actions`` uint8 [batch x K x action_length], host array or device tensor; every
        candidate is simulated from the current state of its environment in one launch.  Returns (rewards [batch x K], done
        [batch x K], flag [batch x K], obs [batch x K x observation_length] | None) -- numpy or torch CUDA tensors like step()."""
        t = self._as_device_tensor(candidate_actions)
        e = self.engine
        if t is None:
            a = np.ascontiguousarray(candidate_actions, dtype=np.uint8)
            B, K = a.shape[0], a.shape[1]
            assert B == self.batch and a.shape[2] == self.action_length
            env_ids = np.repeat(np.arange(B, dtype=np.int32), K)
            e.simulate_candidates(a.reshape(B * K, -1), env_ids)
            obs = e.observations(simulation=2).reshape(B, K, -1) if want_obs else None
            return (e.read('REWARD', simulation=2).sum(axis=1).reshape(B, K), e.read('DONE', simulation=2).astype(bool).reshape(B, K),
                    e.read('FLAG', simulation=2).reshape(B, K), obs)
        B, K = int(t.shape[0]), int(t.shape[1])
        assert B == self.batch and int(t.shape[2]) == self.action_length
        self._sync_torch(t)
        e.simulate_candidates_device(t.data_ptr(), np.repeat(np.arange(B, dtype=np.int32), K))
        obs, done, flag, ill, rew = self._device_results(want_obs, simulation=2)
        e.sync()
        return (rew.sum(dim=1).reshape(B, K), done.bool().reshape(B, K), flag.reshape(B, K),
                obs.reshape(B, K, -1) if want_obs else None)

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
        """One step of the single-controller mode (BASELINE.json configs[3]): scatter the root's
        [global_batch x action_length] actions, step the shard, gather (done, flag, reward) of every environment on the root.
        Returns (done, flag, reward) on the root, None elsewhere.  Observations stay on their GPU
        (gather_to_root(env.engine.observations()) fetches them).
        With the "nccl" backend (RCCL over xGMI) the whole exchange is device resident: the root passes a torch CUDA tensor (a
        host array is uploaded once), every rank receives its rows straight into device memory, the engine reads them there
        (``Engine.step_device``) and writes done / flag / reward into device tensors that are gathered as they are; the root
        gets torch CUDA tensors back.  With "gloo" (CPU tests) the same exchange runs through host arrays."""
        if self.device_exchange:      # (also on a world of ONE rank: the RCCL path on the hardware at hand)
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
                raise RuntimeError('device_exchange=True needs an initialised "nccl" (RCCL) process group')
            return self._controller_step_device(global_actions, root, auto_reset)
        if not getattr(self, '_warned_host_exchange', False):
            # (until round 5 an initialised "nccl" group switched controller_step to device tensors by itself; now it is opted into)
            self._warned_host_exchange = True
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
                    import warnings
                    warnings.warn('BatchedRunEnv.controller_step: the process group is "nccl" (RCCL) but device_exchange=False -- the '
                                  'exchange is staged through host arrays and numpy comes back; pass device_exchange=True to '
                                  'BatchedRunEnv for the device-resident exchange (torch CUDA tensors back)')
            except ImportError:
                pass
        acts = self.scatter_from_root(global_actions, root=root)
        self.engine.step(acts, auto_reset=auto_reset)
        e = self.engine
        res = np.stack([e.read('DONE').astype(np.float64), e.read('FLAG').astype(np.float64), e.read('REWARD').sum(axis=1)], axis=1)
        full = self.gather_to_root(res, root=root)
        if full is None:
            return None
        return full[:, 0].astype(bool), full[:, 1].astype(np.int32), full[:, 2]

    def _engine_stream(self):
        """The engine's HIP stream as a torch stream (ppn_stream): lets torch-side work and engine-side work be ordered with
        stream waits (events) instead of host synchronisations."""
        import torch
        if getattr(self, '_ext_stream', None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.engine.stream_ptr(), device='cuda:%d' % self.device)
        return self._ext_stream

    def _controller_step_device(self, global_actions, root, auto_reset):
        """No host synchronisation inside: the engine's stream waits (event) for the scatter, the step kernel's epilogue has
        written (done, flag, reward sum) into ONE [batch x 3] row block (PPN_F_STEP_REPORT) that is copied on the engine's stream
        into the send buffer, the gather waits (event) for that copy.  The host only enqueues: step t + 1's scatter is queued while
        step t's gather is still in flight (two send / receive buffer sets alternate).  Until round 4: scatter -> host sync ->
        step -> three reads -> host sync -> pack -> gather, 5.8 M env-steps/s on one rank against 12.7 M without any exchange."""
        import torch
        import torch.distributed as dist
        dev = 'cuda:%d' % self.device
        sizes = [shard_range(self.global_batch, r, self.world_size) for r in range(self.world_size)]
        mx = max(b_ - a_ for a_, b_ in sizes)
        if getattr(self, '_xbuf', None) is None:
            self._xbuf = [dict(recv=torch.zeros((mx, self.action_length), dtype=torch.uint8, device=dev),
                               res=torch.zeros((mx, 3), dtype=torch.float64, device=dev),
                               out=None)
                          for _ in range(2)]
            self._xturn = 0
        buf = self._xbuf[self._xturn]
        self._xturn ^= 1
        if self.rank == root and buf['out'] is None:      # (whichever call first names this rank as root: ADVICE r05)
            buf['out'] = [torch.empty((mx, 3), dtype=torch.float64, device=dev) for _ in range(self.world_size)]
        cur = torch.cuda.current_stream(torch.device(dev))
        ext = self._engine_stream()
        parts = None
        if self.rank == root:
            ga = global_actions if torch.is_tensor(global_actions) else torch.from_numpy(np.ascontiguousarray(global_actions, dtype=np.uint8))
            ga = ga.to(dev, non_blocking=True)
            assert tuple(ga.shape) == (self.global_batch, self.action_length)
            parts = []
            for a_, b_ in sizes:
                p = ga[a_:b_]
                if b_ - a_ < mx:
                    p = torch.cat([p, torch.zeros((mx - (b_ - a_), self.action_length), dtype=torch.uint8, device=dev)])
                parts.append(p.contiguous())
        cur.wait_stream(ext)                 # (this buffer set's previous step -- two steps ago -- has been consumed by the engine)
        dist.scatter(buf['recv'], parts, src=root)
        ext.wait_stream(cur)                 # the engine's stream waits for the scatter: an event, no host synchronisation
        self.engine.step_device(buf['recv'].data_ptr(), auto_reset=auto_reset)      # the first `batch` rows are this shard's
        self.engine.read_into_device('STEP_REPORT', buf['res'].data_ptr(), 24 * self.batch)
        cur.wait_stream(ext)                 # the gather waits for the report copy
        dist.gather(buf['res'], buf['out'], dst=root)
        if self.rank != root:
            return None
        full = torch.cat([o[:b_ - a_] for o, (a_, b_) in zip(buf['out'], sizes)])
        return full[:, 0] != 0, full[:, 1].to(torch.int32), full[:, 2]

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
