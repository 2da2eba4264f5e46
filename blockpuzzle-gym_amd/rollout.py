"""Batched rollout workers with the reference's call signatures.

  RolloutStudent    mirrors gym_blocks/rollout.py:9-210
  PGRolloutStudent  mirrors gym_blocks/policy_gradient/rollout.py (keeps the
                    reward and returns the discounted return `G`, :217-231,255-258)

The reference steps `rollout_batch_size` Python envs one by one
(rollout.py:121-137); here the whole batch is one HIP launch per step, or -- when
no policy has to look at the observations between steps -- ONE launch for the
whole T-step rollout (`generate_rollouts_fused`) that also writes the
batch-major float32 episode arrays the reference returns (rollout.py:155-172).
"""
import ctypes as C
import pickle
import warnings
from collections import deque

import numpy as np
import torch

from . import _lib
from .envs import PhysicsError, _ptr, _stream

MAX_ROLLOUT_RESTARTS = 10      # the reference retries without bound (rollout.py:136-142)
NAN_CHECKS = ('per_step', 'per_rollout')


class _NaNCaught(Exception):
    """np.isnan(o_new).any() held after step t (rollout.py:139-142): raised inside a rollout body in per_step mode."""


def configure_dims(env):
    """config.py:159-183: one probe reset+step to read the dimensions."""
    env.reset()
    obs, _, _, info = env.step(env.action_space.sample()
                               if env.scalar_api else np.stack([env.action_space.sample()
                                                                for _ in range(env.num_envs)]))
    last = (lambda x: x.shape[-1])
    dims = {'o': last(obs['observation']), 'u': env.action_space.shape[0], 'g': last(obs['desired_goal'])}
    for key, value in info.items():
        value = np.array(value.cpu() if torch.is_tensor(value) else value)
        if value.ndim == 0 or not env.scalar_api:
            dims['info_{}'.format(key)] = 1
        else:
            dims['info_{}'.format(key)] = value.shape[0]
    return dims


def time_major_empty(T, B, dtype, device):
    """A (T, B) device tensor for a time-major stream (masks, rewards, flags) whose rows are
    bpg_row_stride(B) elements apart: rows a power of two apart land on the same HBM channels
    (DESIGN.md section 2).  A view of a (T, ld) allocation; stride(1) == 1."""
    ld = int(_lib.lib().bpg_row_stride(int(B)))
    return torch.empty((T, ld), dtype=dtype, device=device)[:, :B]


def discounted_return(reward_tb, gamma, out=None):
    """policy_gradient/rollout.py:255-258 on the GPU.  reward_tb: (T, B) float32
    (time-major) torch tensor; returns G (B, T) float64, bit-identical to the
    reference's in-order float64 accumulation."""
    assert reward_tb.dim() == 2 and reward_tb.dtype == torch.float32 and reward_tb.is_cuda
    if reward_tb.stride(1) != 1 or reward_tb.stride(0) < reward_tb.shape[1]:
        reward_tb = reward_tb.contiguous()
    T, B = reward_tb.shape                       # rows may be padded (time_major_empty): stride(0) >= B
    if out is None:
        out = torch.empty(B, T, dtype=torch.float64, device=reward_tb.device)
    _lib.check(_lib.lib().bpg_discounted_return_ld(_ptr(reward_tb), B, int(reward_tb.stride(0)) if T > 1 else B, T,
                                                   float(gamma), _ptr(out), reward_tb.device.index,
                                                   _stream(reward_tb.device)))
    return out


def discounted_return_scan(reward_tb, gamma, out=None):
    """The same return as an O(T) reverse scan G[t] = fma(gamma, G[t+1], r[t]): NOT the
    reference's summation order -- within 4 T 2^-53 sum_k gamma^k |r[t+k]| of it (include/bpgym.h)."""
    assert reward_tb.dim() == 2 and reward_tb.dtype == torch.float32 and reward_tb.is_cuda
    if reward_tb.stride(1) != 1 or reward_tb.stride(0) < reward_tb.shape[1]:
        reward_tb = reward_tb.contiguous()
    T, B = reward_tb.shape
    if out is None:
        out = torch.empty(B, T, dtype=torch.float64, device=reward_tb.device)
    _lib.check(_lib.lib().bpg_discounted_return_scan(_ptr(reward_tb), B, int(reward_tb.stride(0)) if T > 1 else B, T,
                                                     float(gamma), _ptr(out), reward_tb.device.index,
                                                     _stream(reward_tb.device)))
    return out


SHARD_ENVS = 262144           # envs per stream shard of the fused workers: the size whose 256-lane launches share CUs well
SHARD_FROM = 524288           # batches from this size on are split (below, one launch is as fast or faster)
MAX_SHARDS = 16


def auto_shards(num_envs):
    """How many stream shards `shards='auto'` splits a batch of `num_envs` envs into: ranges of about 262 144 envs from
    524 288 envs on (nothing to gain below)."""
    if num_envs < SHARD_FROM:
        return 1
    return max(1, min(MAX_SHARDS, num_envs // SHARD_ENVS))


class _ViewShards:
    """S views of one env batch (bpg_create_view: contiguous env ranges that share the batch's packed state and own their
    launch scratch), a stream each: the fused workers launch one shard per stream side by side -- the partition that spreads
    a job over GPUs (distributed.make_sharded), once more inside a GPU.  fork() / join() order the shard streams against the
    current stream (stream-level waits); nothing else synchronises."""

    def __init__(self, env, S):
        B, L = env.num_envs, _lib.lib()
        per = (B // S) // 4 * 4                   # shard boundaries on multiples of 4 envs: 16-byte aligned state slices
        if per < 4:
            raise ValueError('%d envs are too few for %d shards' % (B, S))
        self.env, self.views = env, []
        for i in range(S):
            lo = i * per
            n = (B - lo) if i == S - 1 else per
            h = C.c_void_p()
            _lib.check(L.bpg_create_view(C.byref(h), env._h, lo, n))
            self.views.append((h, torch.cuda.Stream(env.device), lo, n))

    def __len__(self):
        return len(self.views)

    def fork(self):
        cur = torch.cuda.current_stream(self.env.device)
        for _, st, _, _ in self.views:
            st.wait_stream(cur)

    def join(self):
        cur = torch.cuda.current_stream(self.env.device)
        for _, st, _, _ in self.views:
            cur.wait_stream(st)

    def close(self):
        views, self.views = self.views, []
        for h, _, _, _ in views:
            _lib.lib().bpg_destroy(h)

    __del__ = close


class _TotalSnapshot:
    """The worker's running total of successful envs as it stands behind everything enqueued so far, copied to pinned host
    memory asynchronously; value() waits for that copy only."""

    def __init__(self, total_dev, slot):
        slot.copy_(total_dev, non_blocking=True)
        self._slot, self._event, self._value = slot, torch.cuda.Event(), None
        self._event.record(torch.cuda.current_stream(total_dev.device))

    def value(self):
        if self._value is None:
            self._event.synchronize()
            self._value = int(self._slot[0])
            self._slot = self._event = None
        return self._value


class _LazyRate:
    """success_rate of one rollout = (total after it - total before it) / B, resolved when somebody looks at it."""

    def __init__(self, before, after, batch):
        self._before, self._after, self._batch, self._rate = before, after, batch, None

    def __float__(self):
        if self._rate is None:
            b = self._before if isinstance(self._before, int) else self._before.value()
            self._rate = float(self._after.value() - b) / self._batch
            self._before = self._after = None
        return self._rate


class _History(deque):
    """success_history: a deque of floats to every reader (rollout.py:37, :174-184 read it with np.mean); entries appended
    by an async_stats worker are resolved the first time they are read."""

    def _settle(self):
        for i in range(len(self)):
            x = deque.__getitem__(self, i)
            if isinstance(x, _LazyRate):
                deque.__setitem__(self, i, float(x))

    def __iter__(self):
        self._settle()
        return deque.__iter__(self)

    def __getitem__(self, i):
        x = deque.__getitem__(self, i)
        if isinstance(x, _LazyRate):
            x = float(x)
            deque.__setitem__(self, i, x)
        return x

    def __reversed__(self):
        self._settle()
        return deque.__reversed__(self)

    def __repr__(self):
        self._settle()
        return deque.__repr__(self)

    def __eq__(self, other):
        self._settle()
        return list(deque.__iter__(self)) == list(other)

    __hash__ = None


class RolloutStudent:

    def __init__(self, make_env, policy, dims, logger, T, rollout_batch_size=1,
                 exploit=False, use_target_net=False, compute_Q=False, noise_eps=0,
                 random_eps=0, history_len=100, render=False, output=None, async_stats=False, nan_check='per_step',
                 **kwargs):
        """Same arguments as the reference (rollout.py:11-14).  `make_env` may
        return a batched env directly (called as make_env(num_envs=B) when it
        accepts that) or a single env, which is re-created with B envs.
        Zero-copy note: the env writes observations / achieved goals straight into the episode a
        rollout returns, so until the next reset `env.observation` IS the last observation slot of
        the most recent episode (stepping the env by hand in between writes into that episode).
        async_stats=True (build-defined; for device-resident pipelines): generating a rollout never waits for the GPU --
        the success count of a rollout (rollout.py:163-167) is copied to pinned host memory behind the rollout's
        launches and `success_history` resolves the entry the first time it is read (np.mean, logs, indexing), as the
        reference's callers do once per epoch (train.py:67-76).  Default False: a float is appended before
        generate_rollouts returns, as in the reference.
        nan_check (only matters with a backend that writes observations, `writes_observation`; without one nothing can
        put a NaN there and nothing is checked): 'per_step' (default) = the reference's timing -- the device-side NaN
        flag is read after EVERY step (one host sync per step) and the rollout starts over at the first step whose
        observations hold a NaN (rollout.py:139-142); 'per_rollout' = build-defined fast mode -- the flag is read once,
        after all T steps (see generate_rollouts for what that changes).  generate_rollouts_graphed is always
        per_rollout: a replay cannot stop in the middle."""
        if nan_check not in NAN_CHECKS:
            raise ValueError('nan_check must be one of %s' % (NAN_CHECKS,))
        self.nan_check = nan_check
        self.make_env, self.policy, self.dims, self.logger, self.T = make_env, policy, dims, logger, T
        self.rollout_batch_size = rollout_batch_size
        self.exploit, self.use_target_net, self.compute_Q = exploit, use_target_net, compute_Q
        self.noise_eps, self.random_eps, self.history_len, self.render = noise_eps, random_eps, history_len, render
        self.kwargs = kwargs
        B = rollout_batch_size
        try:
            env = make_env(num_envs=B)
        except TypeError:
            env = make_env()
        if getattr(env, 'num_envs', None) != B or getattr(env, 'scalar_api', False):
            env = env.with_num_envs(B)
        self.env = env
        self.envs = [env]                    # the reference keeps a list (rollout.py:33)
        assert self.T > 0
        self.device = env.device
        self.output = output or ('torch' if getattr(policy, 'device_tensors', False) else 'numpy')
        self.info_keys = [key.replace('info_', '') for key in dims.keys() if key.startswith('info_')]
        self.success_history = _History(maxlen=history_len)
        self.async_stats = bool(async_stats)
        self.Q_history = deque(maxlen=history_len)
        self.n_episodes = 0
        self._watch_o, self._nan_flag = False, None
        # running total of successful envs over all rollouts of this worker: the LAST step launch of a rollout (or the
        # fused launch) adds the rollout's count, _finish reads the total once and takes the difference -- no launch of
        # its own for `successful = successes[-1]` (rollout.py:163-167), nothing to zero between rollouts (hipGraph
        # replays included)
        self._view_shards = {}               # S -> _ViewShards of self.env (created on first use)
        self._succ_total = torch.zeros(1, dtype=torch.int64, device=env.device)
        self._succ_seen = 0                  # the total before the next rollout: an int, or a _TotalSnapshot not read yet
        self._pinned = torch.empty(max(2 * history_len, 64), dtype=torch.int64).pin_memory() if self.async_stats else None
        self._pinned_next = 0
        f32 = dict(dtype=torch.float32, device=self.device)
        self.g = torch.empty((B, dims['g']), **f32)
        self.initial_o = torch.empty((B, dims['o']), **f32)
        self.initial_ag = torch.empty((B, dims['g']), **f32)
        self.reset_all_rollouts()
        self.clear_history()

    def _shards_for(self, shards):
        """shards='auto': auto_shards(B); an int: that many (1 / None = one launch on the current stream).  Returns None for 1."""
        S = auto_shards(self.rollout_batch_size) if shards == 'auto' else int(shards or 1)
        if S <= 1:
            return None
        if S not in self._view_shards:
            self._view_shards[S] = _ViewShards(self.env, S)
        return self._view_shards[S]

    # -- rollout.py:47-64 -------------------------------------------------
    def reset_all_rollouts(self, test=False, num_grey=None, observation_out=None, achieved_goal_out=None):
        if callable(num_grey):       # explicit draws that differ from reset to reset (a restarted rollout resets up to
            num_grey = num_grey()    # three times, fetch_env.py:647 draws each time): asked once per reset
        obs = self.env.reset(num_grey=num_grey, observation_out=observation_out, achieved_goal_out=achieved_goal_out)
        if test:                                     # maximum difficulty (rollout.py:53-55)
            obs = self.env.unwrapped.set_test(achieved_goal_out=achieved_goal_out)
        # the env hands out its observation / goal buffers and a fresh achieved_goal tensor: no copies here
        self.initial_o = obs['observation']
        self.initial_ag = obs['achieved_goal']
        self.g = obs['desired_goal']

    def reset_rollout(self, i, test=False):
        mask = torch.zeros(self.rollout_batch_size, dtype=torch.uint8, device=self.device)
        mask[i] = 1
        obs = self.env.reset(env_mask=mask)
        if test:
            obs = self.env.unwrapped.set_test(env_mask=mask)
        self.initial_o[i] = obs['observation'][i]
        self.initial_ag[i] = obs['achieved_goal'][i]
        self.g[i] = obs['desired_goal'][i]

    def increase_difficulty(self):
        # rollout.py:66-73
        max_level = False
        for env in self.envs:
            max_level = env.unwrapped.increase_difficulty()
        if not max_level:
            return self.envs[0].unwrapped.get_difficulty()
        return None

    # -- rollout.py:75-172 ------------------------------------------------
    def _policy_in(self, *ts):
        if self.output == 'torch':
            return ts
        return tuple(t.cpu().numpy() for t in ts)

    def _alloc_episode(self, time_major=False, o=True, g=True, zero_u=True):
        """Episode buffers.  time_major=True allocates (T[+1], B, .) and the caller returns
        transposed views -- exactly what the reference hands out: convert_episode_to_batch_major
        (util.py:118-128) is a swapaxes(0, 1) VIEW of the time-major arrays.
        zero_u=False: every action slot is written by the rollout (the step launch stores u, or the body assigns it), so
        the fill launch -- 52 MB at 65 536 envs, a node of every replayed graph -- is skipped."""
        B, T, d = self.rollout_batch_size, self.T, self.dims
        f32 = dict(dtype=torch.float32, device=self.device)
        shp = (lambda n, k: (n, B, k)) if time_major else (lambda n, k: (B, n, k))
        ep = dict(u=(torch.zeros if zero_u else torch.empty)(shp(T, d['u']), **f32), ag=torch.empty(shp(T + 1, d['g']), **f32),
                  info_is_success=torch.empty(shp(T, 1), **f32))
        if o:
            ep['o'] = torch.empty(shp(T + 1, d['o']), **f32)
        if g:
            ep['g'] = torch.empty(shp(T, d['g']), **f32)
        return ep

    def _resync_success_total(self):
        """After a rollout body whose result is thrown away (restart, graph warm-up): forget what it counted."""
        self._succ_seen = int(self._succ_total.item())

    def _finish(self, episode, Qs=(), counted=False):
        # rollout.py:163-167 `successful = successes[-1]; success_rate = np.mean(successful)`: the last
        # column of info_is_success is the has_succeeded latch of the state the rollout ended in, so
        # the mean is the HIP success count over B (exact in float64, as NumPy's mean of 0/1 values).
        # counted: the rollout's launches added that count to the worker's running total themselves
        if counted and self.async_stats:
            n = self._pinned.numel()
            if self._pinned_next >= n and self._pinned_next % n == 0:
                self.success_history._settle()           # the ring of pinned slots wraps: everything older is read by now
                if not isinstance(self._succ_seen, int):
                    self._succ_seen = self._succ_seen.value()
            snap = _TotalSnapshot(self._succ_total, self._pinned[self._pinned_next % n:self._pinned_next % n + 1])
            self._pinned_next += 1
            success_rate = _LazyRate(self._succ_seen, snap, self.rollout_batch_size)
            self._succ_seen = snap
        else:
            if counted:
                total = int(self._succ_total.item())
                before = self._succ_seen if isinstance(self._succ_seen, int) else self._succ_seen.value()
                successes, self._succ_seen = total - before, total
            else:
                successes = int(self.env.success_count().item())
            success_rate = float(successes) / self.rollout_batch_size
        self.success_history.append(success_rate)
        if self.compute_Q and len(Qs):
            # rollout.py:168-169 `self.Q_history.append(np.mean(Qs))`: NumPy's mean over the stacked per-step arrays, in
            # their own dtype -- the same expression on the same values (device tensors are brought to the host once)
            self.Q_history.append(np.mean([q.detach().cpu().numpy() if torch.is_tensor(q) else q for q in Qs]))
        self.n_episodes += self.rollout_batch_size
        if 'is_success' not in self.info_keys:       # only keys named in dims (rollout.py:35,160-161)
            episode = {k: v for k, v in episode.items() if k != 'info_is_success'}
        if self.output == 'numpy':
            return {k: v.cpu().numpy() for k, v in episode.items()}
        return episode

    def generate_rollouts(self, render=False, test=False, compute_Attention=False, num_grey=None, episode_return=False):
        """rollout.py:75-172 with the policy in the loop: one HIP step launch
        for the whole batch per time step.  Returns the batch-major episode.
        episode_return=True: also `episode_return` (B,) float32, every env's undiscounted return r[0] + ... + r[T-1],
        accumulated by the step launches themselves (bpg_step_args.d_return_acc) -- the per-step counterpart of
        generate_rollouts_fused(episode_return=True), same bits.
        num_grey: explicit (B,) draws for Variation's resets, or a callable returning them (asked once per reset).

        Restarts (rollout.py:136-142), pinned by tests/golden/restart_*.npz recorded from the reference:
          * a backend failure (PhysicsError, the stand-in for MujocoException, :136-137) starts the rollout over -- one
            reset, by the retry itself.  The reference has stepped envs 0 .. i-1 when env i raises; a backend that says
            which env failed (PhysicsError.failed_env + the step's contacts) gets exactly that, otherwise no env takes
            the failed step;
          * a NaN in the observations a backend wrote (:139-142) is detected on the device; the attempt is thrown
            away, everything is reset (:141) and generated again (:142, which resets once more: Variation draws
            num_grey three times in all).
        nan_check='per_step' (default) reads the flag after every step, so the discarded attempt ends with the NaN
        step, as in the reference -- with quirk Q1 the base variants keep what the discarded steps touched
        (fetch_env.py:247-255), so WHERE the attempt ends shows in the retried episode's achieved_goal.
        nan_check='per_rollout' is NOT the reference's timing: the flag is read once after all T steps, i.e. the
        discarded attempt runs to the end (the backend is called T instead of t+1 times and, for the base variants,
        steps t+1 .. T-1 leave their touches in the state the retry starts from).  It saves T-1 host syncs per rollout
        with such a backend and is the only mode a graph replay can offer (generate_rollouts_graphed)."""
        for attempt in range(MAX_ROLLOUT_RESTARTS + 1):
            try:
                ep, Qs = self._rollout_body(test, num_grey, compute_Attention, episode_return)
            except PhysicsError as e:
                self._log('warning', 'physics backend failed during rollout generation (%s). Trying again...' % e)
                self._step_envs_before_failure(e)
                self._resync_success_total()
                compute_Attention = False        # `return self.generate_rollouts(render, test)`, rollout.py:137
                continue
            except _NaNCaught:
                caught = True
            else:
                caught = self._nan_caught()
            if caught:
                self._log('warning', 'NaN caught during rollout generation. Trying again...')
                self._resync_success_total()
                self.reset_all_rollouts(test, num_grey=num_grey)
                compute_Attention = False        # rollout.py:142
                continue
            return self._finish(ep, Qs, counted=True)
        raise RuntimeError('rollout generation failed %d times in a row' % (MAX_ROLLOUT_RESTARTS + 1))

    def _step_envs_before_failure(self, e):
        """The reference's per-env loop has stepped envs 0 .. i-1 when env i raises (rollout.py:121-137)."""
        i = getattr(e, 'failed_env', None)
        if i is None or int(i) <= 0:
            return
        if e.masks is None and e.contacts is None:
            return
        self.env.step_prefix(int(i), getattr(self.env, 'last_action', None), contact_masks=e.masks, contacts=e.contacts)

    def _log(self, level, msg):
        fn = getattr(self.logger, level, None) or getattr(self.logger, 'info', None)
        if fn is not None:
            fn(msg)
        else:
            warnings.warn(msg)

    def _nan_caught(self):
        """One read of the device-side flag (per rollout in per_rollout mode: the step loop never synchronises)."""
        flag = getattr(self, '_nan_flag', None)
        return flag is not None and self._watch_o and bool(flag.item())

    def _check_step_nan(self, t):
        """per_step mode: np.isnan(o_new).any() right after step t, as the reference checks it (rollout.py:139)."""
        if self._watch_o and self.nan_check == 'per_step' and bool(self._nan_flag.item()):
            raise _NaNCaught(t)

    def _reset_into_episode(self, test, num_grey, ag_out=None):
        """reset_all_rollouts with the episode's first observation as the reset's target, and where
        the T+1 observations of the episode live.  A backend that rewrites the whole observation every
        step (`writes_observation`) gets time slot t+1 as its target during step t (env.observation
        points at it), and slot 0 is what the reset (and the backend's on_reset) wrote; otherwise
        nothing changes the observation between resets and the episode's `o` is that one reset
        observation, expanded (stride 0) -- rollout.py:144 appends T+1 copies of an array nobody wrote
        to.  Either way the reset launch writes into episode-owned memory: no copies."""
        B, T = self.rollout_batch_size, self.T
        self._watch_o = bool(getattr(self.env.contact_source, 'writes_observation', False))
        f32 = dict(dtype=torch.float32, device=self.device)
        if self._watch_o:
            if getattr(self, '_nan_flag', None) is None:
                self._nan_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._nan_flag.zero_()
            o = torch.empty((T + 1, B, self.dims['o']), **f32)
            self.env.observation_nan_flag = None       # the reference checks o_new of the steps only
            self.reset_all_rollouts(test, num_grey=num_grey, observation_out=o[0], achieved_goal_out=ag_out)
            self.env.observation_nan_flag = self._nan_flag
            return o
        self.env.observation_nan_flag = None
        o0 = torch.empty((B, self.dims['o']), **f32)
        self.reset_all_rollouts(test, num_grey=num_grey, observation_out=o0, achieved_goal_out=ag_out)
        return o0[None].expand(T + 1, B, self.dims['o'])

    def _step_into(self, u, ep, t, reward_out=None, u_store=None, return_acc=None):
        """One env step that writes obs['achieved_goal'] / info['is_success'] (and, through the
        backend, obs['observation']; optionally the step reward) straight into time slot t+1 / t
        of the episode.  The last step of the rollout also adds the success count to the worker's total."""
        env = self.env
        cnt = self._succ_total if t == self.T - 1 else None
        if self._watch_o:
            slot = env.observation = ep['o'][t + 1]          # the backend's write lands in the episode
            env._observation_checked = False
            r, _, _ = env._step_core(u, ag_out=ep['ag'][t + 1], succ_out=ep['info_is_success'][t],
                                     reward_out=reward_out, u_out=u_store, count_out=cnt, want_reward=False, want_flags=False,
                                     return_acc=return_acc)
            if not env._observation_checked:                 # written by other means than assemble_observation
                _lib.check(_lib.lib().bpg_nan_flag_f32(_ptr(slot), slot.shape[0], slot.shape[1], int(slot.stride(0)),
                                                       _ptr(self._nan_flag), self.device.index, _stream(self.device)))
        else:
            r, _, _ = env._step_core(u, ag_out=ep['ag'][t + 1], succ_out=ep['info_is_success'][t],
                                     reward_out=reward_out, u_out=u_store, count_out=cnt, want_reward=False, want_flags=False,
                                     return_acc=return_acc)
        return r

    def _rollout_body(self, test, num_grey, compute_Attention=False, episode_return=False):
        env, B, T = self.env, self.rollout_batch_size, self.T
        # -0.0f + r[0] + r[1] + ...: the sum that starts from r[0] (-0.0f when every step succeeded)
        acc = torch.full((B,), -0.0, dtype=torch.float32, device=self.device) if episode_return else None
        # time-major slabs: every per-step write below is one contiguous block; o / ag of step t
        # ARE the episode's time slot t (rollout.py:144-153 appends copies), and the env writes
        # the next achieved_goal straight into slot t+1
        ep = self._alloc_episode(time_major=True, g=False, o=False, zero_u=False)
        ep['o'] = self._reset_into_episode(test, num_grey, ag_out=ep['ag'][0])   # slot 0 = initial_ag (rollout.py:82-83)
        Qs = []
        # Without a backend that writes observations the step launch gets the ADDRESSES of the episode's time slots (one
        # bpg_step_args per rollout, integer arithmetic per step) instead of a tensor view per slot and step: the eager
        # loop is bound by the host (17 us of Python per step at 65 536 envs before, about half of it view objects)
        fast = not self._watch_o
        if fast:
            sa = _lib.StepArgs(ld_ag=self.dims['g'], d_return_acc=None if acc is None else acc.data_ptr())
            ag0, succ0, u0 = ep['ag'].data_ptr(), ep['info_is_success'].data_ptr(), ep['u'].data_ptr()
            ag_sz, succ_sz, u_sz = 4 * B * self.dims['g'], 4 * B, 4 * B * self.dims['u']
            o_const = ep['o'][0]                      # nothing rewrites the observation between resets: one view
            assert ep['ag'].is_contiguous() and ep['u'].is_contiguous() and self.dims['u'] == 4
        for t in range(T):
            o, ag = (o_const if fast else ep['o'][t]), ep['ag'][t]
            policy_output = self.policy.get_actions(
                *self._policy_in(o, ag, self.g),
                compute_Q=self.compute_Q,
                noise_eps=self.noise_eps if not self.exploit else 0.,
                random_eps=self.random_eps if not self.exploit else 0.,
                use_target_net=self.use_target_net)
            if self.compute_Q:
                u, Q = policy_output[0], policy_output[1]
                Qs.append(Q)
            else:
                u = policy_output[0] if (compute_Attention and isinstance(policy_output, tuple)) else policy_output
            if not (torch.is_tensor(u) and u.dtype == torch.float32 and u.device == self.device and u.dim() == 2
                    and u.shape[0] == B and u.is_contiguous()):
                u = torch.as_tensor(u, dtype=torch.float32).to(self.device).reshape(B, -1).contiguous()
            if u.shape[1] != self.dims['u']:
                raise ValueError('the policy returned actions of width %d, dims[\'u\'] = %d' % (u.shape[1], self.dims['u']))
            # reward ignored: recomputed for HER (rollout.py:123-125); the step launch also stores the
            # actions into the episode's slot (acts.append(u.copy()), rollout.py:147)
            if fast:
                sa.d_ag, sa.d_success_f32, sa.d_u_out = ag0 + (t + 1) * ag_sz, succ0 + t * succ_sz, u0 + t * u_sz
                sa.d_success_count = self._succ_total.data_ptr() if t == T - 1 else None
                env._step_with_args(u, sa)
            else:
                self._step_into(u, ep, t, u_store=ep['u'][t], return_acc=acc)
                self._check_step_nan(t)
        if self._watch_o:
            self.initial_o = ep['o'][T]                            # rollout.py:153 `self.initial_o[:] = o`
        ep = {k: v.transpose(0, 1) for k, v in ep.items()}         # batch-major views
        if acc is not None:
            ep['episode_return'] = acc
        # the goal is constant over the episode (rollout.py:151 appends the same array T times)
        ep['g'] = self.g[:, None, :].expand(B, T, self.dims['g'])
        return ep, Qs

    def _graph_body(self, test, num_grey):
        """What a replay runs: the rollout body, returning (episode dict, Qs)."""
        return self._rollout_body(test, num_grey, episode_return=getattr(self, '_graph_episode_return', False))

    def generate_rollouts_graphed(self, test=False, num_grey=None, _key=(), episode_return=False):
        """generate_rollouts for DEVICE-RESIDENT policies with the whole T-step loop (reset,
        policy, step kernels, episode writes) captured once into a hipGraph and replayed: below
        ~64 K envs the per-step path is bound by host launches (~50 us per step), a replay is
        one launch.  Requirements: `policy.device_tensors`, a capturable policy / contact
        source (no host syncs, no host->device copies inside), output='torch'.  The returned
        tensors are the graph's static buffers: they are overwritten by the next replay.
        Launch arguments are baked at capture; what varies between replays lives on the device:
        explicit `num_grey` draws are copied into a static tensor before each replay, and with
        num_grey=None Variation's draws come from the env's counter-based generator, whose
        stream index is a device word that every (replayed) reset advances."""
        assert self.output == 'torch' and getattr(self.policy, 'device_tensors', False)
        B = self.rollout_batch_size
        explicit = num_grey is not None
        self._graph_episode_return = bool(episode_return)
        _key = tuple(_key) + (bool(episode_return),)
        if (getattr(self, '_graph', None) is None or self._graph_test != bool(test)
                or self._graph_explicit != explicit or self._graph_key != _key):
            self._static_ng = torch.zeros(B, dtype=torch.int32, device=self.device)
            if explicit:
                self._static_ng.copy_(torch.as_tensor(num_grey).to(self.device, torch.int32))
            ng = self._static_ng if explicit else None       # None: the env's counter-based draw, whose
            side = torch.cuda.Stream(self.device)            # device-side stream index advances per replay
            side.wait_stream(torch.cuda.current_stream(self.device))
            snapshot = self.env.state_dict()                   # the warm-up pass must not count
            mode, self.nan_check = self.nan_check, 'per_rollout'   # a replay cannot stop in the middle: one read after it
            try:
                with torch.cuda.stream(side):                      # warm-up outside the capture
                    self._graph_body(test, ng)
                    self.env.load_state_dict(snapshot)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                self._resync_success_total()                       # the warm-up pass counted: not a rollout
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._graph_ep, self._graph_Qs = self._graph_body(test, ng)
            finally:
                self.nan_check = mode
            self._graph, self._graph_test, self._graph_explicit, self._graph_key = graph, bool(test), explicit, _key
        if explicit:
            self._static_ng.copy_(torch.as_tensor(num_grey).to(self.device, torch.int32))
        self._graph.replay()
        if self._nan_caught():
            raise FloatingPointError('NaN in the observations of a graph-replayed rollout (the replay cannot be '
                                     'restarted from the host: use generate_rollouts)')
        return self._finish(dict(self._graph_ep), self._graph_Qs, counted=True)

    def generate_rollouts_fused(self, test=False, masks=None, num_grey=None, actions=None,
                                want_reward=False, materialize_constants=False, gamma=None, episode_return=False,
                                episode_fail_count=False, shards=1):
        """Whole rollout in ONE kernel launch (no policy between steps): contacts
        come from `masks` (T, B) int32/uint32 device tensor, or the synthetic
        source when None.  Same episode dict as generate_rollouts; `o` is the
        env's physics-less observation repeated, `u` = `actions` or zeros.
        The arrays that are constant over the episode (`g`, and `o` without a physics
        backend) are returned as stride-0 expanded views unless materialize_constants=True,
        which makes the kernel write contiguous float32 copies of them as well.
        gamma: also return `G` (B, T) float64, the policy-gradient return of the rollout
        (policy_gradient/rollout.py:255-258, bit-identical), computed by the same call from the
        step-ok bits of the launch -- no reward stream is written or read for it.
        episode_return: also return `episode_return` (B,) float32, every env's undiscounted return
        r[0] + ... + r[T-1] (bpg_rollout_args.d_episode_return: the per-env array a multi-GPU job
        all-gathers, distributed.all_gather_per_env), from a counter of the same launch.
        episode_fail_count: also `episode_fail_count` (B,) uint8, the same quantity as one byte per env (the number of steps
        whose success condition did not hold; episode_return == -episode_fail_count bit for bit, T <= 255): the form a
        multi-GPU job puts on the wire (a quarter of the bytes; distributed.episode_return_from_fail_count expands it).
        shards: S > 1 (or 'auto') runs the batch as S contiguous env ranges on S streams (views of the env, one launch each,
        joined before this returns): every output is the same array, every bit the one launch's.  Default 1: measured, it
        does not pay here -- see evaluate_rollouts_fused."""
        env, B, T = self.env, self.rollout_batch_size, self.T
        o0 = torch.empty((B, self.dims['o']), dtype=torch.float32, device=self.device)
        self.reset_all_rollouts(test, num_grey=num_grey, observation_out=o0)      # the episode owns the reset observation
        ep = self._alloc_episode(o=materialize_constants, g=materialize_constants)
        if materialize_constants:
            ep['o'].copy_(o0[:, None, :].expand(B, T + 1, self.dims['o']))
        else:
            ep['o'] = o0[:, None, :].expand(B, T + 1, self.dims['o'])
            ep['g'] = self.g[:, None, :].expand(B, T, self.dims['g'])
        if actions is not None:
            ep['u'].copy_(torch.as_tensor(actions).to(self.device))               # stored as given (rollout.py:147)
        reward = time_major_empty(T, B, torch.float32, self.device) if want_reward else None
        G = torch.empty((B, T), dtype=torch.float64, device=self.device) if gamma is not None else None
        if G is not None:       # the handle's step-ok words: allocated once, outside any stream capture (idempotent)
            _lib.check(_lib.lib().bpg_reserve_return_scratch(env._h))
        ret = torch.empty(B, dtype=torch.float32, device=self.device) if episode_return else None
        nfail = torch.empty(B, dtype=torch.uint8, device=self.device) if episode_fail_count else None
        args = _lib.RolloutArgs(T=T, d_masks=None if masks is None else masks.data_ptr(),
                                d_G=None if G is None else G.data_ptr(), gamma=0.0 if gamma is None else float(gamma),
                                d_episode_return=None if ret is None else ret.data_ptr(),
                                d_episode_fail_count=None if nfail is None else nfail.data_ptr(),
                                d_reward_tb=None if reward is None else reward.data_ptr(),
                                d_ep_ag=ep['ag'].data_ptr(),
                                d_ep_g=ep['g'].data_ptr() if materialize_constants else None,
                                d_ep_success=ep['info_is_success'].data_ptr(),
                                d_success_count=self._succ_total.data_ptr(),
                                ld_out=0 if reward is None else int(reward.stride(0)))
        if masks is not None:            # rows may be padded (time_major_empty): any row stride >= B
            assert masks.shape == (T, B) and masks.stride(1) == 1 and masks.element_size() == 4 \
                and (T == 1 or masks.stride(0) >= B)
            args.ld_masks = int(masks.stride(0)) if T > 1 else B
        vs = self._shards_for(shards)
        if vs is None:
            _lib.check(_lib.lib().bpg_rollout_v2(env._h, C.byref(args), _stream(self.device)))
        else:
            # env range [lo, lo + n) of every array: time-major streams are column ranges (same row stride), batch-major
            # arrays contiguous row ranges
            dg = self.dims['g']
            per_env_bytes = dict(d_masks=4, d_reward_tb=4, d_ep_ag=4 * (T + 1) * dg, d_ep_g=4 * T * dg, d_ep_success=4 * T,
                                 d_G=8 * T, d_episode_return=4, d_episode_fail_count=1)
            if args.ld_out == 0 and reward is not None:
                args.ld_out = B
            if masks is not None and args.ld_masks == 0:
                args.ld_masks = B
            vs.fork()
            for h, st, lo, n in vs.views:
                a = _lib.RolloutArgs.from_buffer_copy(args)
                for name, nbytes in per_env_bytes.items():
                    base = getattr(args, name)
                    if base:
                        setattr(a, name, base + lo * nbytes)
                if G is not None:
                    _lib.check(_lib.lib().bpg_reserve_return_scratch(h))
                _lib.check(_lib.lib().bpg_rollout_v2(h, C.byref(a), C.c_void_p(st.cuda_stream)))
            vs.join()
        out = self._finish(ep, counted=True)
        if want_reward:
            out['reward_tb'] = reward if self.output == 'torch' else reward.cpu().numpy()
        if G is not None:
            out['G'] = G if self.output == 'torch' else G.cpu().numpy()
        if ret is not None:
            out['episode_return'] = ret if self.output == 'torch' else ret.cpu().numpy()
        if nfail is not None:
            out['episode_fail_count'] = nfail if self.output == 'torch' else nfail.cpu().numpy()
        return out

    def shard_ranges(self, shards='auto'):
        """[(first_env, n_envs)] of the stream shards the fused workers split this worker's batch into (`shards` as in
        evaluate_rollouts_fused): contiguous, boundaries on multiples of 4 envs, the last shard takes the remainder."""
        vs = self._shards_for(shards)
        return [(0, self.rollout_batch_size)] if vs is None else [(lo, n) for _, _, lo, n in vs.views]

    def shard_streams(self, rows, dtype, shards='auto'):
        """One time-major (rows, n_shard) device array PER SHARD -- arrays of the shards' own, each with its own
        bpg_row_stride -- for the masks / reward_tb / flags_tb of evaluate_rollouts_fused: a physics backend that fills the
        contact masks shard by shard, or a consumer that reads the streams shard by shard, loses nothing by it, and launches on
        such arrays stream at 0.78 of the HBM peak instead of 0.67-0.72 when many are in flight per join."""
        return [time_major_empty(rows, n, dtype, self.device) for _, n in self.shard_ranges(shards)]

    def evaluate_rollouts_fused(self, masks, n_rollouts=None, test=False, num_grey=None, reward_tb=None,
                                flags_tb=None, episode_return=None, episode_fail_count=None, shards=1):
        """The evaluator's test loop -- `for _ in range(n_test_rollouts): evaluator.generate_rollouts(...)`
        (train.py:63-64, play.py:59) -- with the contacts of all K rollouts read from one stream, in ONE
        launch: rollout r is reset + T steps on rows r*T .. r*T+T-1 of `masks` ((K*T, B) int32 device
        tensor, rows may be padded) and appends its success rate to success_history exactly as K calls of
        generate_rollouts_fused would (rollout.py:163-170); no episode dict is kept, as the test loop
        drops it.  num_grey: (K, B) explicit draws for BlocksTouchVariation, or None for the env's
        generator (its next K draws).  reward_tb / flags_tb: optional (K*T, B) float32 / uint8 streams
        the launch fills (step reward; bit 0 is_success, bit 1 done); episode_return: optional (K, B) float32
        the launch fills with every env's undiscounted return of each rollout; episode_fail_count: optional (K, B) uint8, the
        same as one byte per env (T <= 255).  Returns the K success rates.

        STREAM SHARDS.  The batch can run as S contiguous env ranges (views of the env, bpg_create_view), each launching its
        rollouts back to back on a stream of its own, forked from and joined into the current stream once per call: the
        launches of different shards overlap, so the ramp-up and drain of one are covered by the others.  Every result bit is
        the one launch's.  Two forms:
          * `masks` (and `reward_tb` / `flags_tb` when given) as a LIST of per-shard arrays -- arrays of the shards' own, one per
            range of shard_ranges(len(masks)); shard_streams() allocates them;
          * `shards=S` (or 'auto') with ONE (K*T, B) array per stream: the shards work on column ranges of the caller's arrays.
        What was measured (bench.py roofline_sharded, profiles/r05_shard_probe.log; 1 M envs as 4 shards): launched back to
        back and joined rarely, shards with arrays of their own stream at 0.78 of the HBM peak on every box against 0.67-0.72
        for the one launch; on column ranges 0.67-0.77 depending on the box.  A CALL of this method has a fixed cost of about
        250 us (validation, fork / join, the read-back of the K counts) that the one launch pays too in part: at the
        evaluator's K = 10-20 rollouts per call the two forms come out even (89.7 against 90.2 us per rollout at K = 20), the
        sharded form wins from about 50 rollouts per call.  Hence the default shards=1 -- one launch on the current stream --
        and the per-shard-arrays form for pipelines that keep many rollouts in flight per join."""
        env, B, T = self.env, self.rollout_batch_size, self.T
        per_shard = isinstance(masks, (list, tuple))
        if per_shard:
            shards = len(masks)
        vs = self._shards_for(shards)
        ranges = [(0, B)] if vs is None else [(lo, n) for _, _, lo, n in vs.views]
        if per_shard and len(ranges) != len(masks):
            raise ValueError('%d per-shard mask arrays for %d shards' % (len(masks), len(ranges)))

        def per_range(x, dt, what):
            """x: None, one (rows, B) array (every shard gets its column range) or a list of per-shard (rows, n) arrays ->
            [(data pointer, row stride in elements)] per shard."""
            if x is None:
                return [(None, 0)] * len(ranges)
            xs = list(x) if isinstance(x, (list, tuple)) else None
            def ok(t, n):
                return torch.is_tensor(t) and t.dim() == 2 and t.shape[1] == n and t.stride(1) == 1 and t.is_cuda and t.dtype in dt
            if xs is None:
                if not ok(x, B):
                    raise ValueError('%s must be a (rows, %d) %s device array with unit inner stride' % (what, B, dt[0]))
                ldx = int(x.stride(0)) if x.shape[0] > 1 else B
                return [(x.data_ptr() + lo * x.element_size(), ldx) for lo, n in ranges]
            if len(xs) != len(ranges) or not all(ok(t, n) for t, (lo, n) in zip(xs, ranges)):
                raise ValueError('%s: per-shard arrays must be (rows, n) %s device arrays for the ranges %s (shard_ranges)'
                                 % (what, dt[0], ranges))
            return [(t.data_ptr(), int(t.stride(0)) if t.shape[0] > 1 else n) for t, (lo, n) in zip(xs, ranges)]

        first = masks[0] if per_shard else masks
        assert first.dim() == 2 and first.shape[0] % T == 0 and first.element_size() == 4
        rows = first.shape[0]
        K = rows // T
        if n_rollouts is not None and int(n_rollouts) != K:
            raise ValueError('masks hold %d rollouts of T = %d rows, n_rollouts = %d' % (K, T, n_rollouts))
        m_ptrs = per_range(masks, (torch.int32, torch.uint32), 'masks')
        r_ptrs = per_range(reward_tb, (torch.float32,), 'reward_tb')
        f_ptrs = per_range(flags_tb, (torch.uint8,), 'flags_tb')
        for x in (masks, reward_tb, flags_tb):
            for t in (x if isinstance(x, (list, tuple)) else ([] if x is None else [x])):
                assert t.shape[0] == rows
        for (rp, rl), (fp, fl) in zip(r_ptrs, f_ptrs):
            if rp and fp and rl != fl:
                raise ValueError('reward_tb and flags_tb must share one row stride')
        ng = None
        if num_grey is not None:
            ng = torch.as_tensor(num_grey).to(self.device, torch.int32).contiguous()
            assert ng.shape == (K, B)
        if test and not env.info.set_test_supported:
            raise NotImplementedError()             # fetch_env.py:100-101, as each of the K calls would
        # (set_test only re-places the cubes, rollout.py:53-55: with the contacts read from the stream the
        # placement has no reader, so nothing is placed here)
        counts = torch.zeros(K, dtype=torch.int64, device=self.device)
        for x, dt in ((episode_return, torch.float32), (episode_fail_count, torch.uint8)):
            if x is not None:
                assert x.shape == (K, B) and x.dtype == dt and x.is_contiguous() and x.is_cuda

        def args_of(i, lo, k0=0, nk=K):
            """Rollouts k0 .. k0+nk-1 of shard i (first env lo) as one launch: its rows of its streams, its column range of the
            (K, B) per-env arrays (row stride B); the K success counts are shared words (atomic adds)."""
            off = lambda p, nbytes, ld: (p + nbytes * k0 * T * ld) if p else None        # noqa: E731  (row k0*T of a (K*T, ld) stream)
            ldo = r_ptrs[i][1] or f_ptrs[i][1]
            return _lib.RolloutArgs(
                T=T, reset_first=1, rollouts=nk, d_masks=off(m_ptrs[i][0], 4, m_ptrs[i][1]), ld_masks=m_ptrs[i][1],
                d_reward_tb=off(r_ptrs[i][0], 4, ldo), d_flags_tb=off(f_ptrs[i][0], 1, ldo), ld_out=ldo,
                d_episode_return=None if episode_return is None else episode_return.data_ptr() + 4 * (lo + k0 * B),
                d_episode_fail_count=None if episode_fail_count is None else episode_fail_count.data_ptr() + (lo + k0 * B),
                d_num_grey=None if ng is None else ng.data_ptr() + 4 * (lo + k0 * B), ld_rollouts=B,
                d_success_count=counts.data_ptr() + 8 * k0)
        if vs is None:
            a = args_of(0, 0)
            _lib.check(_lib.lib().bpg_rollout_v2(env._h, C.byref(a), _stream(self.device)))
        else:
            # One launch PER ROLLOUT per shard, back to back on the shard's stream: what overlaps between the shards is the ends of
            # their launches, and a launch that runs all K rollouts of a shard has none inside it (measured: K = 20 in one launch
            # per shard is slower than the one launch of the whole batch; K launches per shard are not).  Generator-drawn num_grey
            # needs the one launch: a view's resets do not advance the generator's stream index.
            per_rollout = not (env.info.is_variation and ng is None)
            # the K x S argument structs depend on the arrays' addresses only: an evaluator that reuses its buffers from call
            # to call (they all do) pays for building them once -- at 20 us of Python per struct the enqueue of 80 launches
            # would otherwise take as long as the launches themselves
            key = (K, per_rollout, len(vs), tuple(m_ptrs), tuple(r_ptrs), tuple(f_ptrs),
                   *(None if x is None else x.data_ptr() for x in (ng, episode_return, episode_fail_count)))
            cache = self.__dict__.setdefault('_eval_launches', {})
            plan = cache.get(key)
            if plan is None:
                if len(cache) >= 8:
                    cache.clear()
                rolls = [(k, 1) for k in range(K)] if per_rollout else [(0, K)]
                S = len(vs)
                arr = (_lib.RolloutArgs * (len(rolls) * S))()                      # launch-major, as bpg_rollout_train reads it
                for l, (k0, nk) in enumerate(rolls):
                    for i, (h, st, lo, n) in enumerate(vs.views):
                        arr[l * S + i] = args_of(i, lo, k0, nk)
                plan = cache[key] = (arr, len(rolls), (C.c_void_p * S)(*[h.value for h, _, _, _ in vs.views]),
                                     (C.c_void_p * S)(*[st.cuda_stream for _, st, _, _ in vs.views]))
            arr, n_launches, handles, streams = plan
            count_ptr, S = counts.data_ptr(), len(vs)
            for idx in range(n_launches * S):          # the one field that is new every call
                arr[idx].d_success_count = count_ptr + 8 * ((idx // S) if per_rollout else 0)
            # ONE call enqueues the fork, the n_launches x S launches and the join (bpg_rollout_train): the host cost of a launch is
            # that of the kernel launch itself, not a trip through the interpreter
            _lib.check(_lib.lib().bpg_rollout_train(handles, streams, S, arr, C.sizeof(_lib.RolloutArgs), n_launches,
                                                    _stream(self.device)))
            if env.info.is_variation and ng is None:     # the views drew from the generator: its stream index moves once
                _lib.check(_lib.lib().bpg_advance_generator(env._h, K, 0, _stream(self.device)))
        env._mark_static_observation_stale()    # the K in-launch resets drew new block counts: the observation rows follow when read
        rates = [float(c) / B for c in counts.tolist()]
        self.success_history.extend(rates)
        self.n_episodes += K * B
        return rates

    # -- rollout.py:174-210 -----------------------------------------------
    def clear_history(self):
        self.success_history.clear()
        self.Q_history.clear()

    def current_success_rate(self):
        return np.mean(self.success_history)

    def current_mean_Q(self):
        return np.mean(self.Q_history)

    def save_policy(self, path):
        with open(path, 'wb') as f:
            pickle.dump(self.policy, f)

    def logs(self, prefix='worker'):
        logs = [('success_rate', np.mean(self.success_history))]
        if self.compute_Q:
            logs += [('mean_Q', np.mean(self.Q_history))]
        logs += [('episode', self.n_episodes)]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + key, val) for key, val in logs]
        return logs

    def seed(self, seed):
        """rollout.py:206-210 seeds env idx with seed + 1000 * idx: the batched env keys env idx of the
        batch by exactly that number (on the global env index), see BatchedBlocksEnv.seed."""
        for env in self.envs:
            env.seed(seed)


class PGRolloutStudent(RolloutStudent):
    """policy_gradient/rollout.py: keeps the step reward and returns the
    discounted return G (B, T) float64 and sigma (B, T, dimu)."""

    def __init__(self, make_env, policy, expert, dims, logger, T, rollout_batch_size=1,
                 exploit=False, use_target_net=False, compute_Q=False, noise_eps=0,
                 random_eps=0, history_len=100, render=False, gamma=None,
                 beta_final=None, annealing_coeff=None, **kwargs):
        super().__init__(make_env, policy, dims, logger, T, rollout_batch_size=rollout_batch_size,
                         exploit=exploit, use_target_net=use_target_net, compute_Q=compute_Q,
                         noise_eps=noise_eps, random_eps=random_eps, history_len=history_len,
                         render=render, **kwargs)
        self.gamma, self.time = gamma, 0.0
        self.beta_final, self.annealing_coeff, self.expert = beta_final, annealing_coeff, expert
        if hasattr(policy, 'kwargs'):
            self.kwargs = policy.kwargs             # policy_gradient/rollout.py:53: trim reads ['info']['env_name'] from it
        self._trim_bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        if gamma is not None:   # the fused form keeps the step rewards as step-ok words in handle-owned scratch: allocated here,
            _lib.check(_lib.lib().bpg_reserve_return_scratch(self.env._h))   # not by a hipMalloc in the middle of the first rollout

    def save_policy_weights(self, path):
        self.policy.save_weights(path)              # policy_gradient/rollout.py:313-314 (train.py:112)

    def beta(self):
        return (1.0 - self.beta_final) * np.exp(-self.time / self.annealing_coeff) + self.beta_final

    def anneal(self):
        self.time += 1.0
        if self.expert is not None:                  # policy_gradient/rollout.py:292-293
            self._log('info', 'Beta = {}'.format(self.beta()))

    def _env_name(self):
        info = (self.kwargs or {}).get('info') if isinstance(self.kwargs, dict) else None
        if info and 'env_name' in info:             # what the reference reads (policy_gradient/rollout.py:117, :152)
            return info['env_name']
        return self.env.spec.id                     # workers built without the policy's kwargs: the env's own id

    def trim(self, o, g, ag, dimo, dimg, num_objs=4, _defer_check=False):
        """policy_gradient/rollout.py:105-171: the inputs of an expert trained on a smaller puzzle -- the corner of
        the goal matrices that belongs to the first `num_objs` objects, and of the observation either its first
        `dimo` entries or (Variation) the env features plus the base features of the GREEN and BLUE blocks.
        One HIP launch on device tensors (numpy inputs are moved to the worker's device and the results come
        back as numpy).  Dtypes as the reference's: g_, ag_ float32; o_ float32, Variation float64.  A
        Variation row without exactly num_objs - 2 coloured blocks raises ValueError as NumPy does at :166."""
        if o.shape[-1] == dimo:                     # "No need to trim": the same objects back (:107-108)
            return o, g, ag
        assert len(o.shape) != 1                    # the reference's 1-D branch is `assert(False)` (:114-115)
        as_numpy = not torch.is_tensor(o)
        dev = lambda x: torch.as_tensor(x).to(self.device, torch.float32).contiguous()   # noqa: E731
        o, g, ag = dev(o), dev(g), dev(ag)
        n, variation = o.shape[0], 'Variation' in self._env_name()
        assert int(min(num_objs, int(g.shape[1] ** 0.5))) ** 2 == dimg       # assert(len(g_[0]) == dimg), :146
        assert (dimo == 10 + 15 * (num_objs - 2)) if variation else (dimo <= o.shape[1])     # assert(o_.shape[1] == dimo), :171
        o_ = torch.empty((n, dimo), dtype=torch.float64 if variation else torch.float32, device=self.device)
        g_ = torch.empty((n, dimg), dtype=torch.float32, device=self.device)
        ag_ = torch.empty((n, dimg), dtype=torch.float32, device=self.device)
        if not _defer_check:
            self._trim_bad.zero_()
        _lib.check(_lib.lib().bpg_trim_expert_inputs(_ptr(o), _ptr(g), _ptr(ag), n, o.shape[1], g.shape[1], dimo, dimg,
                                                     int(num_objs), int(variation), _ptr(o_), _ptr(g_), _ptr(ag_),
                                                     _ptr(self._trim_bad), self.device.index, _stream(self.device)))
        if variation and not _defer_check and int(self._trim_bad.item()):     # _defer_check: the rollout reads the flag once, at its end
            raise ValueError('trim: an observation row does not hold exactly %d GREEN / BLUE blocks '
                             '(policy_gradient/rollout.py:166)' % (num_objs - 2))
        if as_numpy:
            return o_.cpu().numpy(), g_.cpu().numpy(), ag_.cpu().numpy()
        return o_, g_, ag_

    def generate_rollouts_fused(self, test=False, masks=None, num_grey=None, actions=None, want_reward=False,
                                materialize_constants=False, gamma=None, episode_return=False, episode_fail_count=False, shards=1):
        """The whole pg rollout in one call without a policy between the steps: the episode arrays and
        the return `G` for this worker's gamma (policy_gradient/rollout.py:255-258, :283-288)."""
        return super().generate_rollouts_fused(test=test, masks=masks, num_grey=num_grey, actions=actions,
                                               want_reward=want_reward, materialize_constants=materialize_constants,
                                               gamma=self.gamma if gamma is None else gamma, episode_return=episode_return,
                                               episode_fail_count=episode_fail_count, shards=shards)

    def generate_rollouts(self, render=False, test=False, exploit=False, num_grey=None, episode_return=False):
        """policy_gradient/rollout.py:174-290.  A backend failure is fatal here as in the reference
        (`assert(False)`, :237-241); NaN observations restart the rollout (:243-246) -- reset_all_rollouts, then
        `self.generate_rollouts(render, test)`: the retry runs with exploit at its default (False), whatever the
        caller passed, and so does this one.  nan_check: see RolloutStudent.generate_rollouts.
        episode_return=True: also `episode_return` (B,) float32 = r[0] + ... + r[T-1] (the reference's returns[0] at
        gamma = 1), accumulated by the step launches."""
        for attempt in range(MAX_ROLLOUT_RESTARTS + 1):
            try:
                ep = self._pg_body(test, exploit, num_grey, episode_return)
                caught = self._nan_caught()
            except _NaNCaught:
                caught = True
            if caught:
                self._log('info', 'NaN caught during rollout generation. Trying again...')
                self._resync_success_total()
                self.reset_all_rollouts(test, num_grey=num_grey)
                exploit = False                  # policy_gradient/rollout.py:246
                continue
            return self._finish(ep, counted=True)
        raise RuntimeError('rollout generation failed %d times in a row' % (MAX_ROLLOUT_RESTARTS + 1))

    def _graph_body(self, test, num_grey):
        return self._pg_body(test, self._graph_exploit, num_grey, getattr(self, '_graph_episode_return', False)), ()

    def generate_rollouts_graphed(self, test=False, exploit=False, num_grey=None, episode_return=False):
        """The policy-gradient rollout (reset, T x (policy, step, reward row), the return kernel) replayed from one
        hipGraph -- see RolloutStudent.generate_rollouts_graphed for the requirements.  Without an expert only: with
        one, every step flips a host-side coin (policy_gradient/rollout.py:196) and calls host code."""
        if self.expert is not None:
            raise NotImplementedError('a rollout with an expert decides per step on the host: use generate_rollouts')
        self._graph_exploit = bool(exploit)
        # The reference draws np.random.rand() once per step even when beta is 0 (policy_gradient/rollout.py:196), and so
        # does the eager path.  A replay runs no host code, and warm-up + capture run the body twice: put the global
        # NumPy stream back where it was and consume exactly T draws per rollout, so that whatever else reads
        # np.random after set_global_seeds (exploration noise, HER sampling) sees the stream an eager rollout leaves.
        state = np.random.get_state()
        try:
            return super().generate_rollouts_graphed(test=test, num_grey=num_grey, _key=('pg', bool(exploit)),
                                                     episode_return=episode_return)
        finally:
            np.random.set_state(state)
            np.random.rand(self.T)

    def _pg_body(self, test, exploit, num_grey, episode_return=False):
        env, B, T = self.env, self.rollout_batch_size, self.T
        acc = torch.full((B,), -0.0, dtype=torch.float32, device=self.device) if episode_return else None
        beta = self.beta() if self.expert is not None else 0
        ep = self._alloc_episode(time_major=True, g=False, o=False, zero_u=False)
        ep['o'] = self._reset_into_episode(test, num_grey, ag_out=ep['ag'][0])
        # "We can't report sigma accurately when we are using the expert": np.zeros((B, dimu)), float64 (:207-209)
        ep['sigma'] = torch.zeros((T, B, self.dims['u']), device=self.device,
                                  dtype=torch.float64 if self.expert is not None else torch.float32)
        reward = time_major_empty(T, B, torch.float32, self.device)
        defer = self.expert is not None and self.output == 'torch'
        if defer:
            self._trim_bad.zero_()
        fast = not self._watch_o                     # addresses of the time slots instead of views (see _rollout_body)
        if fast:
            sa = _lib.StepArgs(ld_ag=self.dims['g'], d_return_acc=None if acc is None else acc.data_ptr())
            ag0, succ0, rew0 = ep['ag'].data_ptr(), ep['info_is_success'].data_ptr(), reward.data_ptr()
            ag_sz, succ_sz, rew_sz = 4 * B * self.dims['g'], 4 * B, 4 * int(reward.stride(0))
            o_const = ep['o'][0]
            assert ep['ag'].is_contiguous()
        for t in range(T):
            o, ag = (o_const if fast else ep['o'][t]), ep['ag'][t]
            if np.random.rand() < beta:            # the expert is in charge (:196-200)
                # device-resident experts: no host read of the bad-row flag per step, one at the end of the rollout
                o_, g_, ag_ = self.trim(o, self.g, ag, self.expert.dimo, self.expert.dimg, _defer_check=defer)
                u, raw = self.expert.get_actions(*self._policy_in(o_, ag_, g_), compute_raw=True)
                sigma = None
            else:
                u, raw, sigma = self.policy.get_actions(*self._policy_in(o, ag, self.g), exploit=exploit)
            if self.expert is not None:
                sigma = None
            u = torch.as_tensor(u, dtype=torch.float32).to(self.device).reshape(B, -1)
            ep['u'][t] = torch.as_tensor(raw, dtype=torch.float32).to(self.device).reshape(B, -1)
            if sigma is not None:
                ep['sigma'][t] = torch.as_tensor(sigma, dtype=torch.float32).to(self.device).reshape(B, -1)
            if fast:
                sa.d_ag, sa.d_success_f32, sa.d_reward = ag0 + (t + 1) * ag_sz, succ0 + t * succ_sz, rew0 + t * rew_sz
                sa.d_success_count = self._succ_total.data_ptr() if t == T - 1 else None
                env._step_with_args(u.contiguous(), sa)
            else:
                self._step_into(u, ep, t, reward_out=reward[t], return_acc=acc)
                self._check_step_nan(t)
        if defer and int(self._trim_bad.item()):
            raise ValueError('trim: an observation row did not hold exactly 2 GREEN / BLUE blocks at some step of the '
                             'rollout (policy_gradient/rollout.py:166)')
        if self._watch_o:
            self.initial_o = ep['o'][T]
        ep = {k: v.transpose(0, 1) for k, v in ep.items()}
        ep['g'] = self.g[:, None, :].expand(B, T, self.dims['g'])
        ep['G'] = discounted_return(reward, self.gamma)
        if acc is not None:
            ep['episode_return'] = acc
        return ep
