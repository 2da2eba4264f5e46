"""Round 5 GPU tests: the reference's restart semantics replayed from fixtures recorded from the reference itself
(tests/golden/restart_*.npz, pg_restart_*.npz <- make_golden.restart_trace), the sharded default of the fused
workers, the u8 episode-return gather, the curriculum and collective entry points of the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from blockpuzzle_gym_amd import _lib
from util import GOLDEN, VARIANTS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dims_of(env_id):
    v = VARIANTS[env_id]
    return dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)


class _Log:
    def __init__(self):
        self.lines = []

    def warning(self, m):
        self.lines.append(m)

    info = warning


# --------------------------------------------------------------------------
# rollout.py:136-142 / policy_gradient/rollout.py:237-246, pinned by the reference
# --------------------------------------------------------------------------
class _FaultyBackend:
    """Replays one generate_rollouts call of a restart fixture: the contact lists of the discarded attempt (if any), the
    fault at its last step -- a NaN in env j's observation, or PhysicsError(failed_env=j) with the step's contacts of the
    envs before j -- then the T contact lists of the attempt the reference returned."""

    def __init__(self, B, dimo):
        self.B, self.dimo = B, dimo
        self.resets = 0

    def load(self, d, r, T):
        kind, j, k = (int(x) for x in d['faults'][r])
        self.kind, self.j, self.k, self.T = kind, j, k, T
        dc = d['r%d_discard_contacts' % r].copy()
        dc[dc < 0] = 0
        self.dcon, self.dncon = dc, d['r%d_discard_ncon' % r]
        c = d['r%d_contacts' % r].copy()
        c[c < 0] = 0
        self.con, self.ncon = c, d['r%d_ncon' % r]
        self.attempt_step, self.faulted, self.calls = 0, kind == 0, 0

    def on_reset(self, env, xy, status, env_mask, test):
        self.resets += 1
        self.attempt_step = 0

    def __call__(self, env, actions):
        self.calls += 1
        t = self.attempt_step
        self.attempt_step += 1
        env.observation.zero_()                       # "physics" writes the whole observation every step
        if not self.faulted:
            pairs = torch.as_tensor(np.ascontiguousarray(self.dcon[:, min(t, self.dcon.shape[1] - 1)]), device=DEV)
            ncon = torch.as_tensor(np.ascontiguousarray(self.dncon[:, min(t, self.dncon.shape[1] - 1)]), device=DEV)
            if t == self.k:
                self.faulted = True
                if self.kind == 2:
                    # envs j .. B-1 have not stepped in the reference: their lists of this step do not exist
                    raise bpg.PhysicsError('scripted simulation failure', failed_env=self.j, contacts=(pairs, ncon))
                env.observation[self.j, 0] = float('nan')
            return pairs, ncon
        return (torch.as_tensor(np.ascontiguousarray(self.con[:, t]), device=DEV),
                torch.as_tensor(np.ascontiguousarray(self.ncon[:, t]), device=DEV))


def _replay_restarts(name, pg, nan_check='per_step'):
    env_id = name + '-v0'
    d = np.load(os.path.join(GOLDEN, ('pg_restart_%s.npz' if pg else 'restart_%s.npz') % name))
    B, T = int(d['B']), int(d['T'])
    backend = _FaultyBackend(B, VARIANTS[env_id]['dimo'])
    src = bpg.CallableContacts(backend, on_reset=backend.on_reset, writes_observation=True)

    class Replay:
        kwargs = {}

        def __init__(self):
            self.calls, self.seen = 0, []

        def get_actions(self, o, ag, g, **kw):
            self.seen.append(dict(kw))
            u = self.u[self.calls]
            self.calls += 1
            return (u, u.copy(), np.zeros_like(u)) if pg else u

    policy, log = Replay(), _Log()
    make = lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src)   # noqa: E731
    if pg:
        worker = bpg.PGRolloutStudent(make, policy, None, dims_of(env_id), log, T, rollout_batch_size=B, gamma=1. - 1. / T,
                                      nan_check=nan_check)
    else:
        worker = bpg.RolloutStudent(make, policy, dims_of(env_id), log, T, rollout_batch_size=B, nan_check=nan_check)
    for r in range(int(d['n_calls'])):
        backend.load(d, r, T)
        # the policy's actions in call order: the discarded attempt's, then the returned episode's (time-major)
        policy.u = np.concatenate([d['r%d_discard_u' % r], d['r%d_u' % r].transpose(1, 0, 2)])
        policy.calls, policy.seen = 0, []
        draws = iter(d['r%d_reset_num_objs' % r] - 4)            # one row per reset of this call
        n_before, h_before, resets_before, lines_before = worker.n_episodes, len(worker.success_history), backend.resets, len(log.lines)
        kw = dict(exploit=True) if pg and name == 'BlocksTouch' else {}    # make_golden: exploit_first
        ep = worker.generate_rollouts(num_grey=lambda: next(draws), **kw)
        for k in ('ag', 'g', 'u', 'info_is_success'):
            np.testing.assert_array_equal(ep[k], d['r%d_%s' % (r, k)], err_msg='call %d: %s' % (r, k))
        if pg:
            np.testing.assert_array_equal(ep['G'].view(np.uint64), d['r%d_G' % r].view(np.uint64))
        assert worker.success_history[-1] == pytest.approx(float(d['r%d_success_rate' % r]), abs=1e-7)
        assert worker.n_episodes - n_before == int(d['r%d_n_episodes_added' % r])
        assert len(worker.success_history) - h_before == int(d['r%d_history_added' % r])
        assert policy.calls == int(d['r%d_policy_calls' % r])
        assert backend.resets - resets_before == d['r%d_reset_num_objs' % r].shape[0]
        assert next(draws, None) is None
        assert len(log.lines) - lines_before == (1 if d['faults'][r][0] else 0)
        # the state the envs are left in: the matrix and the latch
        np.testing.assert_array_equal(worker.env.achieved_goal_f64().cpu().numpy(), d['r%d_final_ag' % r])
        np.testing.assert_array_equal(((worker.env.state_ever >> 31) & 1).cpu().numpy().astype(np.uint8),
                                      d['r%d_final_has_succeeded' % r])
        if pg:
            np.testing.assert_array_equal(np.asarray([bool(k.get('exploit')) for k in policy.seen], np.uint8),
                                          d['r%d_exploit_seen' % r])
    assert worker.n_episodes == int(d['n_episodes'])


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation', 'ToppleTower', 'BlocksTouchChooseCurriculum', 'GripperTouch'])
def test_golden_restarts_rollout_student(name):
    """RolloutStudent.generate_rollouts around a NaN observation and a MujocoException (rollout.py:136-142), bit for bit
    what the reference returned -- including what the discarded attempt left in the touch matrix (quirk Q1)."""
    _replay_restarts(name, pg=False)


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation'])
def test_golden_restarts_pg_rollout_student(name):
    """policy_gradient/rollout.py:243-246: reset, then a retry whose `exploit` is back at its default."""
    _replay_restarts(name, pg=True)


def test_per_rollout_nan_check_is_the_documented_deviation():
    """nan_check='per_rollout' reads the flag once after all T steps: the discarded attempt runs to its end, so the
    backend is called T times instead of t+1 and (quirk Q1) the retried episode of a base variant can start from more
    touches than the reference's.  The test pins the call count and that the returned episode is still a clean one."""
    d = np.load(os.path.join(GOLDEN, 'restart_BlocksTouch.npz'))
    B, T = int(d['B']), int(d['T'])
    env_id = 'BlocksTouch-v0'
    backend = _FaultyBackend(B, VARIANTS[env_id]['dimo'])
    src = bpg.CallableContacts(backend, on_reset=backend.on_reset, writes_observation=True)

    class Zero:
        def get_actions(self, o, ag, g, **kw):
            return np.zeros((B, 4), np.float32)

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src), Zero(),
                           dims_of(env_id), _Log(), T, rollout_batch_size=B, nan_check='per_rollout')
    backend.load(d, 1, T)                              # ('nan', env 2, step 7)
    ep = w.generate_rollouts()
    assert backend.calls == T + T                      # per_step (the reference): 8 + T
    assert not np.isnan(ep['o']).any() and w.n_episodes == B and len(w.success_history) == 1
    with pytest.raises(ValueError):
        bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), Zero(), dims_of(env_id), None, T,
                           rollout_batch_size=B, nan_check='sometimes')


def test_policy_action_width_is_checked():
    """a (B, k != 4) float32 device tensor from the policy must not reach the kernel as a raw pointer."""
    env_id, B, T = 'BlocksTouch-v0', 16, 3

    class Wide:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return torch.zeros((B, 5), device=DEV)

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), Wide(), dims_of(env_id), None, T,
                           rollout_batch_size=B, output='torch')
    with pytest.raises(ValueError):
        w.generate_rollouts()
    with pytest.raises(ValueError):
        w.env._step_with_args(torch.zeros((B, 3), device=DEV), _lib.StepArgs())


# --------------------------------------------------------------------------
# the byte form of the episode return (bpg_rollout_args.d_episode_fail_count)
# --------------------------------------------------------------------------
def bits32(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize('B,T', [(1, 50), (1003, 50), (65536, 50), (262144, 7), (524288, 50), (1048576, 50), (4096, 255)])
def test_episode_fail_count_is_the_return_in_one_byte(B, T):
    """Every launch shape of the stream path, the episode writer, one and several rollouts per launch, with and without the
    step-ok words: d_episode_fail_count[e] == number of -1.0f rewards of env e, and expanding it gives the float32 return
    the same launch wrote, bit for bit (-0.0f where every step succeeded)."""
    L = _lib.lib()
    env_id = 'BlocksTouchVariation-v0'
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=3)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for K in (1, 3):
        masks = bpg.time_major_empty(K * T, B, torch.int32, DEV)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, K * T, C.c_void_p(masks.data_ptr()), int(masks.stride(0)) if K * T > 1 else B, st))
        masks |= (torch.rand((K * T, B), device=DEV) < 0.2).to(torch.int32) << 14        # pair (2, 3): some successes
        reward = bpg.time_major_empty(K * T, B, torch.float32, DEV)
        ret = torch.empty((K, B), dtype=torch.float32, device=DEV)
        cnt = torch.full((K, B), 77, dtype=torch.uint8, device=DEV)
        a = _lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_masks=masks.data_ptr(), ld_masks=int(masks.stride(0)),
                             d_reward_tb=reward.data_ptr(), ld_out=int(reward.stride(0)), d_episode_return=ret.data_ptr(),
                             d_episode_fail_count=cnt.data_ptr())
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), st))
        want = (reward.view(K, T, B) == -1.0).sum(1)
        assert torch.equal(cnt.to(torch.int64), want)
        np.testing.assert_array_equal(bits32(bpg.distributed.episode_return_from_fail_count(cnt.view(-1))), bits32(ret.view(-1)))
        # the count alone (no float32 return asked for), and with the step-ok words (d_G) in the same launch
        cnt2 = torch.zeros(B, dtype=torch.uint8, device=DEV)
        G = torch.empty((B, T), dtype=torch.float64, device=DEV) if T <= 64 else None
        a = _lib.RolloutArgs(T=T, reset_first=1, d_masks=masks.data_ptr(), ld_masks=int(masks.stride(0)),
                             d_episode_fail_count=cnt2.data_ptr(), d_G=None if G is None else G.data_ptr(), gamma=0.98)
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), st))
        assert torch.equal(cnt2, cnt[0])
    # the episode writer
    if B <= 65536:
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None, dims_of(env_id), None, T,
                               rollout_batch_size=B, output='torch')
        ep = w.generate_rollouts_fused(masks=masks[:T], want_reward=True, episode_return=True, episode_fail_count=True)
        assert torch.equal(ep['episode_fail_count'].to(torch.int64), (ep['reward_tb'] == -1.0).sum(0))
        np.testing.assert_array_equal(bits32(bpg.distributed.episode_return_from_fail_count(ep['episode_fail_count'])),
                                      bits32(ep['episode_return']))
    if T == 255:                                    # one byte per env: longer horizons are refused
        a = _lib.RolloutArgs(T=256, reset_first=1, d_episode_fail_count=cnt2.data_ptr())
        assert L.bpg_rollout_v2(env._h, C.byref(a), st) == _lib.ERR_ARG
    # unaligned / odd-length expansion
    c = torch.randint(0, 256, (1000003,), dtype=torch.uint8, device=DEV)
    for off in (0, 1, 5):
        out = bpg.distributed.episode_return_from_fail_count(c[off:].contiguous() if off == 0 else c[off:].clone())
        np.testing.assert_array_equal(bits32(out), bits32(-(c[off:].to(torch.float32))))


# --------------------------------------------------------------------------
# curriculum in the handle (fetch_env.py:351-358, :419-432, :623-630)
# --------------------------------------------------------------------------
def test_curriculum_traces_through_the_c_abi():
    """The difficulty traces recorded from the reference's increase_difficulty (variants.json) replayed through
    bpg_increase_difficulty / bpg_get_difficulty / bpg_get_curriculum -- the state lives in the handle, the Python env's
    attributes are views of it -- including the reference's exceptions (NotImplementedError, AttributeError: quirk Q6)."""
    L = _lib.lib()
    for env_id, v in VARIANTS.items():
        env = bpg.make(env_id, num_envs=4, device=DEV)
        top, cur = C.c_int(-1), _lib.Curriculum()
        if 'error' in v['difficulty']:
            want = {'NotImplementedError': _lib.ERR_UNSUPPORTED, 'AttributeError': _lib.ERR_ATTRIBUTE}[v['difficulty']['error']]
            assert L.bpg_increase_difficulty(env._h, C.byref(top)) == want
            assert L.bpg_get_difficulty(env._h) == 0
            continue
        for row in v['difficulty']['rows']:
            assert L.bpg_increase_difficulty(env._h, C.byref(top)) == 0
            assert bool(top.value) is row['ret'] and L.bpg_get_difficulty(env._h) == row['difficulty'] == env.get_difficulty()
            _lib.check(L.bpg_get_curriculum(env._h, C.byref(cur)))
            assert cur.obj_range == row['obj_range'] == env.obj_range
            if row['wrong_obj_range'] is not None:
                assert cur.wrong_obj_range == row['wrong_obj_range'] == env.wrong_obj_range
        # what a placement at this level samples from: reset() the current ranges, set_test() the maximum ones
        pa = _lib.PlacementArgs()
        _lib.check(L.bpg_placement_from_curriculum(env._h, 0, C.byref(pa)))
        assert (pa.obj_range, pa.max_obj_range, pa.test) == (cur.obj_range, cur.max_obj_range, 0)
        rc = L.bpg_placement_from_curriculum(env._h, 1, C.byref(pa))
        assert rc == (0 if v['set_test'] == 'ok' else _lib.ERR_UNSUPPORTED)
        # a checkpoint carries the level
        sd = env.state_dict()
        other = bpg.make(env_id, num_envs=4, device=DEV)
        other.load_state_dict(sd)
        assert other.get_difficulty() == env.get_difficulty() and other.obj_range == env.obj_range
        # the attribute views write through
        env.obj_range = 0.123
        _lib.check(L.bpg_get_curriculum(env._h, C.byref(cur)))
        assert cur.obj_range == 0.123


# --------------------------------------------------------------------------
# collectives issued by the library (bpg_comm_*): one rank is all a one-GPU box can hold
# --------------------------------------------------------------------------
def test_library_collectives_on_a_one_rank_communicator():
    """ncclAllGather / ncclAllReduce from librccl.so (loaded lazily by libbpgym.so) on the caller's stream: with one rank
    the gather is a copy and the reduce the identity -- the path (dlopen, communicator, datatype, stream) is the one N ranks
    take.  Multi-rank RCCL needs one GPU per rank: the driver's 8-GPU run is the first to see it."""
    B, T = 65536, 50
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=5)
    comm = bpg.distributed.RcclComm.from_process_group(torch.device(DEV))
    assert (comm.rank, comm.world) == (0, 1)
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):
        cnt = torch.empty(B, dtype=torch.uint8, device=DEV)
        a = _lib.RolloutArgs(T=T, reset_first=1, d_episode_fail_count=cnt.data_ptr())
        st = C.c_void_p(side.cuda_stream)
        _lib.check(_lib.lib().bpg_rollout_v2(env._h, C.byref(a), st))
        gathered = comm.all_gather(cnt)                         # same stream: ordered behind the launch without an event
        ret = bpg.distributed.episode_return_from_fail_count(gathered)
        f = comm.all_gather(ret)
        stats = torch.tensor([[3, B], [5, B]], dtype=torch.int64, device=DEV)
        comm.all_reduce_sum_(stats)
    side.synchronize()
    assert torch.equal(gathered, cnt) and torch.equal(f, ret) and stats.tolist() == [[3, B], [5, B]]
    assert int(cnt.max()) <= T and bool((ret <= 0).all())
    comm.close()
    L = _lib.lib()
    assert L.bpg_allgather_u8(None, None, None, 1, None) == _lib.ERR_ARG
    assert L.bpg_comm_rank(None) == -1 and L.bpg_comm_world_size(None) == 0 and L.bpg_comm_destroy(None) == 0
    h = C.c_void_p()
    assert L.bpg_comm_adopt(C.byref(h), None, 0, 1, 0) == _lib.ERR_ARG


def test_return_scratch_is_reserved_up_front_and_all_gather_fills_out():
    """the step-ok words behind d_G are allocated by bpg_reserve_return_scratch (the pg worker's constructor
    calls it), so the FIRST d_G rollout can already sit inside a stream capture; all_gather_per_env fills `out` on the
    no-group path too."""
    L = _lib.lib()
    B, T = 4096, 50
    env = bpg.make('BlocksTouch-v0', num_envs=B, device=DEV)
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)
    a = _lib.RolloutArgs(T=T, reset_first=1, d_G=G.data_ptr(), gamma=0.98)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(DEV)
    with torch.cuda.stream(s):
        s.synchronize()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g, stream=s):
                rc = L.bpg_rollout_v2(env._h, C.byref(a), C.c_void_p(s.cuda_stream))
        finally:
            pass
    assert rc == _lib.ERR_ARG and b'bpg_reserve_return_scratch' in L.bpg_last_error()
    assert L.bpg_reserve_return_scratch(env._h) == 0 and L.bpg_reserve_return_scratch(env._h) == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    g.replay()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(G).all())
    x = torch.arange(7, dtype=torch.float32, device=DEV)
    out = torch.full((7,), -1.0, device=DEV)
    assert bpg.distributed.all_gather_per_env(x, total_envs=7, out=out) is out and torch.equal(out, x)
    # the round-3 names are stubs now (a stale caller gets an error code, whatever it passes)
    assert L.bpg_step_record() == _lib.ERR_UNSUPPORTED and L.bpg_rollout() == _lib.ERR_UNSUPPORTED


# --------------------------------------------------------------------------
# the fused workers as stream shards (views of the batch) == one launch
# --------------------------------------------------------------------------
@pytest.mark.parametrize('env_id,B,S,explicit', [('BlocksTouchVariation-v0', 1048576, 'auto', True),
                                                 ('BlocksTouchVariation-v0', 524288 + 12, 'auto', False),
                                                 ('BlocksTouch-v0', 40004, 3, True), ('ToppleTower-v0', 1003, 2, True)])
def test_sharded_evaluator_equals_one_launch(env_id, B, S, explicit):
    """RolloutStudent.evaluate_rollouts_fused(shards='auto': from 524 288 envs on, ranges of about 262 144 envs on streams of
    their own; or an explicit count) against the default one launch: every stream, every per-env array, every success rate
    and the state the batch is left in, bit for bit -- explicit num_grey draws and the generator's (whose stream index the
    sharded form advances once on the parent)."""
    T, K = 50 if B < 100000 else 10, 3
    dims = dims_of(env_id)
    mk = lambda: bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=17), None, dims, None, T,   # noqa: E731
                                    rollout_batch_size=B, output='torch')
    one, many = mk(), mk()
    assert bpg.rollout.auto_shards(1048576) == 4 and bpg.rollout.auto_shards(524288) == 2 and bpg.rollout.auto_shards(400000) == 1
    gen = torch.Generator(device=DEV).manual_seed(B)
    masks = bpg.time_major_empty(K * T, B, torch.int32, DEV)
    masks.copy_(torch.randint(0, 1 << 21, (K * T, B), device=DEV, generator=gen, dtype=torch.int32)
                & torch.randint(0, 1 << 21, (K * T, B), device=DEV, generator=gen, dtype=torch.int32))
    ng = torch.randint(0, 3, (K, B), device=DEV, generator=gen, dtype=torch.int32) if explicit else None
    outs = []
    for w, sh in ((one, 1), (many, S)):
        for rep in range(2):                       # twice: the second call starts from the state (and stream index) of the first
            rw = bpg.time_major_empty(K * T, B, torch.float32, DEV)
            fl = torch.empty((K * T, int(rw.stride(0))), dtype=torch.uint8, device=DEV)[:, :B]
            er = torch.empty((K, B), dtype=torch.float32, device=DEV)
            ef = torch.empty((K, B), dtype=torch.uint8, device=DEV)
            rates = w.evaluate_rollouts_fused(masks, num_grey=ng, reward_tb=rw, flags_tb=fl, episode_return=er,
                                              episode_fail_count=ef, shards=sh)
        torch.cuda.synchronize()
        outs.append((rates, rw, fl, er, ef, w.env.state_now.clone(), w.env.state_ever.clone(), w.env.state_counters.clone(),
                     w.env.observation.clone()))
    if S == 'auto':
        assert len(many._view_shards) == 1 and len(next(iter(many._view_shards.values()))) == bpg.rollout.auto_shards(B)
    # the form that pays: arrays of the shards' own (shard_streams), one per range of shard_ranges
    own = mk()
    S_own = len(many.shard_ranges(S))
    ranges = own.shard_ranges(S_own)
    assert ranges == many.shard_ranges(S) and sum(n for _, n in ranges) == B and all(lo % 4 == 0 for lo, _ in ranges)
    m_own = own.shard_streams(K * T, torch.int32, S_own)
    for t, (lo, n) in zip(m_own, ranges):
        t.copy_(masks[:, lo:lo + n])
    for rep in range(2):
        rw_own, fl_own = own.shard_streams(K * T, torch.float32, S_own), own.shard_streams(K * T, torch.float32, S_own)
        fl_own = [torch.empty((K * T, int(r.stride(0))), dtype=torch.uint8, device=DEV)[:, :r.shape[1]] for r in rw_own]
        er = torch.empty((K, B), dtype=torch.float32, device=DEV)
        ef = torch.empty((K, B), dtype=torch.uint8, device=DEV)
        rates = own.evaluate_rollouts_fused(m_own, num_grey=ng, reward_tb=rw_own, flags_tb=fl_own, episode_return=er, episode_fail_count=ef)
    torch.cuda.synchronize()
    outs.append((rates, torch.cat(rw_own, 1), torch.cat(fl_own, 1), er, ef, own.env.state_now.clone(), own.env.state_ever.clone(),
                 own.env.state_counters.clone(), own.env.observation.clone()))
    a, b, c = outs
    assert a[0] == b[0] == c[0]
    for x, y, name in zip(a[1:], c[1:], ('reward', 'flags', 'episode_return', 'episode_fail_count', 'now', 'ever', 'counters', 'obs')):
        assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y), 'own arrays: ' + name
    with pytest.raises(ValueError):
        own.evaluate_rollouts_fused(m_own[:-1] + [m_own[-1], m_own[-1]])          # one array too many for the shards there are
    for x, y, name in zip(a[1:], b[1:], ('reward', 'flags', 'episode_return', 'episode_fail_count', 'now', 'ever', 'counters', 'obs')):
        assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y), name
    assert list(one.success_history) == list(many.success_history) and one.n_episodes == many.n_episodes == 2 * K * B


@pytest.mark.parametrize('env_id,B,S', [('BlocksTouchVariation-v0', 8192, 4), ('BlocksTouchChoose-v0', 1003, 3)])
def test_sharded_episode_writer_equals_one_launch(env_id, B, S):
    """generate_rollouts_fused(shards=S): the episode arrays, reward stream, G, episode return / fail count and the success
    rate of S view launches on S streams are those of the one launch."""
    T = 50
    dims = dims_of(env_id)
    mk = lambda cls, **kw: cls(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=3), None, *kw.pop('pre', ()),   # noqa: E731
                               dims, None, T, rollout_batch_size=B, output='torch', **kw)
    one, many = mk(bpg.PGRolloutStudent, pre=(None,), gamma=0.98), mk(bpg.PGRolloutStudent, pre=(None,), gamma=0.98)
    gen = torch.Generator(device=DEV).manual_seed(7)
    for rep in range(2):
        masks = (torch.randint(0, 1 << 21, (T, B), device=DEV, generator=gen, dtype=torch.int32)
                 & torch.randint(0, 1 << 21, (T, B), device=DEV, generator=gen, dtype=torch.int32))
        ng = torch.randint(0, 3, (B,), device=DEV, generator=gen, dtype=torch.int32)
        kw = dict(masks=masks, num_grey=ng, want_reward=True, episode_return=True, episode_fail_count=True, materialize_constants=rep == 1)
        e1, e2 = one.generate_rollouts_fused(shards=1, **kw), many.generate_rollouts_fused(shards=S, **kw)
        assert set(e1) == set(e2)
        for k in e1:
            x, y = e1[k], e2[k]
            assert x.shape == y.shape and x.dtype == y.dtype, k
            if x.is_floating_point():
                x, y = x.contiguous().view(torch.int32 if x.dtype == torch.float32 else torch.int64), \
                    y.contiguous().view(torch.int32 if y.dtype == torch.float32 else torch.int64)
            assert torch.equal(x, y), k
        assert one.success_history[-1] == many.success_history[-1]
    # the C ABI's own checks for views
    L = _lib.lib()
    h = C.c_void_p()
    env = one.env
    assert L.bpg_create_view(C.byref(h), env._h, 2, 8) == _lib.ERR_ARG                      # not 16-byte aligned
    assert L.bpg_create_view(C.byref(h), env._h, 0, B + 1) == _lib.ERR_ARG
    _lib.check(L.bpg_create_view(C.byref(h), env._h, 4, 8))
    assert L.bpg_num_envs(h) == 8 and L.bpg_seed(h, 1) == _lib.ERR_ARG and L.bpg_advance_generator(h, 1, 0, None) == _lib.ERR_ARG
    h2 = C.c_void_p()
    assert L.bpg_create_view(C.byref(h2), h, 0, 4) == _lib.ERR_ARG                          # no views of views
    assert L.bpg_get_difficulty(h) == env.get_difficulty()
    _lib.check(L.bpg_destroy(h))


def test_rollout_train_argument_checks():
    """bpg_rollout_train: one handle and one stream per shard, all on one device, the caller's struct size as the stride."""
    L = _lib.lib()
    env = bpg.make('BlocksTouch-v0', num_envs=64, device=DEV)
    h = (C.c_void_p * 2)(env._h.value, env._h.value)
    s0, s1 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    st = (C.c_void_p * 2)(s0.cuda_stream, s1.cuda_stream)
    a = (_lib.RolloutArgs * 2)(_lib.RolloutArgs(T=3, reset_first=1), _lib.RolloutArgs(T=3, reset_first=1))
    size = C.sizeof(_lib.RolloutArgs)
    assert L.bpg_rollout_train(h, st, 2, a, size, 1, None) == _lib.ERR_ARG and b'of its own' in L.bpg_last_error()
    assert L.bpg_rollout_train(h, st, 0, a, size, 1, None) == _lib.ERR_ARG
    assert L.bpg_rollout_train(h, st, 1, a, 12, 1, None) == _lib.ERR_ARG
    assert L.bpg_rollout_train(None, st, 1, a, size, 1, None) == _lib.ERR_ARG
    assert L.bpg_rollout_train(h, st, 1, a, size, 2, None) == 0          # one shard, two launches: a plain train on its stream
    torch.cuda.synchronize()
    assert int((env.state_now >> 21).max()) == 3
