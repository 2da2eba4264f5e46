"""GPU tests for the reset -> placement -> backend wiring, the backend-failure / NaN restart of the
rollout workers, the policy-in-the-loop path at 1 048 576 envs, the launch shapes of the fused
rollout at BASELINE.json's other batch sizes, and the stateless kernels' argument ranges.
Bit-exact against the oracle / golden fixtures unless a test says otherwise."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from oracle import bpg_oracle as O
from util import GOLDEN, VARIANTS, ENV_IDS, short, legal_masks

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dims_of(env_id):
    v = VARIANTS[env_id]
    return dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


# --------------------------------------------------------------------------
# reset() / set_test() place the cubes and hand them to the backend
# --------------------------------------------------------------------------
def _placement_group(env_id, want_test):
    """Rows of the placement fixture that share launch constants (one kernel launch)."""
    d = np.load(os.path.join(GOLDEN, 'placement_%s.npz' % short(env_id)))
    keys = np.stack([d['grip_xy'][:, 0], d['grip_xy'][:, 1], d['obj_range'], np.nan_to_num(d['max_obj_range']),
                     np.nan_to_num(d['wrong_obj_range']), d['test'].astype(float), d['challenge'].astype(float)], 1)
    groups = {}
    for k, row in enumerate(map(tuple, keys)):
        groups.setdefault(row, []).append(k)
    for row, idx in groups.items():
        if bool(row[5]) == want_test and not row[6]:
            return d, row, np.asarray(idx)
    return d, None, None


def _variates(d, idx):
    var = {k: np.nan_to_num(d[k][idx], nan=0.5) for k in ('eu', 'gz', 'gu')}
    for k in var:
        if var[k].shape[1] < 2:
            var[k] = np.concatenate([var[k], np.full((len(idx), 2), 0.5)], 1)
    return var


@pytest.mark.parametrize('env_id', ['BlocksTouchCurriculum-v0', 'BlocksTouchChooseCurriculum-v0',
                                    'BlocksTouchVariation-v0', 'GripperTouch-v0'])
def test_reset_hands_the_golden_placement_to_the_backend(env_id):
    """fetch_env.py:247-255: reset() -> _reset_sim -> _randomize_objects.  With the recorded
    standard variates the positions a CallableContacts backend receives in on_reset are the
    reference's, bit for bit; set_test() (fetch_env.py:365-368, :443-446) re-places at the
    maximum ranges (Variation's set_test does not, :641-644)."""
    seen = []

    def on_reset(env, xy, status, env_mask, test):
        assert xy is env.block_xy and status is env.placement_status      # the env's own buffers, no copies
        seen.append((xy.clone(), status.clone(), None if env_mask is None else env_mask.clone(), test))

    for want_test in (False, True):
        d, row, idx = _placement_group(env_id, want_test)
        if row is None:
            continue
        env = bpg.make(env_id, num_envs=len(idx), device=DEV,
                       contact_source=bpg.CallableContacts(lambda e, a: None, on_reset=on_reset))
        env.obj_range, env.max_obj_range, env.wrong_obj_range = row[2], row[3], row[4]
        env.grip_xy = row[:2]
        seen.clear()
        if not want_test:
            env.reset(num_grey=d['num_objs'][idx] - 4, variates=_variates(d, idx))
        else:
            env.reset(num_grey=d['num_objs'][idx] - 4)              # generator-drawn placement first
            if not env.info.set_test_supported:
                with pytest.raises(NotImplementedError):
                    env.set_test()
                continue
            env.set_test(variates=_variates(d, idx))
            if env.info.is_variation:                               # :641-644 returns _get_obs() only
                assert len(seen) == 1
                continue
            assert len(seen) == 2 and seen[0][3] is False
        xy, status, em, test = seen[-1]
        assert test is want_test and em is None and (status == 0).all()
        got, exp = xy.cpu().numpy(), d['xy'][idx]
        for j, k in enumerate(idx):
            nb = int(d['num_objs'][k]) - 2
            np.testing.assert_array_equal(got[j, :nb].view(np.uint64), exp[j, :nb].view(np.uint64))


def test_partial_reset_places_only_the_masked_envs():
    """reset_rollout(i) (rollout.py:47-58): only env i is re-placed, the hook sees the mask."""
    B = 64
    seen = []
    env = bpg.make('BlocksTouchChooseCurriculum-v0', num_envs=B, device=DEV, seed=3,
                   contact_source=bpg.CallableContacts(lambda e, a: None,
                                                       on_reset=lambda e, xy, st, em, test: seen.append((em, test))))
    env.reset()
    first = env.block_xy.clone()
    assert not torch.isnan(first).any() and seen[-1] == (None, False)
    mask = torch.zeros(B, dtype=torch.uint8, device=DEV)
    mask[[3, 40]] = 1
    env.reset(env_mask=mask)
    changed = (env.block_xy != first).flatten(1).any(1).cpu().numpy()
    assert changed[[3, 40]].all() and changed.sum() == 2
    assert torch.equal(seen[-1][0], mask) and seen[-1][1] is False
    # the rollout worker's reset_rollout goes through the same path (test mode: set_test for env i only)
    w = bpg.RolloutStudent(lambda num_envs: env.with_num_envs(num_envs), None, dims_of(env.env_id), None, 5,
                           rollout_batch_size=B)
    before = w.env.block_xy.clone()
    n_calls = len(seen)
    w.reset_rollout(7, test=True)
    moved = (w.env.block_xy != before).flatten(1).any(1).cpu().numpy()
    assert moved[7] and moved.sum() == 1
    assert len(seen) == n_calls + 2 and seen[-1][1] is True and int(seen[-1][0].sum()) == 1


def test_sources_that_need_no_positions_skip_the_sampler():
    env = bpg.make('BlocksTouch-v0', num_envs=32, device=DEV)
    env.reset()
    assert torch.isnan(env.block_xy).all()                          # nothing consumed them: not sampled
    s = bpg.ScriptedContacts(masks=torch.zeros((3, 32), dtype=torch.int32, device=DEV))
    env = bpg.make('BlocksTouch-v0', num_envs=32, device=DEV, contact_source=s)
    env.reset(); env.step(None); env.step(None)
    assert s.t == 2
    env.reset(env_mask=np.ones(32, np.uint8))                       # a partial reset does not rewind the script
    assert s.t == 2
    env.reset()
    assert s.t == 0


def test_variation_static_observation_is_written_by_the_reset_launch():
    """fetch_env.py:579-607 without physics: obs[0] = num_blocks, one-hot colours of present
    blocks, zero padding; the physics-owned entries are zeroed by reset() (nothing behind the
    boundary knows them) and left alone by a state restore."""
    B = 1000
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV)
    assert (env.observation[:, 0] == 4).all()                       # constructor state: all four blocks
    env.observation.fill_(7.0)
    ng = np.random.RandomState(0).randint(3, size=B).astype(np.int32)
    env.reset(num_grey=ng)
    o = env.observation.cpu().numpy()
    nb = ng + 2
    colors = VARIANTS['BlocksTouchVariation-v0']['obj_colors']
    exp = np.zeros((B, 87), np.float32)
    exp[:, 0] = nb
    for s in range(4):
        sl = slice(11 + 19 * s, 11 + 19 * (s + 1))
        present = nb > s
        blk = exp[:, sl]
        blk[~present] = 0.0
        onehot = np.zeros(4, np.float32)
        onehot[colors[s + 2]] = 1.0
        blk[present, 15:] = onehot
        exp[:, sl] = blk
    np.testing.assert_array_equal(o, exp)
    # partial reset: only masked rows are rewritten
    env.observation.fill_(-3.0)
    mask = np.zeros(B, np.uint8); mask[::10] = 1
    env.reset(num_grey=ng, env_mask=mask)
    o = env.observation.cpu().numpy()
    assert (o[mask == 0] == -3.0).all() and (o[mask == 1, 0] == nb[mask == 1]).all()
    # strided target (a wider buffer) and a caller-provided buffer that becomes env.observation
    wide = torch.full((B, 100), 5.0, device=DEV)
    env.reset(num_grey=ng, observation_out=wide[:, :87])
    assert env.observation.data_ptr() == wide.data_ptr()
    np.testing.assert_array_equal(wide[:, :87].cpu().numpy(), exp)
    assert (wide[:, 87:] == 5.0).all()
    # a restored checkpoint refreshes the physics-free entries from the state, nothing else
    sd = env.state_dict()
    env.reset(num_grey=(ng + 1) % 3)
    env.observation[:, 1:11] = 9.0
    env.load_state_dict(sd)
    assert (env.observation[:, 0].cpu().numpy() == nb).all() and (env.observation[:, 1:11] == 9.0).all()
    # the other variants have no physics-free entries: reset zeroes the rows
    e2 = bpg.make('BlocksTouch-v0', num_envs=300, device=DEV)
    e2.observation.fill_(3.0)
    e2.reset(env_mask=(np.arange(300) % 2).astype(np.uint8))
    o2 = e2.observation.cpu().numpy()
    assert (o2[1::2] == 0).all() and (o2[0::2] == 3.0).all()


# --------------------------------------------------------------------------
# backend failures restart the rollout (rollout.py:136-142)
# --------------------------------------------------------------------------
class _Logger:
    def __init__(self):
        self.lines = []

    def warning(self, m):
        self.lines.append(m)

    info = warning


def _flaky_worker(env_id, B, T, fail_at, mode, cls=bpg.RolloutStudent, **kw):
    d = np.load(os.path.join(GOLDEN, 'obs_%s.npz' % short(env_id)))
    phys = {k: np.repeat(d[k][:1], B, 0) for k in bpg.BatchedBlocksEnv.PHYS_FIELDS}
    state = dict(calls=0, resets=0)

    row = torch.as_tensor(d['obs'][0].astype(np.float32), device=DEV)

    def physics(env, actions):
        state['calls'] += 1
        if state['calls'] == fail_at and mode == 'raise':
            raise bpg.PhysicsError('simulation unstable')
        if mode == 'nan_direct':                    # a backend that writes the observation by its own means:
            env.observation.copy_(row.expand(B, -1))      # the worker's separate NaN pass has to find it
            if state['calls'] == fail_at:
                env.observation[5, 3] = float('nan')
        else:                                       # 'nan': the physics state blows up -> caught inside the
            ph = phys                               # assemble launch (bpg_phys_state.d_nan_flag), no second pass
            if state['calls'] == fail_at and mode == 'nan':
                ph = dict(phys)
                ph['grip_velp'] = phys['grip_velp'].copy()
                ph['grip_velp'][5, 1] = np.nan
            env.assemble_observation(ph, dt=float(d['dt']))
        return torch.zeros(B, dtype=torch.int32, device=DEV)

    def on_reset(env, xy, status, env_mask, test):
        state['resets'] += 1

    class Pol:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            z = torch.zeros((B, 4), device=DEV)
            return (z, z, z) if cls is bpg.PGRolloutStudent else z

    log = _Logger()
    src = bpg.CallableContacts(physics, on_reset=on_reset, writes_observation=True)
    make = lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src)
    if cls is bpg.PGRolloutStudent:
        w = cls(make, Pol(), None, dims_of(env_id), log, T, rollout_batch_size=B, gamma=0.98, **kw)
    else:
        w = cls(make, Pol(), dims_of(env_id), log, T, rollout_batch_size=B, **kw)
    exp_obs = torch.as_tensor(d['obs'][0].astype(np.float32), device=DEV)
    return w, state, log, exp_obs


@pytest.mark.parametrize('nan_check', ['per_step', 'per_rollout'])
@pytest.mark.parametrize('mode', ['raise', 'nan', 'nan_direct'])
def test_rollout_restarts_after_a_backend_failure(mode, nan_check):
    env_id, B, T = 'BlocksTouchChoose-v0', 64, 6
    w, state, log, exp_obs = _flaky_worker(env_id, B, T, fail_at=4, mode=mode, nan_check=nan_check)
    resets0 = state['resets']
    ep = w.generate_rollouts()
    # the failed attempt consumed 4 backend calls -- the exception, or (per_step, the reference's timing, rollout.py:139-142)
    # the step whose observation held the NaN -- or all 6 (per_rollout: NaN found at the end); then a clean rollout
    assert state['calls'] == (T if (mode != 'raise' and nan_check == 'per_rollout') else 4) + T
    assert len(log.lines) == 1 and ('NaN caught' in log.lines[0]) == (mode != 'raise')
    assert state['resets'] - resets0 == (2 if mode == 'raise' else 3)   # NaN: reset_all_rollouts + the restart's own reset
    assert not torch.isnan(ep['o']).any()
    assert torch.equal(ep['o'][:, 1:], exp_obs.expand(B, T, -1))
    assert w.n_episodes == B and len(w.success_history) == 1            # the failed attempt is not counted


@pytest.mark.parametrize('nan_check', ['per_step', 'per_rollout'])
def test_pg_rollout_restarts_on_nan_and_keeps_backend_errors_fatal(nan_check):
    env_id, B, T = 'BlocksTouch-v0', 32, 5
    w, state, log, exp_obs = _flaky_worker(env_id, B, T, fail_at=2, mode='nan', cls=bpg.PGRolloutStudent, nan_check=nan_check)
    ep = w.generate_rollouts()
    assert state['calls'] == (2 if nan_check == 'per_step' else T) + T and len(log.lines) == 1 and not torch.isnan(ep['o']).any()
    assert ep['G'].shape == (B, T)
    w, state, log, _ = _flaky_worker(env_id, B, T, fail_at=2, mode='raise', cls=bpg.PGRolloutStudent)
    with pytest.raises(bpg.PhysicsError):                               # policy_gradient/rollout.py:237-241 asserts
        w.generate_rollouts()


def test_a_backend_that_keeps_failing_gives_up():
    env_id, B, T = 'BlocksTouch-v0', 8, 3

    def physics(env, actions):
        raise bpg.PhysicsError('always')

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV,
                                                     contact_source=bpg.CallableContacts(physics)),
                           type('P', (), {'device_tensors': True,
                                          'get_actions': lambda self, o, ag, g, **kw: torch.zeros((B, 4), device=DEV)})(),
                           dims_of(env_id), None, T, rollout_batch_size=B)
    with pytest.warns(UserWarning), pytest.raises(RuntimeError):
        w.generate_rollouts()


def test_assemble_observation_reports_nans_in_its_own_launch():
    env_id, B = 'BlocksTouchVariation-v0', 1000
    d = np.load(os.path.join(GOLDEN, 'obs_%s.npz' % short(env_id)))
    phys = {k: np.repeat(d[k][:1], B, 0) for k in bpg.BatchedBlocksEnv.PHYS_FIELDS}
    env = bpg.make(env_id, num_envs=B, device=DEV)
    env.reset()
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    env.assemble_observation(phys, dt=float(d['dt']), nan_flag=flag)
    assert int(flag) == 0
    for key, idx in (('obj_velr', (999, 1, 2)), ('gripper_state', (0, 1)), ('obj_pos', (517, 0, 0))):
        flag.zero_()
        bad = dict(phys)
        bad[key] = phys[key].copy()
        bad[key][idx] = np.nan
        for dtype in (torch.float32, torch.float64):
            flag.zero_()
            o = env.assemble_observation(bad, dt=float(d['dt']), dtype=dtype, nan_flag=flag)
            assert int(flag) == 1 and bool(torch.isnan(o).any())
    # a NaN in the slot of a block the episode does not have is never written -> not reported
    nb = (env.num_objs_per_env() - 2).cpu().numpy()
    e = int(np.nonzero(nb < 4)[0][0])
    flag.zero_()
    bad = dict(phys)
    bad['obj_rot'] = phys['obj_rot'].copy()
    bad['obj_rot'][e, 3, 0] = np.nan
    o = env.assemble_observation(bad, dt=float(d['dt']), nan_flag=flag)
    assert int(flag) == 0 and not bool(torch.isnan(o).any())


def test_nan_flag_kernel():
    L = bpg._lib.lib()
    x = torch.randn(1000, 87, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)

    def run(t):
        bpg._lib.check(L.bpg_nan_flag_f32(C.c_void_p(t.data_ptr()), t.shape[0], t.shape[1], int(t.stride(0)),
                                          C.c_void_p(flag.data_ptr()), 0, None))
        return int(flag.item())
    assert run(x) == 0
    x[999, 86] = float('inf')
    assert run(x) == 0
    wide = torch.randn(1000, 100, device=DEV)
    wide[:, 87:] = float('nan')                                     # beyond cols: not looked at
    assert run(wide[:, :87]) == 0
    x[500, 3] = float('nan')
    assert run(x) == 1 and run(torch.zeros(4, 4, device=DEV)) == 1  # sticky until the caller clears it


# --------------------------------------------------------------------------
# policy in the loop at BASELINE config 5's size
# --------------------------------------------------------------------------
def test_per_step_worker_matches_fused_kernel_at_1m_envs():
    """1 048 576 envs, T = 50: RolloutStudent.generate_rollouts (one step launch per t, a
    device-resident policy between the steps) produces the same episode as the single fused
    launch on the same contacts -- achieved_goal rows, success flags, success rate."""
    env_id, B, T = 'BlocksTouchVariation-v0', 1 << 20, 50
    L = bpg._lib.lib()
    gen = bpg.make(env_id, num_envs=B, device=DEV, seed=5)
    num_grey = (torch.arange(B, device=DEV, dtype=torch.int32) * 5) % 3
    gen.reset(num_grey=num_grey)
    masks = torch.empty((T, B), dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks(gen._h, 0, T, C.c_void_p(masks.data_ptr()), None))
    del gen

    class Pol:
        device_tensors = True
        calls = 0

        def get_actions(self, o, ag, g, **kw):
            assert o.shape == (B, 87) and ag.shape == (B, 36) and g.shape == (B, 36)
            Pol.calls += 1
            return ag[:, :4] * 0.25                                  # reads the observation slot, no copies

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV,
                                                     contact_source=bpg.ScriptedContacts(masks=masks)),
                           Pol(), dims_of(env_id), None, T, rollout_batch_size=B)
    ep = w.generate_rollouts(num_grey=num_grey)
    assert Pol.calls == T
    assert ep['o'].shape == (B, T + 1, 87) and ep['o'].stride(1) == 0      # nothing wrote observations: a view
    rate_step = w.success_history[-1]
    ag_step = ep['ag']
    succ_step = ep['info_is_success']
    f = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None,
                           dims_of(env_id), None, T, rollout_batch_size=B, output='torch')
    ep2 = f.generate_rollouts_fused(masks=masks, num_grey=num_grey)
    assert torch.equal(ag_step, ep2['ag'])
    assert torch.equal(succ_step, ep2['info_is_success'])
    assert torch.equal(ep['u'][:, 1], ep2['ag'][:, 1, :4] * 0.25)
    assert rate_step == f.success_history[-1]


# --------------------------------------------------------------------------
# fused rollout at the other batch sizes of BASELINE.json (launch shapes)
# --------------------------------------------------------------------------
@pytest.mark.parametrize('B', [65536, 131072 + 2, 524288, 65537, 2097152, 4194304 + 2])   # the last two: grid-stride trips
@pytest.mark.parametrize('reset_first', [1, 0])
def test_fused_rollout_shapes_agree_with_single_steps(B, reset_first):
    """Every launch shape rollout_geometry() picks (1 / 2 envs per lane, pipelined / plain loop,
    odd batch -> 1 per lane) == T single-step launches: rewards, flags, final state, count."""
    env_id, T = 'BlocksTouchVariation-v0', 23            # 23 = 2 pipelined batches of 10 + 3 leftover steps
    L = bpg._lib.lib()
    rng = np.random.RandomState(B % 1000)
    ng = torch.as_tensor(rng.randint(3, size=B).astype(np.int32), device=DEV)
    a, b = (bpg.make(env_id, num_envs=B, device=DEV, seed=9) for _ in range(2))
    masks = torch.empty((T, B), dtype=torch.int32, device=DEV)
    for e in (a, b):
        e.reset(num_grey=ng)
    if not reset_first:                                  # continue an episode: per-lane elapsed counters
        for e in (a, b):
            e.step(None)
        part = torch.zeros(B, dtype=torch.uint8, device=DEV); part[::3] = 1
        for e in (a, b):
            e.reset(num_grey=ng, env_mask=part)
    bpg._lib.check(L.bpg_synth_masks(a._h, 3, T, C.c_void_p(masks.data_ptr()), None))
    reward = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flags = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(),
                                d_reward_tb=reward.data_ptr(), d_flags_tb=flags.data_ptr(),
                                d_success_count=cnt.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(a._h, C.byref(args), None))
    if reset_first:
        b.reset(num_grey=ng)
    for t in range(T):
        _, r, done, info = b.step(None, contact_masks=masks[t])
        assert torch.equal(r.view(torch.int32), reward[t].view(torch.int32))
        assert torch.equal((flags[t] & 1) != 0, info['is_success']) and torch.equal((flags[t] & 2) != 0, done)
    assert torch.equal(a.state_now, b.state_now) and torch.equal(a.state_ever, b.state_ever)
    assert int(cnt[0]) == int(b.success_count()) and int(cnt[0]) > 0
    # reward-only and flags-only instantiations write the same streams
    for kw in (dict(d_reward_tb=reward.data_ptr()), dict(d_flags_tb=flags.data_ptr()), dict()):
        r0, f0 = reward.clone(), flags.clone()
        c, c2 = (bpg.make(env_id, num_envs=B, device=DEV, seed=9) for _ in range(2))
        for e in (c, c2):                                   # both continue from the same (post-rollout) state
            e.load_state_dict(a.state_dict())
        args = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(), **kw)
        bpg._lib.check(L.bpg_rollout_v2(c._h, C.byref(args), None))
        full = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(),
                                    d_reward_tb=r0.data_ptr(), d_flags_tb=f0.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(c2._h, C.byref(full), None))
        assert torch.equal(c.state_now, c2.state_now) and torch.equal(c.state_ever, c2.state_ever)
        if 'd_reward_tb' in kw:
            assert torch.equal(reward.view(torch.int32), r0.view(torch.int32))
        if 'd_flags_tb' in kw:
            assert torch.equal(flags, f0)


@pytest.mark.parametrize('B', [65536, 524288])
def test_synthetic_rollout_shapes_vs_oracle_spec(B):
    """In-kernel contacts at both envs-per-lane shapes: the reward stream equals stepping the
    oracle's statement of the source (spot-checked envs), with and without the in-launch reset."""
    env_id, T, seed, base = 'BlocksTouchVariation-v0', 12, 77, 1000
    L = bpg._lib.lib()
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=seed, env_id_base=base)
    ng = torch.as_tensor(np.random.RandomState(1).randint(3, size=B).astype(np.int32), device=DEV)
    reward = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flags = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_reward_tb=reward.data_ptr(),
                                d_flags_tb=flags.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(args), None))
    pick = np.r_[0:64, B // 2:B // 2 + 64, B - 64:B]
    exp_masks = O.synth_masks(seed, base + pick, T, 4 + ng.cpu().numpy()[pick])
    r = reward.cpu().numpy()[:, pick]
    # reward is -0.0 exactly when the goal pair (2,3) touches in that step's mask
    p23 = O.lib().orc_pair_index(2, 3)
    np.testing.assert_array_equal(r.view(np.uint32) == 0x80000000, ((exp_masks >> p23) & 1) == 1)
    # continuing WITHOUT reset uses the per-lane elapsed counters: steps T .. 2T-1 of the same source
    args2 = bpg._lib.RolloutArgs(T=T, reset_first=0, d_reward_tb=reward.data_ptr(), d_flags_tb=flags.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(args2), None))
    exp2 = O.synth_masks(seed, base + pick, 2 * T, 4 + ng.cpu().numpy()[pick])[T:]
    np.testing.assert_array_equal(reward.cpu().numpy()[:, pick].view(np.uint32) == 0x80000000, ((exp2 >> p23) & 1) == 1)


def test_synthetic_source_rates_and_independence_on_the_gpu():
    """1 048 576 envs x 8 steps of the in-kernel source: every free pair is Bernoulli(5/32) to 4
    sigma, consecutive steps are uncorrelated, pairs of one step correlate by at most the 0.04 the
    shared low-order digit introduces (docs/HISTORY.md 5), table-block pairs are forced, the diagonal
    and absent objects never appear."""
    B, T = 1 << 20, 8
    L = bpg._lib.lib()
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=2024)
    env.reset(num_grey=torch.full((B,), 2, dtype=torch.int32, device=DEV))
    m = torch.empty((T, B), dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks(env._h, 0, T, C.c_void_p(m.data_ptr()), None))
    idx = O.lib().orc_pair_index
    forced = [idx(1, k) for k in range(2, 6)]
    free = [idx(i, j) for i in range(6) for j in range(i + 1, 6) if not (i == 1 and j >= 2)]
    diag = [idx(i, i) for i in range(6)]
    b = torch.stack([(m >> p) & 1 for p in range(21)], -1).to(torch.float32)          # (T, B, 21)
    rate = b.mean((0, 1)).cpu().numpy()
    assert (rate[forced] == 1).all() and (rate[diag] == 0).all()
    p, n = 5 / 32, T * B
    assert np.abs(rate[free] - p).max() < 4 * np.sqrt(p * (1 - p) / n)
    z = (b[..., free] - p) / np.sqrt(p * (1 - p))
    same = (z.reshape(-1, len(free)).T @ z.reshape(-1, len(free)) / n).cpu().numpy()
    off = same - np.diag(np.diag(same))
    assert np.abs(off).max() < 0.045 and np.abs(np.diag(same) - 1).max() < 0.01
    lag = (z[:-1].reshape(-1, len(free)).T @ z[1:].reshape(-1, len(free)) / ((T - 1) * B)).cpu().numpy()
    assert np.abs(lag).max() < 6 / np.sqrt((T - 1) * B)
    # neighbouring envs are independent too
    nb = (z[:, :-1].reshape(-1, len(free)) * z[:, 1:].reshape(-1, len(free))).mean(0).cpu().numpy()
    assert np.abs(nb).max() < 6 / np.sqrt(T * (B - 1))


# --------------------------------------------------------------------------
# stateless kernels: argument ranges, capture
# --------------------------------------------------------------------------
def test_discounted_return_any_horizon_and_graph_capture():
    """Generic horizons (LDS path) are bit-exact too, T > 319 is an argument error, and the
    call only enqueues work: it can be captured into a hipGraph and replayed."""
    rng = np.random.RandomState(4)
    gamma = 0.97
    for T, B in [(13, 70), (64, 65), (51, 130), (200, 64), (319, 5)]:
        r = rng.normal(size=(T, B)).astype(np.float32)
        G = bpg.discounted_return(torch.as_tensor(r, device=DEV), gamma).cpu().numpy()
        for b in range(0, B, 7):
            np.testing.assert_array_equal(G[b].view(np.uint64),
                                          O.discounted_return(r[:, b].astype(np.float64), gamma).view(np.uint64))
    with pytest.raises(bpg._lib.BpgError):
        bpg.discounted_return(torch.zeros((320, 4), device=DEV), gamma)
    for T in (50, 33):
        r = torch.as_tensor(rng.normal(size=(T, 256)).astype(np.float32), device=DEV)
        out = torch.empty((256, T), dtype=torch.float64, device=DEV)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            bpg.discounted_return(r, gamma, out=out)                # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        expect = out.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            bpg.discounted_return(r, gamma, out=out)
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, expect)


def test_compute_reward_long_rows():
    """len up to 256 (the header's range): rows longer than 159 floats use 64-row tiles."""
    rng = np.random.RandomState(8)
    for n, ln in [(300, 159), (300, 160), (257, 200), (130, 256), (64, 255)]:
        ag = rng.randint(-1, 2, size=(n, ln)).astype(np.float32)
        g = (rng.randint(-1, 2, size=(n, ln)) * (rng.rand(n, ln) < 0.1)).astype(np.float32)
        ag[::2] = np.where(g[::2] != 0, g[::2], ag[::2])
        got = bpg.compute_reward(torch.as_tensor(ag, device=DEV), torch.as_tensor(g, device=DEV))
        np.testing.assert_array_equal(bits(got), O.compute_reward(ag, g).view(np.uint32))
    with pytest.raises(bpg._lib.BpgError):
        bpg.compute_reward(torch.zeros((4, 257), device=DEV), torch.zeros((4, 257), device=DEV))


def test_compute_reward_with_one_goal_row():
    """env.compute_reward(ag, env.goal): ONE goal row against every row of ag (NumPy broadcasting in
    fetch_env.py:141-143) goes through bpg_compute_reward_f32_goal -- no materialised (n, len) goal.
    Every variant's row length (9, 16, 25, 36: float4 and scalar staging), ragged tile tails, leading
    axes, numpy and torch inputs; == the row-by-row kernel and the oracle."""
    rng = np.random.RandomState(12)
    for env_id in ENV_IDS:
        v = VARIANTS[env_id]
        env = bpg.make(env_id, num_envs=2, device=DEV)
        goal = np.asarray(env.goal, np.float64)
        ln = goal.shape[0]
        for lead in [(1,), (127,), (129,), (5, 51), (1000,)]:
            ag = rng.randint(-1, 2, size=lead + (ln,)).astype(np.float32)
            hit = rng.rand(*lead) < 0.3                                      # rows that satisfy the goal
            ag[hit] = np.where(goal != 0, goal, ag[hit]).astype(np.float32)
            exp = O.compute_reward(ag.reshape(-1, ln), goal).reshape(lead)
            got_np = env.compute_reward(ag, goal, {})                         # numpy in -> numpy out, 1-D goal
            np.testing.assert_array_equal(got_np.view(np.uint32), exp.view(np.uint32))
            tag = torch.as_tensor(ag, device=DEV)
            tg = torch.as_tensor(goal.astype(np.float32), device=DEV)
            for gform in (tg, tg[None], tg.expand(*lead, ln), tg.expand(*lead, ln).contiguous()):
                got = bpg.compute_reward(tag, gform)
                np.testing.assert_array_equal(bits(got).reshape(lead), exp.view(np.uint32))
            assert (exp == 0).any() and (exp != 0).any() or lead == (1,)
    # keyword spelling of the reference's call site (config.py:110-111)
    r = env.compute_reward(achieved_goal=ag, desired_goal=goal, info={})
    np.testing.assert_array_equal(r.view(np.uint32), exp.view(np.uint32))


# --------------------------------------------------------------------------
# seeding contract (build-defined replacement of seed + 1000 * idx)
# --------------------------------------------------------------------------
def test_seeding_contract():
    """robot_env.py:53-55 / rollout.py:206-210 give env idx the stream seed + 1000*idx of gym's
    hashed MT19937 (third-party).  The replacement contract: every generator-drawn quantity is a
    pure function of (seed, GLOBAL env index, stream index, step) --
      * seed() returns [seed] and restarts the stream indices: same seed => same draws again;
      * different seeds / different env indices give different, uncorrelated draws;
      * each generator-drawn reset / placement advances its stream index by one: consecutive
        episodes differ, and a state_dict round trip resumes the streams;
      * how the batch is split over handles (shards) does not matter."""
    env_id, B = 'BlocksTouchVariation-v0', 4096
    L = bpg._lib.lib()

    def draws(env):
        env.reset()
        ng = env.num_objs_per_env().clone()
        m = torch.empty((4, env.num_envs), dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_synth_masks(env._h, 0, 4, C.c_void_p(m.data_ptr()), None))
        xy, _ = env.randomize_objects()
        return ng, m, xy.nan_to_num()

    env = bpg.make(env_id, num_envs=B, device=DEV, seed=42)
    assert env.seed(42) == [42]
    first = draws(env)
    second = draws(env)                                              # next episode: stream indices advanced
    assert not torch.equal(first[0], second[0]) and not torch.equal(first[2], second[2])
    assert int(env.state_counters[0]) == 2 and int(env.state_counters[1]) == 2
    sd = env.state_dict()
    third = draws(env)
    env.load_state_dict(sd)
    again = draws(env)
    assert all(torch.equal(x, y) for x, y in zip(third, again))      # checkpoint resumes the streams
    env.seed(42)
    assert int(env.state_counters[0]) == 0
    replay = draws(env)
    assert all(torch.equal(x, y) for x, y in zip(first, replay))     # same seed: same draws
    other = draws(bpg.make(env_id, num_envs=B, device=DEV, seed=43))
    assert (first[0] != other[0]).float().mean() > 0.5 and (first[1] != other[1]).float().mean() > 0.9
    lo = draws(bpg.make(env_id, num_envs=B // 2, device=DEV, seed=42))
    hi = draws(bpg.make(env_id, num_envs=B // 2, device=DEV, seed=42, env_id_base=B // 2))
    for k in range(3):
        assert torch.equal(torch.cat([lo[k], hi[k]], 0 if k != 1 else 1), first[k])


# --------------------------------------------------------------------------
# ragged (CSR) contact lists
# --------------------------------------------------------------------------
@pytest.mark.parametrize('env_id', ['BlocksTouchVariation-v0', 'ToppleTower-v0', 'GripperTouch-v0'])
def test_ragged_contact_lists_vs_oracle(env_id):
    """sim.data.contact of all envs compacted into one (total, 2) geom-id array with CSR offsets
    (0..40 contacts per env, many envs with none) == the oracle's per-env list form, step after
    step; a malformed offset table is clamped instead of indexing outside the array."""
    v = VARIANTS[env_id]
    B, T = 500, 12
    rng = np.random.RandomState(7)
    env = bpg.make(env_id, num_envs=B, device=DEV)
    oenvs = [O.OracleEnv(env_id) for _ in range(B)]
    num_grey = rng.randint(3, size=B).astype(np.int32)
    env.reset(num_grey=num_grey)
    for e, g in zip(oenvs, num_grey):
        e.reset(num_grey=int(g))
    model = num_grey if 'Variation' in env_id else np.zeros(B, np.int32)
    ngeom = np.array([len(v['models'][m]['id2obj']) for m in model])
    for t in range(T):
        ncon = rng.randint(0, 41, size=B) * (rng.rand(B) < 0.6)
        offsets = np.concatenate([[0], np.cumsum(ncon)]).astype(np.int64)
        total = int(offsets[-1])
        owner = np.repeat(np.arange(B), ncon.astype(np.int64))
        pairs = (rng.randint(0, 1 << 30, size=(total, 2)) % ngeom[owner][:, None]).astype(np.int16)
        obs, r, done, info = env.step(None, contacts=(pairs, offsets))
        exp = [e.step(v['models'][int(m)]['id2obj'], pairs[offsets[b]:offsets[b + 1]])
               for b, (e, m) in enumerate(zip(oenvs, model))]
        np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), np.stack([x[0] for x in exp]).astype(np.float32))
        np.testing.assert_array_equal(bits(r), np.array([np.float32(x[1]).view(np.uint32) for x in exp]))
        np.testing.assert_array_equal(info['is_success'].cpu().numpy(), np.array([x[3] for x in exp]))
    # the CSR launch == the padded launch on the same lists
    L = bpg._lib.lib()
    maxcon = 40
    pad = np.zeros((B, maxcon, 2), np.int16)
    for b in range(B):
        pad[b, :offsets[b + 1] - offsets[b]] = pairs[offsets[b]:offsets[b + 1]]
    m1 = torch.empty(B, dtype=torch.int32, device=DEV)
    m2 = torch.empty(B, dtype=torch.int32, device=DEV)
    dp, do = torch.as_tensor(pairs, device=DEV), torch.as_tensor(offsets, device=DEV)
    dpad, dn = torch.as_tensor(pad, device=DEV), torch.as_tensor(np.diff(offsets).astype(np.int32), device=DEV)
    bpg._lib.check(L.bpg_contacts_csr_to_masks(env._h, C.c_void_p(dp.data_ptr()), C.c_void_p(do.data_ptr()), total,
                                               C.c_void_p(m1.data_ptr()), None))
    bpg._lib.check(L.bpg_contacts_to_masks(env._h, C.c_void_p(dpad.data_ptr()), C.c_void_p(dn.data_ptr()), maxcon,
                                           C.c_void_p(m2.data_ptr()), None))
    assert torch.equal(m1, m2)
    # offsets beyond the array / decreasing: clamped, no fault
    bad = offsets.copy(); bad[B // 2] = -5; bad[B // 2 + 1] = total + 1000; bad[-1] = total + 7
    dbad = torch.as_tensor(bad, device=DEV)
    bpg._lib.check(L.bpg_contacts_csr_to_masks(env._h, C.c_void_p(dp.data_ptr()), C.c_void_p(dbad.data_ptr()), total,
                                               C.c_void_p(m1.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(m1[:B // 2 - 1], m2[:B // 2 - 1])


# --------------------------------------------------------------------------
# a physics backend behind the boundary, end to end
# --------------------------------------------------------------------------
def test_toy_physics_backend_end_to_end():
    """examples/toy_physics_rollout.py: reset -> placement -> on_reset -> scripted policy reading the
    assembled observation -> set_action_vector -> toy kinematics -> geom-id contact lists -> touch
    matrix / success.  The pusher solves BlocksTouch-v0 with the toy backend; an idle policy does not."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('toy_physics_rollout', os.path.join(root, 'examples', 'toy_physics_rollout.py'))
    toy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(toy)
    B = 8192
    worker, ep, phys = toy.run(B, device=DEV, seed=3)
    ag = ep['ag'].reshape(B, 51, 4, 4)
    assert (ag[:, 0, 2, 3] == -1).all()                              # placed >= 7.5 cm apart: never touched yet
    assert (ep['info_is_success'][:, 0, 0] == 0).all()
    assert worker.success_history[-1] > 0.95
    first = (ag[:, :, 2, 3] == 1).float().argmax(1)                  # the push takes a few steps
    assert first[ep['info_is_success'][:, -1, 0] == 1].float().mean() > 4
    assert (ag[:, 1:, 1, 2] == 1).all() and (ag[:, 1:, 1, 3] == 1).all()   # cubes rest on the table
    o = ep['o']
    # observation slots hold what the backend assembled: block positions move only for the pushed cube
    moved = (o[:, -1, 10:12] - o[:, 0, 10:12]).norm(dim=1)
    assert (moved[ep['info_is_success'][:, -1, 0] == 1] > 0.01).all()

    class Idle:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return torch.zeros(o.shape[0], 4, device=o.device)
    worker.policy = Idle()
    worker.generate_rollouts()
    assert worker.success_history[-1] == 0.0


# --------------------------------------------------------------------------
# padded rows of the time-major streams (bpg_row_stride, ld_masks / ld_out)
# --------------------------------------------------------------------------
def test_row_stride_helper():
    L = bpg._lib.lib()
    for n in (1, 255, 65536, 65537, (1 << 20) - 1):
        assert L.bpg_row_stride(n) == n                 # no shift below 2^20 envs
    for n in (1 << 20, (1 << 20) + 1, 3 << 19, (1 << 22) + 2, 1 << 25):
        ld = L.bpg_row_stride(n)
        assert ld >= n and ld % 256 == 0 and (ld // 256) % 2 == 1 and ld - n < 512
    assert L.bpg_row_stride(0) == 0


@pytest.mark.parametrize('B,pad', [(65536, 256), (131072 + 2, 254), (524288, 256), (4096, 3), (1 << 20, 256)])
def test_padded_time_major_streams_equal_contiguous_ones(B, pad):
    """Every consumer / producer of the (T, B) streams with rows ld = B + pad elements apart (even and odd
    ld: the odd one falls back to one env per lane) == the contiguous form: synthetic masks, the packed
    fused rollout, the episode writer, the discounted return (register and generic kernels)."""
    env_id, T, L = 'BlocksTouchVariation-v0', 50, bpg._lib.lib()
    ld = B + pad
    P = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    a, b = (bpg.make(env_id, num_envs=B, device=DEV, seed=21) for _ in range(2))
    ng = torch.as_tensor(np.random.RandomState(B % 997).randint(3, size=B).astype(np.int32), device=DEV)
    for e in (a, b):
        e.reset(num_grey=ng)
    m0 = torch.empty((T, B), dtype=torch.int32, device=DEV)
    m1 = torch.full((T, ld), -1, dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks(a._h, 0, T, P(m0), None))
    bpg._lib.check(L.bpg_synth_masks_ld(a._h, 0, T, P(m1), ld, None))
    assert torch.equal(m1[:, :B], m0) and bool((m1[:, B:] == -1).all())           # the pad is never written

    r0 = torch.empty((T, B), dtype=torch.float32, device=DEV)
    f0 = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    r1 = torch.full((T, ld), 7.0, dtype=torch.float32, device=DEV)
    f1 = torch.full((T, ld), 0xEE, dtype=torch.uint8, device=DEV)
    c0, c1 = (torch.zeros(2, dtype=torch.int64, device=DEV) for _ in range(2))
    for reset_first in (1, 0):
        a0 = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_num_grey=ng.data_ptr(), d_masks=m0.data_ptr(),
                                  d_reward_tb=r0.data_ptr(), d_flags_tb=f0.data_ptr(), d_success_count=c0.data_ptr())
        a1 = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_num_grey=ng.data_ptr(), d_masks=m1.data_ptr(),
                                  d_reward_tb=r1.data_ptr(), d_flags_tb=f1.data_ptr(), d_success_count=c1.data_ptr(),
                                  ld_masks=ld, ld_out=ld)
        bpg._lib.check(L.bpg_rollout_v2(a._h, C.byref(a0), None))
        bpg._lib.check(L.bpg_rollout_v2(b._h, C.byref(a1), None))
        assert torch.equal(r1[:, :B].view(torch.int32), r0.view(torch.int32)) and torch.equal(f1[:, :B], f0)
        assert bool((r1[:, B:] == 7.0).all()) and bool((f1[:, B:] == 0xEE).all())
        assert torch.equal(a.state_now, b.state_now) and torch.equal(a.state_ever, b.state_ever)
        assert int(c0[0]) == int(c1[0])
    # discounted return over the padded reward stream: T = 50 (register kernel) and T = 7 (generic)
    for Tr in (50, 7):
        G0 = torch.empty((B, Tr), dtype=torch.float64, device=DEV)
        G1 = torch.empty((B, Tr), dtype=torch.float64, device=DEV)
        bpg._lib.check(L.bpg_discounted_return(P(r0), B, Tr, 0.98, P(G0), 0, None))
        bpg._lib.check(L.bpg_discounted_return_ld(P(r1), B, ld, Tr, 0.98, P(G1), 0, None))
        assert torch.equal(G0.view(torch.int64), G1.view(torch.int64))
    assert torch.equal(bpg.discounted_return(r1[:, :B], 0.98).view(torch.int64),
                       bpg.discounted_return(r0, 0.98).view(torch.int64))
    # mixed: padded masks in, contiguous out
    r2 = torch.empty((T, B), dtype=torch.float32, device=DEV)
    a2 = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m1.data_ptr(),
                              d_reward_tb=r2.data_ptr(), ld_masks=ld)
    a0 = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m0.data_ptr(), d_reward_tb=r0.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(a._h, C.byref(a0), None))
    bpg._lib.check(L.bpg_rollout_v2(b._h, C.byref(a2), None))
    assert torch.equal(r2.view(torch.int32), r0.view(torch.int32))
    # the episode writer reading padded masks
    if B <= 131074:
        def episode(env, masks, ldm):
            ag = torch.empty((B, T + 1, 36), dtype=torch.float32, device=DEV)
            sc = torch.empty((B, T, 1), dtype=torch.float32, device=DEV)
            rw = torch.full((T, ld), 3.0, dtype=torch.float32, device=DEV)
            args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(),
                                        d_ep_ag=ag.data_ptr(), d_ep_success=sc.data_ptr(), d_reward_tb=rw.data_ptr(),
                                        ld_masks=ldm, ld_out=ld)
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(args), None))
            return ag, sc, rw
        ag0, sc0, rw0 = episode(a, m0, 0)
        ag1, sc1, rw1 = episode(b, m1, ld)
        assert torch.equal(ag0, ag1) and torch.equal(sc0, sc1)
        assert torch.equal(rw0[:, :B].view(torch.int32), r0.view(torch.int32)) and torch.equal(rw0, rw1)
    # a stride below the batch size is refused
    bad = bpg._lib.RolloutArgs(T=T, d_masks=m0.data_ptr(), ld_masks=B - 2)
    assert L.bpg_rollout_v2(a._h, C.byref(bad), None) != 0
    assert L.bpg_synth_masks_ld(a._h, 0, 1, P(m0), B - 1, None) != 0
    assert L.bpg_discounted_return_ld(P(r0), B, B - 1, 50, 0.98, P(G0), 0, None) != 0


def test_worker_streams_use_the_recommended_row_stride():
    B, T = 1 << 16, 50
    t = bpg.time_major_empty(T, 1 << 20, torch.float32, DEV)
    assert t.shape == (T, 1 << 20) and t.stride() == ((1 << 20) + 256, 1)
    del t
    w = bpg.RolloutStudent(lambda num_envs: bpg.make('BlocksTouchVariation-v0', num_envs=num_envs, device=DEV, seed=2),
                           None, dims_of('BlocksTouchVariation-v0'), None, T, rollout_batch_size=B, output='torch')
    env = w.env
    masks = torch.empty((T, B + 256), dtype=torch.int32, device=DEV)[:, :B]      # rows of the caller's choosing
    bpg._lib.check(bpg._lib.lib().bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(masks.data_ptr()), masks.stride(0), None))
    # the same explicit block counts for both rollouts: the masks were generated for the episode that was current
    # then, and a pair of an object the NEXT episode does not have would (as in the reference, fetch_env.py:154-157)
    # stay set -- the two rollouts would differ for a reason that has nothing to do with the row stride
    ng = env.num_objs_per_env().to(torch.int32) - 4
    ep1 = w.generate_rollouts_fused(masks=masks, want_reward=True, num_grey=ng)
    ep2 = w.generate_rollouts_fused(masks=masks.contiguous(), want_reward=True, num_grey=ng)
    assert ep1['reward_tb'].stride(0) == bpg._lib.lib().bpg_row_stride(B)
    assert torch.equal(ep1['reward_tb'].view(torch.int32), ep2['reward_tb'].view(torch.int32))
    assert torch.equal(ep1['ag'], ep2['ag'])
    G = bpg.discounted_return(ep1['reward_tb'], 0.98)
    assert torch.equal(G.view(torch.int64), bpg.discounted_return(ep2['reward_tb'].contiguous(), 0.98).view(torch.int64))
