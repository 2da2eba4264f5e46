"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs, against the golden fixtures recorded from the reference, and
-- at BASELINE.json's full size -- through size-independent properties.
Everything here is bit-exact (integer / bitmask work; rewards compared as bits)."""
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from oracle import bpg_oracle as O
from util import GOLDEN, VARIANTS, ENV_IDS, short, legal_masks, fixture_masks

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def i32(a):
    return torch.as_tensor(np.ascontiguousarray(a).view(np.int32), device=DEV)


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def dims_of(env_id):
    v = VARIANTS[env_id]
    return dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)


# --------------------------------------------------------------------------
# golden fixtures recorded from the reference
# --------------------------------------------------------------------------
@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_step_traces_scalar_api(env_id):
    """Reference env.step()/reset() traces replayed through the drop-in single-env
    API with the contacts exactly as the reference read them (geom-id pairs)."""
    v = VARIANTS[env_id]
    d = np.load(os.path.join(GOLDEN, 'trace_%s.npz' % short(env_id)))
    env = bpg.make(env_id, device=DEV)
    assert env._max_episode_steps == 50
    for ep, L in enumerate(d['lens']):
        obs = env.reset(num_grey=[int(d['num_objs'][ep]) - 4])
        if d['test_mode'][ep]:
            obs = env.unwrapped.set_test()
        assert obs['observation'].shape == (v['dimo'],) and obs['observation'].dtype == np.float64
        assert obs['achieved_goal'].dtype == np.float64
        assert str(obs['desired_goal'].dtype) == v['desired_goal_dtype']
        np.testing.assert_array_equal(obs['achieved_goal'], d['ag_reset'][ep])
        np.testing.assert_array_equal(obs['desired_goal'], d['g'][ep])
        for t in range(L):
            nc = int(d['ncon'][ep, t])
            pairs = d['contacts'][ep, t][None].copy()
            pairs[pairs < 0] = 0
            obs, r, done, info = env.step(d['actions'][ep, t], contacts=(pairs, np.array([nc], np.int32)))
            np.testing.assert_array_equal(obs['achieved_goal'], d['ag'][ep, t])
            assert isinstance(r, np.float32) and r.view(np.uint32) == d['reward_bits'][ep, t]
            assert done is bool(d['done'][ep, t])
            assert info['is_success'] is bool(d['success'][ep, t])


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_rollout_student(env_id):
    """Reference RolloutStudent.generate_rollouts episodes (rollout.py:75-172)
    over consecutive rollouts, policy in the loop, contacts as geom-id pairs."""
    d = np.load(os.path.join(GOLDEN, 'rollout_%s.npz' % short(env_id)))
    B, T = int(d['B']), int(d['T'])

    class Replay:
        def __init__(self):
            self.u, self.t = None, 0

        def get_actions(self, o, ag, g, **kw):
            assert o.shape == (B, VARIANTS[env_id]['dimo']) and o.dtype == np.float32
            assert set(kw) == {'compute_Q', 'noise_eps', 'random_eps', 'use_target_net'}
            u = self.u[:, self.t]
            self.t += 1
            return u

    policy = Replay()
    src = bpg.ScriptedContacts()
    worker = bpg.RolloutStudent(lambda: bpg.make(env_id, device=DEV, contact_source=src), policy,
                                dims_of(env_id), None, T, rollout_batch_size=B)
    worker.env.contact_source = src
    worker.seed(123)
    for r, test in enumerate(d['modes']):
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        src.pairs = np.ascontiguousarray(con.transpose(1, 0, 2, 3))       # (T, B, maxcon, 2)
        src.ncon = np.ascontiguousarray(d['r%d_ncon' % r].T)
        src.masks, src.t = None, 0
        policy.u, policy.t = d['r%d_u' % r], 0
        ep = worker.generate_rollouts(test=bool(test), num_grey=d['r%d_num_objs' % r] - 4)
        for k in ('ag', 'g', 'u', 'info_is_success'):
            assert ep[k].dtype == np.float32
            np.testing.assert_array_equal(ep[k], d['r%d_%s' % (r, k)], err_msg=k)
        assert ep['o'].shape == d['r%d_o' % r].shape
        assert worker.success_history[-1] == pytest.approx(float(d['r%d_success_rate' % r]), abs=1e-7)
    assert worker.n_episodes == int(d['n_episodes'])


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_rollout_fused(env_id):
    """Same reference episodes through the single-launch fused rollout kernel."""
    d = np.load(os.path.join(GOLDEN, 'rollout_%s.npz' % short(env_id)))
    B, T = int(d['B']), int(d['T'])
    worker = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None,
                                dims_of(env_id), None, T, rollout_batch_size=B)
    pinned_by_reference = []
    for r, test in enumerate(d['modes']):
        # the reference's contact lists (geom-id pairs of every step) -> the mask stream, on the device, in one launch
        # (env.contacts_to_masks: the model of each env is the one of the episode about to be drawn); the oracle's
        # host-side conversion is only the cross-check
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        num_grey = d['r%d_num_objs' % r] - 4
        dmasks = worker.env.contacts_to_masks(np.ascontiguousarray(con.transpose(1, 0, 2, 3)),
                                              np.ascontiguousarray(d['r%d_ncon' % r].T), num_grey=num_grey)
        masks = fixture_masks(env_id, d['r%d_contacts' % r], d['r%d_ncon' % r], d['r%d_model_idx' % r])
        np.testing.assert_array_equal(dmasks.cpu().numpy().view(np.uint32), masks)
        # Variation picks the sim -- and with it the id2obj table -- of an episode by num_objs = 4 + num_grey
        # (fetch_env.py:647-651), and the conversion runs BEFORE the rollout's reset: it must take the table of the episode
        # that is ABOUT to be drawn (d_num_grey), not the one of the state.  The reference pins that choice: with the table of
        # the previous episode's state the masks differ wherever the block count changed and a contact involves a geom
        # whose id moved, and the episode below would not be the reference's.
        if env_id.startswith('BlocksTouchVariation'):
            changed = (worker.env.num_objs_per_env().cpu().numpy() - 4) != num_grey
            stale = worker.env.contacts_to_masks(np.ascontiguousarray(con.transpose(1, 0, 2, 3)),
                                                 np.ascontiguousarray(d['r%d_ncon' % r].T), num_grey=None)
            same = (stale.cpu().numpy().view(np.uint32) == masks).all(0)
            assert same[~changed].all()                       # same model: same masks
            if changed.any():
                pinned_by_reference.append(bool((~same[changed]).any()))
        ep = worker.generate_rollouts_fused(test=bool(test), masks=dmasks, num_grey=num_grey)
        for k in ('ag', 'g', 'info_is_success'):
            np.testing.assert_array_equal(ep[k], d['r%d_%s' % (r, k)], err_msg=k)
        assert worker.success_history[-1] == pytest.approx(float(d['r%d_success_rate' % r]), abs=1e-7)
    if env_id.startswith('BlocksTouchVariation'):
        assert any(pinned_by_reference)          # the fixture does exercise a table switch that changes the masks


def test_golden_compute_reward():
    d = np.load(os.path.join(GOLDEN, 'compute_reward.npz'))
    keys = sorted({k.rsplit('_', 1)[0] for k in d.files})
    env = bpg.make('BlocksTouch-v0', device=DEV)
    for k in keys:
        ag, g, g2 = d[k + '_ag'], d[k + '_g'], d[k + '_g2']
        r = env.compute_reward(ag, g, None)
        assert r.dtype == np.float32 and r.shape == (257,)
        np.testing.assert_array_equal(r.view(np.uint32), d[k + '_rbits'])
        # config.py:110-111 keyword spelling
        r2 = env.compute_reward(achieved_goal=ag, desired_goal=g2, info=None)
        np.testing.assert_array_equal(r2.view(np.uint32), d[k + '_rbits2'])
        n2 = ag.shape[1]
        r3 = env.compute_reward(ag[:240].reshape(4, 60, n2).astype(np.float64),
                                g2[:240].reshape(4, 60, n2).astype(np.float64), None)
        assert r3.shape == (4, 60)
        np.testing.assert_array_equal(r3.view(np.uint32), d[k + '_rbits3'])


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation', 'ToppleTower', 'GripperTouch', 'BlocksTouchChoose'])
def test_golden_pg_rollout_return(name):
    """policy_gradient/rollout.py: reference G (float64) bit-for-bit."""
    env_id = name + '-v0'
    d = np.load(os.path.join(GOLDEN, 'pg_rollout_%s.npz' % name))
    B, T = int(d['B']), int(d['T'])

    class Replay:
        kwargs = {}

        def get_actions(self, o, ag, g, exploit=False):
            u = self.u[:, self.t]
            self.t += 1
            return u, u.copy(), np.zeros_like(u)

    policy = Replay()
    src = bpg.ScriptedContacts()
    worker = bpg.PGRolloutStudent(lambda: bpg.make(env_id, device=DEV), policy, None, dims_of(env_id), None, T,
                                  rollout_batch_size=B, gamma=1. - 1. / T)
    worker.env.contact_source = src
    for r in range(len(d['modes'])):
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        src.pairs = np.ascontiguousarray(con.transpose(1, 0, 2, 3))
        src.ncon = np.ascontiguousarray(d['r%d_ncon' % r].T)
        src.t = 0
        policy.u, policy.t = d['r%d_u' % r], 0
        ep = worker.generate_rollouts(num_grey=d['r%d_num_objs' % r] - 4)
        assert ep['G'].dtype == np.float64
        np.testing.assert_array_equal(ep['G'].view(np.uint64), d['r%d_G' % r].view(np.uint64))
        np.testing.assert_array_equal(ep['ag'], d['r%d_ag' % r])
        np.testing.assert_array_equal(ep['info_is_success'], d['r%d_info_is_success' % r])


# --------------------------------------------------------------------------
# oracle on the same seeded inputs
# --------------------------------------------------------------------------
@pytest.mark.parametrize('env_id', ENV_IDS)
@pytest.mark.parametrize('B', [4096, 1003, 1])
def test_fused_rollout_vs_oracle(env_id, B):
    """Random legal contact masks, two consecutive rollouts (quirk Q1 carries the
    matrix over for the base variants), vector (B%4==0) and scalar kernels."""
    T = 50
    rng = np.random.RandomState(B)
    worker = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None,
                                dims_of(env_id), None, T, rollout_batch_size=B)
    envs = None
    for r in range(2):
        num_grey = rng.randint(3, size=B).astype(np.int32)
        nobj = 4 + num_grey if 'Variation' in env_id else VARIANTS[env_id]['num_objs']
        masks = legal_masks(rng, T, B, nobj, density=2 + r)
        ep = worker.generate_rollouts_fused(masks=i32(masks), num_grey=num_grey, want_reward=True)
        ref, envs = O.generate_rollouts(env_id, B, T, masks, num_grey=num_grey, envs=envs)
        np.testing.assert_array_equal(ep['ag'], ref['ag'])
        np.testing.assert_array_equal(ep['g'], ref['g'])
        np.testing.assert_array_equal(ep['info_is_success'], ref['info_is_success'])
        np.testing.assert_array_equal(ep['reward_tb'].view(np.uint32), ref['reward'].view(np.uint32))
        assert worker.success_history[-1] == pytest.approx(ref['success_rate'], abs=1e-6)


@pytest.mark.parametrize('env_id', ['GripperTouch-v0', 'BlocksTouchChoose-v0', 'BlocksTouchVariation-v0',
                                    'ToppleTower-v0'])
@pytest.mark.parametrize('B', [2048, 777])
def test_packed_rollout_and_steps_vs_oracle(env_id, B):
    """The packed fused rollout (no episode buffers), the per-step mask kernel
    and the contact-list kernel all agree with the oracle: reward bits, flags
    (is_success | done<<1) and the final touch matrix.  T=60 crosses the
    TimeLimit (done stays set)."""
    T = 60
    rng = np.random.RandomState(7 * B)
    v = VARIANTS[env_id]
    num_grey = rng.randint(3, size=B).astype(np.int32)
    nobj = 4 + num_grey if 'Variation' in env_id else np.full(B, v['num_objs'])
    masks = legal_masks(rng, T, B, nobj)
    # oracle
    oenvs = [O.OracleEnv(env_id) for _ in range(B)]
    exp_r = np.zeros((T, B), np.uint32)
    exp_f = np.zeros((T, B), np.uint8)
    for b, e in enumerate(oenvs):
        e.reset(num_grey=int(num_grey[b]))
        for t in range(T):
            _, r, done, succ = e.step_mask(int(masks[t, b]))
            exp_r[t, b] = np.float32(r).view(np.uint32)
            exp_f[t, b] = int(succ) | (int(done) << 1) | (int(r == 0) << 2)
    exp_ag = np.stack([e.achieved_goal() for e in oenvs])

    L = bpg._lib.lib()
    import ctypes as C
    # (1) per-step mask kernel through the gym-style API
    env = bpg.make(env_id, num_envs=B, device=DEV)
    env.reset(num_grey=num_grey)
    dm = i32(masks)
    for t in range(T):
        obs, r, done, info = env.step(None, contact_masks=dm[t])
        np.testing.assert_array_equal(bits(r), exp_r[t])
        f = info['is_success'].cpu().numpy().astype(np.uint8) | (done.cpu().numpy().astype(np.uint8) << 1)
        np.testing.assert_array_equal(f, exp_f[t] & 3)
    np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), exp_ag.astype(np.float32))
    np.testing.assert_array_equal(env.achieved_goal_f64().cpu().numpy(), exp_ag)
    assert int(env.success_count().item()) == int((exp_f[-1] & 1).sum())
    # (2) packed fused rollout
    env2 = bpg.make(env_id, num_envs=B, device=DEV)
    env2.reset(num_grey=num_grey)
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flg = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, d_masks=dm.data_ptr(), d_reward_tb=rew.data_ptr(), d_flags_tb=flg.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(env2._h, C.byref(args), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(bits(rew), exp_r)
    np.testing.assert_array_equal(flg.cpu().numpy(), exp_f)
    assert torch.equal(env2.state_now, env.state_now) and torch.equal(env2.state_ever, env.state_ever)
    # (3) contact-list kernel: rebuild geom-id lists from the masks
    env3 = bpg.make(env_id, num_envs=B, device=DEV)
    env3.reset(num_grey=num_grey)
    pair_of = {L.bpg_pair_index(i, j): (i, j) for i in range(6) for j in range(i, 6)}
    geom_of = []                                   # per model: object id -> a geom id
    for m in v['models']:
        geom_of.append({o: g for g, o in enumerate(m['id2obj']) if o >= 0})
    none_geom = [m['id2obj'].index(-1) for m in v['models']]
    for t in range(0, T, 7):
        maxcon = 24 if (t // 7) % 2 == 0 else 23      # LDS-staged (maxcon % 4 == 0) and direct kernels
        pairs = np.zeros((B, maxcon, 2), np.int16)
        ncon = np.zeros(B, np.int32)
        for b in range(B):
            mi = int(num_grey[b]) if 'Variation' in env_id else 0
            lst = [(none_geom[mi], geom_of[mi][1])]                 # a None-mapped contact is ignored
            for p in range(21):
                if masks[t, b] >> p & 1:
                    i, j = pair_of[p]
                    lst.append((geom_of[mi][j], geom_of[mi][i]) if (p + b) % 2 else (geom_of[mi][i], geom_of[mi][j]))
            lst.append(lst[-1])                                      # duplicates are harmless
            pairs[b, :len(lst)] = lst
            ncon[b] = len(lst)
        env3.load_state_dict(env.state_dict())
        e_ref = bpg.make(env_id, num_envs=B, device=DEV)
        e_ref.load_state_dict(env.state_dict())
        _, r_a, d_a, i_a = env3.step(None, contacts=(pairs, ncon))
        _, r_b, d_b, i_b = e_ref.step(None, contact_masks=dm[t])
        assert torch.equal(r_a.view(torch.int32), r_b.view(torch.int32))
        assert torch.equal(env3.state_now, e_ref.state_now) and torch.equal(env3.state_ever, e_ref.state_ever)


@pytest.mark.parametrize('env_id', ['BlocksTouchVariation-v0', 'BlocksTouch-v0'])
@pytest.mark.parametrize('B', [8192, 1001])
def test_single_launch_rollout_reset_and_count(env_id, B):
    """reset_first + d_success_count inside the rollout launch == bpg_reset, then
    bpg_rollout, then bpg_success_count; ping-pong counter zeroing."""
    import ctypes as C
    T = 50
    L = bpg._lib.lib()
    rng = np.random.RandomState(B)
    num_grey = torch.as_tensor(rng.randint(3, size=B).astype(np.int32), device=DEV)
    nobj = 4 + num_grey.cpu().numpy() if 'Variation' in env_id else VARIANTS[env_id]['num_objs']
    a, b = [bpg.make(env_id, num_envs=B, device=DEV, seed=3) for _ in range(2)]
    counts = torch.tensor([0, 12345], dtype=torch.int64, device=DEV)
    for r in range(3):
        dm = i32(legal_masks(rng, T, B, nobj, density=4))
        rew_a = torch.empty((T, B), dtype=torch.float32, device=DEV)
        rew_b = torch.empty_like(rew_a)
        cur = r & 1
        args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=num_grey.data_ptr(), d_masks=dm.data_ptr(),
                                    d_reward_tb=rew_a.data_ptr(),
                                    d_success_count=counts.data_ptr() + 8 * cur,
                                    d_zero_count=counts.data_ptr() + 8 * (1 - cur))
        bpg._lib.check(L.bpg_rollout_v2(a._h, C.byref(args), None))
        b.reset(num_grey=num_grey)
        args_b = bpg._lib.RolloutArgs(T=T, d_masks=dm.data_ptr(), d_reward_tb=rew_b.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(b._h, C.byref(args_b), None))
        assert torch.equal(rew_a.view(torch.int32), rew_b.view(torch.int32))
        assert torch.equal(a.state_now, b.state_now) and torch.equal(a.state_ever, b.state_ever)
        assert int(counts[cur].item()) == int(b.success_count().item()) == int((b.state_ever < 0).sum().item())
        assert int(counts[1 - cur].item()) == 0
    # the episode-writer kernel takes the same fused reset / count path
    ag = torch.empty((B, T + 1, a.dimg), dtype=torch.float32, device=DEV)
    counts.zero_()
    args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=num_grey.data_ptr(), d_masks=dm.data_ptr(),
                                d_ep_ag=ag.data_ptr(), d_success_count=counts.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(a._h, C.byref(args), None))
    b.reset(num_grey=num_grey)
    bpg._lib.check(L.bpg_rollout_v2(b._h, C.byref(args_b), None))
    assert torch.equal(a.state_ever, b.state_ever) and int(counts[0].item()) == int(b.success_count().item())
    assert torch.equal(ag[:, T], b._get_obs()['achieved_goal'])


def test_synthetic_source_matches_oracle_spec():
    """bpg_synth_masks / in-kernel synthetic contacts == the oracle's statement of
    the (build-defined) generator, including the global env index offset."""
    import ctypes as C
    B, T, base, seed = 3000, 12, 123456, 99
    for env_id in ['BlocksTouchVariation-v0', 'BlocksTouch-v0']:
        env = bpg.make(env_id, num_envs=B, device=DEV, seed=seed, env_id_base=base)
        rng = np.random.RandomState(5)
        num_grey = rng.randint(3, size=B).astype(np.int32)
        env.reset(num_grey=num_grey)
        nobj = env.num_objs_per_env().cpu().numpy()
        out = torch.empty((T, B), dtype=torch.int32, device=DEV)
        bpg._lib.check(bpg._lib.lib().bpg_synth_masks(env._h, 0, T, C.c_void_p(out.data_ptr()), None))
        exp = O.synth_masks(seed, base + np.arange(B), T, nobj)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), exp)
        # stepping with the in-kernel source == stepping with those masks
        env_b = bpg.make(env_id, num_envs=B, device=DEV, seed=seed, env_id_base=base)
        env_b.reset(num_grey=num_grey)
        for t in range(T):
            _, r1, d1, i1 = env.step(None)
            _, r2, d2, i2 = env_b.step(None, contact_masks=out[t])
            assert torch.equal(r1.view(torch.int32), r2.view(torch.int32))
        assert torch.equal(env.state_now, env_b.state_now) and torch.equal(env.state_ever, env_b.state_ever)


def test_compute_reward_random_vs_oracle():
    rng = np.random.RandomState(3)
    for n, ln in [(1, 36), (127, 9), (128, 16), (1000, 25), (5000, 36), (300, 7)]:
        ag = rng.randint(-1, 2, size=(n, ln)).astype(np.float32)
        g = rng.randint(-1, 2, size=(n, ln)).astype(np.float32) * (rng.rand(n, ln) < 0.2)
        ag[::2] = np.where(g[::2] != 0, g[::2], ag[::2])
        got = bpg.compute_reward(torch.as_tensor(ag, device=DEV), torch.as_tensor(g.astype(np.float32), device=DEV))
        np.testing.assert_array_equal(bits(got), O.compute_reward(ag, g).view(np.uint32))
    assert bpg.compute_reward(np.zeros((0, 36), np.float32), np.zeros((0, 36), np.float32)).shape == (0,)


def test_discounted_return_random_vs_oracle():
    rng = np.random.RandomState(11)
    for T, B in [(50, 130), (7, 64), (1, 3), (100, 65)]:
        r = np.where(rng.rand(T, B) < 0.7, -1.0, -0.0).astype(np.float32)
        if T == 7:      # arbitrary float rewards: catches fused multiply-adds (rounding differs)
            r = rng.normal(size=(T, B)).astype(np.float32)
        G = bpg.discounted_return(torch.as_tensor(r, device=DEV), 1. - 1. / 50).cpu().numpy()
        for b in range(0, B, 9):
            np.testing.assert_array_equal(G[b].view(np.uint64),
                                          O.discounted_return(r[:, b].astype(np.float64), 1. - 1. / 50).view(np.uint64))


def test_partial_reset_and_curriculum_surface():
    env = bpg.make('BlocksTouch-v0', num_envs=8, device=DEV)
    env.reset()
    m = np.full(8, 1 << bpg._lib.lib().bpg_pair_index(2, 3), np.uint32)
    _, r, done, info = env.step(None, contact_masks=m.view(np.int32))
    assert info['is_success'].all() and (bits(r) == 0x80000000).all()
    mask = np.array([1, 0, 0, 0, 1, 0, 0, 0], np.uint8)
    obs = env.reset(env_mask=mask)
    _, r, done, info = env.step(None, contact_masks=np.zeros(8, np.int32))
    # reset envs lost the latch, the others keep it; the matrix itself persists (Q1)
    np.testing.assert_array_equal(info['is_success'].cpu().numpy(), mask == 0)
    assert (obs['achieved_goal'][:, 2 * 4 + 3] == 1).all()
    # curriculum surface (quirk Q6), golden difficulty traces
    for env_id, v in VARIANTS.items():
        e = bpg.make(env_id, num_envs=2, device=DEV)
        if 'error' in v['difficulty']:
            with pytest.raises({'NotImplementedError': NotImplementedError,
                                'AttributeError': AttributeError}[v['difficulty']['error']]):
                e.unwrapped.increase_difficulty()
        else:
            for row in v['difficulty']['rows']:
                assert e.unwrapped.increase_difficulty() is row['ret']
                assert e.unwrapped.get_difficulty() == row['difficulty']
                assert e.obj_range == row['obj_range']
                if row['wrong_obj_range'] is not None:
                    assert e.wrong_obj_range == row['wrong_obj_range']
        if v['set_test'] != 'ok':
            with pytest.raises(NotImplementedError):
                e.unwrapped.set_test()


def test_configure_dims_probe():
    """config.py:159-183 probe on the drop-in single env."""
    for env_id, v in VARIANTS.items():
        env = bpg.make(env_id, device=DEV)
        assert bpg.configure_dims(env) == dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)
        assert env.seed(5) == [5]


# --------------------------------------------------------------------------
# BASELINE.json full size: size-independent properties
# --------------------------------------------------------------------------
def test_full_size_properties():
    """1 048 576 envs: (a) the fused T-step launch equals T single-step launches,
    (b) sharding invariance: two half-size handles with env_id_base offsets equal
    the full batch, (c) success count == popcount of the final flags, (d) rewards
    are only -1.0f / -0.0f and success implies a -0.0 at some step, (e) the
    episode writer's ag rows equal the float view of the packed state."""
    import ctypes as C
    B, T, seed = 1 << 20, 50, 2026
    env_id = 'BlocksTouchVariation-v0'
    L = bpg._lib.lib()
    full = bpg.make(env_id, num_envs=B, device=DEV, seed=seed)
    full.reset()
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flg = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, d_masks=None, d_reward_tb=rew.data_ptr(), d_flags_tb=flg.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(full._h, C.byref(args), None))
    # (a)
    step = bpg.make(env_id, num_envs=B, device=DEV, seed=seed)
    step.reset()
    for t in range(T):
        _, r, done, info = step.step(None)
        assert torch.equal(r.view(torch.int32), rew[t].view(torch.int32))
        f = info['is_success'].to(torch.uint8) | (done.to(torch.uint8) << 1) | ((r == 0).to(torch.uint8) << 2)
        assert torch.equal(f, flg[t])
    assert torch.equal(step.state_now, full.state_now) and torch.equal(step.state_ever, full.state_ever)
    # (b)
    halves = [bpg.make(env_id, num_envs=B // 2, device=DEV, seed=seed, env_id_base=k * (B // 2)) for k in range(2)]
    for k, h in enumerate(halves):
        h.reset()
        rw = torch.empty((T, B // 2), dtype=torch.float32, device=DEV)
        a = bpg._lib.RolloutArgs(T=T, d_masks=None, d_reward_tb=rw.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(h._h, C.byref(a), None))
        sl = slice(k * (B // 2), (k + 1) * (B // 2))
        assert torch.equal(rw.view(torch.int32), rew[:, sl].view(torch.int32))
        assert torch.equal(h.state_ever, full.state_ever[sl])
    # (c)
    assert int(full.success_count().item()) == int((flg[-1] & 1).sum().item())
    assert (flg[-1] & 2).all() and not (flg[-2] & 2).any()          # done exactly at step 50
    # (d)
    rb = rew.view(torch.int32)
    ok = rb == -2147483648
    assert (ok | (rb == -1082130432)).all()
    assert torch.equal(ok.any(dim=0), (flg[-1] & 1).bool())
    # (e) episode writer on a smaller batch
    Bs = 1 << 16
    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=seed), None,
                           dims_of(env_id), None, T, rollout_batch_size=Bs, output='torch')
    ep = w.generate_rollouts_fused(want_reward=True)
    assert torch.equal(ep['reward_tb'].view(torch.int32), rew[:, :Bs].view(torch.int32))
    assert torch.equal(ep['ag'][:, T], w.env._get_obs()['achieved_goal'])
    assert torch.equal(ep['info_is_success'][:, :, 0].T.to(torch.uint8), flg[:, :Bs] & 1)
    assert (ep['ag'][:, 0] == -1).all()
    g = torch.as_tensor(np.asarray(VARIANTS[env_id]['goal'], np.float32), device=DEV)
    assert (ep['g'] == g).all()


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_observation_assembly(env_id):
    """Reference _get_obs observation vector (fixture recorded on seeded random site / robot
    values): float64 bit-exact, float32 == the reference's float32 store; all fixture rows
    as one batch, num_objs set per env through the reset's explicit num_grey."""
    d = np.load(os.path.join(GOLDEN, 'obs_%s.npz' % short(env_id)))
    B = len(d['obs'])
    env = bpg.make(env_id, num_envs=B, device=DEV)
    env.reset(num_grey=d['num_objs'] - 4)
    nb = env.num_objs - 2
    phys = {k: d[k] for k in env.PHYS_FIELDS}
    for k in ('obj_pos', 'obj_rot', 'obj_velp', 'obj_velr'):
        phys[k] = np.ascontiguousarray(d[k][:, :nb])
    o64 = env.assemble_observation(phys, dt=float(d['dt']), dtype=torch.float64)
    np.testing.assert_array_equal(o64.cpu().numpy().view(np.uint64), d['obs'].view(np.uint64))
    o32 = env.assemble_observation(phys, dt=float(d['dt']))
    assert o32 is env.observation
    np.testing.assert_array_equal(o32.cpu().numpy(), d['obs'].astype(np.float32))
    # straight into a batch-major episode slot (row stride = (T+1)*dimo)
    ep_o = torch.zeros(B, 3, env.dimo, dtype=torch.float32, device=DEV)
    env.assemble_observation(phys, dt=float(d['dt']), out=ep_o[:, 1])
    np.testing.assert_array_equal(ep_o[:, 1].cpu().numpy(), d['obs'].astype(np.float32))
    assert (ep_o[:, 0] == 0).all() and (ep_o[:, 2] == 0).all()


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_reset_placement(env_id):
    """Reference _randomize_objects replayed on the GPU with the recorded standard variates:
    bit-identical float64 positions (fixture rows grouped by their launch constants)."""
    d = np.load(os.path.join(GOLDEN, 'placement_%s.npz' % short(env_id)))
    keys = np.stack([d['grip_xy'][:, 0], d['grip_xy'][:, 1], d['obj_range'], np.nan_to_num(d['max_obj_range']),
                     np.nan_to_num(d['wrong_obj_range']), d['test'].astype(float), d['challenge'].astype(float)], 1)
    groups = {}
    for k, row in enumerate(map(tuple, keys)):
        groups.setdefault(row, []).append(k)
    assert len(groups) >= 3
    for row, idx in groups.items():
        idx = np.asarray(idx)
        env = bpg.make(env_id, num_envs=len(idx), device=DEV)
        env.reset(num_grey=d['num_objs'][idx] - 4)
        env.obj_range, env.max_obj_range, env.wrong_obj_range = row[2], row[3], row[4]
        env.challenge = bool(row[6])
        var = {k: np.nan_to_num(d[k][idx], nan=0.5) for k in ('eu', 'gz', 'gu')}
        for k in var:       # kernels need at least the minimum count of columns
            if var[k].shape[1] < 2:
                var[k] = np.concatenate([var[k], np.full((len(idx), 2), 0.5)], 1)
        xy, status = env.randomize_objects(test=bool(row[5]), grip_xy=row[:2], variates=var)
        assert (status == 0).all()
        got, exp = xy.cpu().numpy(), d['xy'][idx]
        for j, k in enumerate(idx):
            nb = int(d['num_objs'][k]) - 2
            np.testing.assert_array_equal(got[j, :nb].view(np.uint64), exp[j, :nb].view(np.uint64))
            assert np.isnan(got[j, nb:]).all()          # absent Variation blocks are left untouched


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_reset_placement_generator_properties(env_id):
    """Counter-based draw mode (build-defined): every constraint the reference's rejection
    loops enforce holds, results are deterministic in (seed, global env index)."""
    B = 20000
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=11)
    env.reset()
    xy, status = env.randomize_objects()
    assert (status == 0).all()
    xy = xy.cpu().numpy()
    nobj = env.num_objs_per_env().cpu().numpy()
    TW, TH, TX, TY, MIN = 0.225, 0.32499999999999996, 1.3, 0.75, 0.07500000000000001
    grip = np.array([1.3419, 0.7491])
    present = np.arange(xy.shape[1])[None, :] < (nobj - 2)[:, None]
    assert not np.isnan(xy[present]).any()
    if env_id in ('GripperTouch-v0', 'ToppleTower-v0'):
        assert (np.linalg.norm(xy[:, 0] - grip, axis=1) >= 0.1).all()
        assert (np.abs(xy[:, 0] - grip) <= env.obj_range).all()
    else:
        moved = xy[:, 1:] if env_id.startswith('BlocksTouch-') or 'Curriculum-v0' == env_id[11:] else xy
        on = (np.abs(moved[..., 0] - TX) <= TW) & (np.abs(moved[..., 1] - TY) <= TH)
        assert on[present[:, -moved.shape[1]:]].all()
        dist01 = np.linalg.norm(xy[:, 0] - xy[:, 1], axis=1)
        assert (dist01 >= MIN - 1e-12).all() and (dist01 <= env.obj_range + 1e-12).all()
    if 'Variation' in env_id or 'Choose' in env_id:
        for a in range(xy.shape[1]):
            for b in range(a + 1, xy.shape[1]):
                both = present[:, a] & present[:, b]
                assert (np.linalg.norm(xy[both, a] - xy[both, b], axis=1) >= MIN - 1e-12).all()
    # shard invariance + determinism
    half = bpg.make(env_id, num_envs=B // 2, device=DEV, seed=11, env_id_base=B // 2)
    half.reset()
    xy2, _ = half.randomize_objects()
    np.testing.assert_array_equal(np.nan_to_num(xy2.cpu().numpy()), np.nan_to_num(xy[B // 2:]))


@pytest.mark.parametrize('B', [1, 2, 3, 5, 63, 64, 65, 129, 255, 257, 4100])
def test_edge_batch_sizes(B):
    """Partial waves / vector tails of every kernel family against the oracle (Variation,
    T=7): per-step masks, fused packed rollout, episode writer, float views, success count."""
    import ctypes as C
    env_id, T = 'BlocksTouchVariation-v0', 7
    rng = np.random.RandomState(B)
    num_grey = rng.randint(3, size=B).astype(np.int32)
    masks = legal_masks(rng, T, B, 4 + num_grey, density=2)
    ref, _ = O.generate_rollouts(env_id, B, T, masks, num_grey=num_grey)
    dm = i32(masks)
    L = bpg._lib.lib()
    # per-step
    env = bpg.make(env_id, num_envs=B, device=DEV, max_episode_steps=T)
    obs = env.reset(num_grey=num_grey)
    np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), ref['ag'][:, 0])
    for t in range(T):
        obs, r, done, info = env.step(None, contact_masks=dm[t])
        np.testing.assert_array_equal(bits(r), ref['reward'][t].view(np.uint32))
        np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), ref['ag'][:, t + 1])
        np.testing.assert_array_equal(info['is_success'].cpu().numpy(), ref['info_is_success'][:, t, 0] != 0)
        assert bool(done.all()) == (t == T - 1) and bool(done.any()) == (t == T - 1)
    assert int(env.success_count().item()) == int(ref['info_is_success'][:, -1, 0].sum())
    # fused packed + episode writer (reset inside the launch)
    for episode in (False, True):
        e2 = bpg.make(env_id, num_envs=B, device=DEV, max_episode_steps=T)
        ng = torch.as_tensor(num_grey, device=DEV)
        rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
        cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
        kw = {}
        if episode:
            ag = torch.empty((B, T + 1, 36), dtype=torch.float32, device=DEV)
            g = torch.empty((B, T, 36), dtype=torch.float32, device=DEV)
            sc = torch.empty((B, T, 1), dtype=torch.float32, device=DEV)
            kw = dict(d_ep_ag=ag.data_ptr(), d_ep_g=g.data_ptr(), d_ep_success=sc.data_ptr())
        args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=dm.data_ptr(),
                                    d_reward_tb=rew.data_ptr(), d_success_count=cnt.data_ptr(), **kw)
        bpg._lib.check(L.bpg_rollout_v2(e2._h, C.byref(args), None))
        np.testing.assert_array_equal(bits(rew), ref['reward'].view(np.uint32))
        assert int(cnt.item()) == int(ref['info_is_success'][:, -1, 0].sum())
        assert torch.equal(e2.state_now, env.state_now) and torch.equal(e2.state_ever, env.state_ever)
        if episode:
            np.testing.assert_array_equal(ag.cpu().numpy(), ref['ag'])
            np.testing.assert_array_equal(g.cpu().numpy(), ref['g'])
            np.testing.assert_array_equal(sc.cpu().numpy(), ref['info_is_success'])


@pytest.mark.parametrize('T', [1, 2, 17, 127, 200, 1023])
def test_rollout_lengths(T):
    """T=1, and T large enough that the episode writer's LDS history tile exceeds the 64 KiB
    default (opt-in dynamic LDS path); max_episode_steps = 100 so `done` flips inside."""
    env_id, B = 'BlocksTouchChoose-v0', 130
    rng = np.random.RandomState(T)
    masks = legal_masks(rng, T, B, 5, density=3)
    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, max_episode_steps=100),
                           None, dims_of(env_id), None, T, rollout_batch_size=B)
    ep = w.generate_rollouts_fused(masks=i32(masks), want_reward=True)
    L = O.lib()
    import ctypes as C
    envs = (O.Env * B)()
    for b in range(B):
        L.orc_env_init(C.byref(envs[b]), O.VARIANTS[env_id], 100)
    ref, _ = O.generate_rollouts(env_id, B, T, masks, envs=envs)
    np.testing.assert_array_equal(ep['ag'], ref['ag'])
    np.testing.assert_array_equal(ep['info_is_success'], ref['info_is_success'])
    np.testing.assert_array_equal(ep['reward_tb'].view(np.uint32), ref['reward'].view(np.uint32))


def test_argument_errors_are_reported():
    import ctypes as C
    L = bpg._lib.lib()
    env = bpg.make('BlocksTouch-v0', num_envs=16, device=DEV)
    with pytest.raises(bpg._lib.BpgError):       # contact lists before an id2obj table -> fine here, table is preset
        bpg._lib.check(L.bpg_step_contacts(env._h, None, None, 4, None, None, None))
    args = bpg._lib.RolloutArgs(T=0)
    assert L.bpg_rollout_v2(env._h, C.byref(args), None) == bpg._lib.ERR_ARG
    args = bpg._lib.RolloutArgs(T=5000)
    assert L.bpg_rollout_v2(env._h, C.byref(args), None) == bpg._lib.ERR_ARG
    buf = torch.empty(16 * 16 + 1, dtype=torch.float32, device=DEV)
    assert L.bpg_get_achieved_goal_f32(env._h, C.c_void_p(buf.data_ptr()), 8, None) == bpg._lib.ERR_ARG   # ld < dimg
    # misaligned (4-byte offset) output still works through the scalar path
    out = buf[1:].view(16, 16)
    bpg._lib.check(L.bpg_get_achieved_goal_f32(env._h, C.c_void_p(out.data_ptr()), 16, None))
    assert (out == -1).all()
    m = torch.zeros(17, dtype=torch.int32, device=DEV)[1:]
    r = torch.empty(17, dtype=torch.float32, device=DEV)[1:]
    bpg._lib.check(L.bpg_step_masks(env._h, C.c_void_p(m.data_ptr()), C.c_void_p(r.data_ptr()), None, None))
    assert (bits(r) == 0xBF800000).all()
    with pytest.raises(KeyError):
        bpg.make('NoSuchEnv-v0')


def test_mat2euler_unpinned_formula():
    """mat2euler is third-party (gym.envs.robotics.rotations) and absent from the reference
    tree: PARITY UNPINNED.  This only checks the kernel against a NumPy statement of the
    recalled gym-0.10 formula, tolerance 1e-13 (device atan2 vs libm), incl. the gimbal branch."""
    rng = np.random.RandomState(0)
    q, _ = np.linalg.qr(rng.normal(size=(500, 3, 3)))
    q[:5, 2, 2] = 0.0
    q[:5, 1, 2] = 0.0                    # cy == 0 -> second branch
    cy = np.sqrt(q[:, 2, 2] ** 2 + q[:, 1, 2] ** 2)
    cond = cy > np.finfo(np.float64).eps * 4.0
    exp = np.empty((500, 3))
    exp[:, 2] = np.where(cond, -np.arctan2(q[:, 0, 1], q[:, 0, 0]), -np.arctan2(-q[:, 1, 0], q[:, 1, 1]))
    exp[:, 1] = -np.arctan2(-q[:, 0, 2], cy)
    exp[:, 0] = np.where(cond, -np.arctan2(q[:, 1, 2], q[:, 2, 2]), 0.0)
    got = bpg.mat2euler(q, device=DEV).cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-13)


def test_rollout_student_torch_policy_matches_numpy_policy():
    """A device-resident policy (`device_tensors = True`: torch in / torch out, nothing crosses
    PCIe) produces the same episode as the NumPy-policy path; pg worker with an expert."""
    env_id, B, T = 'BlocksTouchChooseCurriculum-v0', 300, 50
    rng = np.random.RandomState(1)
    acts = rng.uniform(-1, 1, size=(T, B, 4)).astype(np.float32)
    masks = legal_masks(rng, T, B, 5, density=4)

    class NpPolicy:
        def __init__(self):
            self.t = 0

        def get_actions(self, o, ag, g, **kw):
            assert isinstance(o, np.ndarray)
            self.t += 1
            return acts[self.t - 1]

    class TorchPolicy(NpPolicy):
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            assert torch.is_tensor(o) and o.is_cuda and ag.shape == (B, 25)
            self.t += 1
            return torch.as_tensor(acts[self.t - 1], device=DEV), torch.zeros(B, device=DEV)

    eps = []
    for pol, compute_Q in ((NpPolicy(), False), (TorchPolicy(), True)):
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV,
                                                         contact_source=bpg.ScriptedContacts(masks=i32(masks))),
                               pol, dims_of(env_id), None, T, rollout_batch_size=B, compute_Q=compute_Q)
        ep = w.generate_rollouts()
        eps.append({k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in ep.items()})
        assert (torch.is_tensor(ep['ag'])) == isinstance(pol, TorchPolicy)
        assert w.logs('test')[0][0] == 'test/success_rate' and w.current_success_rate() == w.success_history[-1]
    for k in eps[0]:
        np.testing.assert_array_equal(eps[0][k], eps[1][k], err_msg=k)
    ref, _ = O.generate_rollouts(env_id, B, T, masks)
    np.testing.assert_array_equal(eps[0]['ag'], ref['ag'])
    np.testing.assert_array_equal(eps[0]['u'], acts.transpose(1, 0, 2))
    # increase_difficulty through the worker (rollout.py:66-73)
    assert w.increase_difficulty() == 1


@pytest.mark.parametrize('env_id', ['BlocksTouchVariation-v0', 'ToppleTower-v0', 'GripperTouch-v0'])
def test_contacts_to_masks_matches_oracle(env_id):
    """bpg_contacts_to_masks (the id2obj half of _step_callback) on the golden rollouts'
    geom-id contact lists == the oracle's mask of the same list; feeding those masks to the
    fused kernel reproduces the reference episode."""
    import ctypes as C
    d = np.load(os.path.join(GOLDEN, 'rollout_%s.npz' % short(env_id)))
    B, T = int(d['B']), int(d['T'])
    L = bpg._lib.lib()
    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None,
                           dims_of(env_id), None, T, rollout_batch_size=B)
    for r in range(2):
        w.env.reset(num_grey=d['r%d_num_objs' % r] - 4)          # selects each env's id2obj model
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        dm = torch.empty((T, B), dtype=torch.int32, device=DEV)
        for t in range(T):
            pairs = torch.as_tensor(np.ascontiguousarray(con[:, t]), device=DEV)
            ncon = torch.as_tensor(np.ascontiguousarray(d['r%d_ncon' % r][:, t]), device=DEV)
            bpg._lib.check(L.bpg_contacts_to_masks(w.env._h, C.c_void_p(pairs.data_ptr()), C.c_void_p(ncon.data_ptr()),
                                                   int(pairs.shape[1]), C.c_void_p(dm[t].data_ptr()), None))
        exp = fixture_masks(env_id, d['r%d_contacts' % r], d['r%d_ncon' % r], d['r%d_model_idx' % r])
        np.testing.assert_array_equal(dm.cpu().numpy().view(np.uint32), exp)
        ep = w.generate_rollouts_fused(masks=dm, num_grey=d['r%d_num_objs' % r] - 4)
        np.testing.assert_array_equal(ep['ag'], d['r%d_ag' % r])
        np.testing.assert_array_equal(ep['info_is_success'], d['r%d_info_is_success' % r])


def test_time_limit_saturation_at_maximum_horizon():
    """max_episode_steps = T = 1023 (the largest the 10-bit elapsed field allows): `done` flips
    exactly at step 1023 and stays set; further steps keep saturating instead of wrapping."""
    import ctypes as C
    env_id, B, T = 'BlocksTouch-v0', 70, 1023
    rng = np.random.RandomState(0)
    masks = legal_masks(rng, T, B, 4, density=5)
    env = bpg.make(env_id, num_envs=B, device=DEV, max_episode_steps=T)
    env.reset()
    flg = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, d_masks=i32(masks).data_ptr(), d_flags_tb=flg.data_ptr(), d_reward_tb=rew.data_ptr())
    bpg._lib.check(bpg._lib.lib().bpg_rollout_v2(env._h, C.byref(args), None))
    f = flg.cpu().numpy()
    assert not (f[:-1] & 2).any() and (f[-1] & 2).all()
    L = O.lib()
    envs = (O.Env * B)()
    for b in range(B):
        L.orc_env_init(C.byref(envs[b]), O.VARIANTS[env_id], T)
    ref, _ = O.generate_rollouts(env_id, B, T, masks, envs=envs)
    np.testing.assert_array_equal(bits(rew), ref['reward'].view(np.uint32))
    for _ in range(3):                                   # past the limit: still done, no wrap-around
        _, _, done, _ = env.step(None, contact_masks=np.zeros(B, np.int32))
        assert done.all()
    assert ((env.state_now >> 21) & 1023 == 1023).all()
    with pytest.raises(bpg._lib.BpgError):
        bpg.make(env_id, num_envs=4, device=DEV, max_episode_steps=1024)


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_golden_set_action_vector(env_id):
    """_set_action's 9-vector (fetch_env.py:170-185 after the robot_env.py:58 clip), bit-exact."""
    d = np.load(os.path.join(GOLDEN, 'obs_%s.npz' % short(env_id)))
    env = bpg.make(env_id, num_envs=len(d['actions']), device=DEV)
    got = env.set_action_vector(d['actions']).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint64), d['set_action'].view(np.uint64))


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_random_contact_lists_vs_oracle(env_id):
    """Ragged geom-id contact lists (0..12 entries drawn from ALL geoms of the env's model:
    fingers, table, cubes and None-mapped robot / floor / mocap geoms, duplicates, self
    pairs) through the contact-list kernel vs the oracle's list form, 2 episodes x 30 steps."""
    v = VARIANTS[env_id]
    B, T, maxcon = 300, 30, 12
    rng = np.random.RandomState(len(env_id))
    env = bpg.make(env_id, num_envs=B, device=DEV)
    oenvs = [O.OracleEnv(env_id) for _ in range(B)]
    for ep in range(2):
        num_grey = rng.randint(3, size=B).astype(np.int32)
        obs = env.reset(num_grey=num_grey)
        model = num_grey if 'Variation' in env_id else np.zeros(B, np.int32)
        ngeom = np.array([len(v['models'][m]['id2obj']) for m in model])
        exp0 = np.stack([e.reset(num_grey=int(g)) for e, g in zip(oenvs, num_grey)])
        np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), exp0.astype(np.float32))
        for t in range(T):
            ncon = rng.randint(0, maxcon + 1, size=B).astype(np.int32)
            pairs = (rng.randint(0, 1 << 30, size=(B, maxcon, 2)) % ngeom[:, None, None]).astype(np.int16)
            obs, r, done, info = env.step(None, contacts=(pairs, ncon))
            exp = [e.step(v['models'][int(m)]['id2obj'], pairs[b, :ncon[b]])
                   for b, (e, m) in enumerate(zip(oenvs, model))]
            np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(),
                                          np.stack([x[0] for x in exp]).astype(np.float32))
            np.testing.assert_array_equal(bits(r), np.array([np.float32(x[1]).view(np.uint32) for x in exp]))
            np.testing.assert_array_equal(done.cpu().numpy(), np.array([x[2] for x in exp]))
            np.testing.assert_array_equal(info['is_success'].cpu().numpy(), np.array([x[3] for x in exp]))


def test_graphed_rollout_equals_eager():
    """The per-step rollout loop captured into a hipGraph and replayed == the eager loop,
    replay after replay (Variation: fresh explicit num_grey per replay)."""
    env_id, B, T = 'BlocksTouchVariation-v0', 2048, 50
    rng = np.random.RandomState(4)
    masks = i32(legal_masks(rng, T, B, 4, density=3))       # pairs among the 4 always-present objects

    class Policy:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return torch.tanh(ag[:, :4] * 0.5 + o[:, :4])     # depends on the evolving state

    def worker():
        src = bpg.ScriptedContacts(masks=masks)
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src),
                               Policy(), dims_of(env_id), None, T, rollout_batch_size=B)
        return w, src
    wg, _ = worker()
    we, src_e = worker()
    for r in range(3):
        ng = rng.randint(3, size=B).astype(np.int32)
        epg = wg.generate_rollouts_graphed(num_grey=ng)
        src_e.t = 0
        epe = we.generate_rollouts(num_grey=ng)
        for k in epe:
            assert torch.equal(epg[k], epe[k]), k
        assert wg.success_history[-1] == we.success_history[-1]
        assert torch.equal(wg.env.state_ever, we.env.state_ever)
    assert (epg['o'][:, 0, 0] == torch.as_tensor(ng + 2, device=DEV).float()).all()   # Variation obs[0] = num_blocks
    # generator-drawn episodes: the stream index is a device word, so replays draw fresh num_grey
    # and track an eager worker that performs the same sequence of resets
    wg2, _ = worker()
    we2, src_e2 = worker()
    seen = []
    for r in range(3):
        epg = wg2.generate_rollouts_graphed()
        src_e2.t = 0
        epe = we2.generate_rollouts()
        seen.append(wg2.env.num_objs_per_env().clone())
        for k in epe:
            assert torch.equal(epg[k], epe[k]), k
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


class _ZeroPolicy:                       # module level so save_policy can pickle it
    def get_actions(self, o, ag, g, **kw):
        return np.zeros((o.shape[0], 4), np.float32)


def test_rollout_worker_small_surface(tmp_path):
    """reset_rollout(i) (rollout.py:47-58), logs() prefixes (:186-204), save_policy (:180-184),
    clear_history, the reference default rollout_batch_size=2 through a single-env factory."""
    import pickle
    P = _ZeroPolicy

    env_id = 'BlocksTouch-v0'
    w = bpg.RolloutStudent(lambda: bpg.make(env_id, device=DEV), P(), dims_of(env_id), None, 50,
                           rollout_batch_size=2, compute_Q=False)
    assert w.env.num_envs == 2 and not w.env.scalar_api
    pair = 1 << bpg._lib.lib().bpg_pair_index(2, 3)
    w.env.step(None, contact_masks=np.array([pair, 0], np.int32))
    assert w.env.state_ever.cpu().numpy()[0] < 0                   # env 0 latched success
    w.reset_rollout(0)
    assert (w.env.state_ever.cpu().numpy() >= 0).all()
    assert w.initial_ag[0, 2 * 4 + 3].item() == 1.0                # matrix persists over the reset (Q1)
    ep = w.generate_rollouts()
    assert ep['o'].shape == (2, 51, 40) and ep['info_is_success'].shape == (2, 50, 1)
    assert w.logs('worker')[0][0] == 'worker/success_rate' and w.logs('')[-1] == ('episode', 2)
    assert w.logs('a/')[0][0] == 'success_rate'      # reference quirk: a trailing '/' drops the prefix (rollout.py:201-204)
    path = tmp_path / 'policy.pkl'
    w.save_policy(str(path))
    assert isinstance(pickle.load(open(path, 'rb')), P)
    w.clear_history()
    assert len(w.success_history) == 0
    # compute_Q: the policy returns (u, Q); mean_Q is logged and averaged over the history (rollout.py:99-107, :176-178)
    class QPolicy:
        def get_actions(self, o, ag, g, compute_Q=False, **kw):
            assert compute_Q
            return np.zeros((o.shape[0], 4), np.float32), np.full((o.shape[0], 1), 0.25, np.float32)
    wq = bpg.RolloutStudent(lambda: bpg.make(env_id, device=DEV), QPolicy(), dims_of(env_id), None, 50,
                            rollout_batch_size=2, compute_Q=True)
    wq.generate_rollouts()
    assert wq.current_mean_Q() == pytest.approx(0.25) and ('worker/mean_Q', pytest.approx(0.25)) in [(k, pytest.approx(v)) for k, v in wq.logs()]
    # the policy-gradient worker hands save_policy_weights to the policy (policy_gradient/rollout.py:313-314)
    saved = []

    class WPolicy(P):
        kwargs = {}

        def save_weights(self, path):
            saved.append(path)
    wp = bpg.PGRolloutStudent(lambda: bpg.make(env_id, device=DEV), WPolicy(), None, dims_of(env_id), None, 50,
                              rollout_batch_size=2, gamma=0.98)
    wp.save_policy_weights(str(tmp_path))
    assert saved == [str(tmp_path)]


def test_callable_contact_source_with_observation():
    """The physics hook of INTEGRATION.md: a CallableContacts backend that returns geom-id
    contact lists (or masks) and fills env.observation through assemble_observation; the
    rollout worker records those observations and the resulting touch matrices."""
    env_id, B, T = 'BlocksTouchChoose-v0', 64, 6
    d = np.load(os.path.join(GOLDEN, 'obs_%s.npz' % short(env_id)))
    phys = {k: np.repeat(d[k][:1], B, 0) for k in bpg.BatchedBlocksEnv.PHYS_FIELDS}
    finger, cube0, cube1 = 17, 23, 24
    calls = []

    def physics(env, actions):
        assert actions.shape == (B, 4) and float(actions.abs().max()) == 3.0     # as the policy gave them
        calls.append(env.set_action_vector(actions))                            # robot_env.py:58's clip happens here
        env.assemble_observation(phys, dt=float(d['dt']))
        if len(calls) % 2:
            pairs = torch.tensor([[cube0, cube1], [finger, cube0]], dtype=torch.int16, device=DEV).expand(B, 2, 2)
            return pairs.contiguous(), torch.full((B,), 2, dtype=torch.int32, device=DEV)
        return torch.zeros(B, dtype=torch.int32, device=DEV)                     # masks form: nothing touches

    class Pol:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return torch.full((B, 4), 3.0, device=DEV)                           # beyond the clip range

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV,
                                                     contact_source=bpg.CallableContacts(physics, writes_observation=True)),
                           Pol(), dims_of(env_id), None, T, rollout_batch_size=B)
    ep = w.generate_rollouts()
    assert len(calls) == T and torch.equal(calls[0][:, :3], torch.full((B, 3), float(np.float32(0.05)), device=DEV).double())
    exp_obs = torch.as_tensor(d['obs'][0].astype(np.float32), device=DEV)
    assert torch.equal(ep['o'][:, 1], exp_obs.expand(B, -1)) and torch.equal(ep['o'][:, T], exp_obs.expand(B, -1))
    ag = ep['ag'].reshape(B, T + 1, 5, 5)
    assert (ag[:, 1, 2, 3] == 1).all() and (ag[:, 1, 0, 2] == 1).all()          # step 1: cubes touch, finger on cube0
    assert (ag[:, 2, 2, 3] == 0).all() and (ag[:, 2, 1, 1] == -1).all()         # step 2: touched before / never
    assert (ep['info_is_success'] == 1).all()                                    # (2,3) is the goal pair


def test_c_abi_from_plain_cpp_host(tmp_path):
    """examples/c_abi_demo.cpp: a C++ host with only the HIP runtime (no Python, no torch)
    drives libbpgym.so through include/bpgym.h -- build it with hipcc and run it."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    libdir = os.path.join(root, 'blockpuzzle-gym_amd', 'lib')
    exe = str(tmp_path / 'c_abi_demo')
    subprocess.check_call([hipcc, '-O2', '-std=c++17', '-I' + os.path.join(root, 'include'), '-o', exe,
                           os.path.join(root, 'examples', 'c_abi_demo.cpp'), '-L' + libdir, '-lbpgym',
                           '-Wl,-rpath,' + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'reward bits 0x80000000' in out.stdout and 'success count 1000 of 1000, mismatches 0' in out.stdout
    assert 'vs contiguous: 0 mismatches' in out.stdout and 'bpg_row_stride(2^20) = 1048832' in out.stdout
    assert 'return from the rollout call: 0 of 1000 rows differ' in out.stdout
    assert '4 rollouts in one launch' in out.stdout and '; 0 differ from 4 separate launches' in out.stdout
    # round 5: a sharded rollout (views), the library's own RCCL all-gather of the byte returns, the curriculum in the handle
    assert 'sharded rollout + library all-gather (world 1): 0 of 4096 gathered returns differ' in out.stdout
    assert 'curriculum: difficulty 4 at the maximum level, obj_range 0.200 of 0.200' in out.stdout


def test_contact_list_input_is_sanitised():
    """Out-of-range input never indexes out of bounds: ncon beyond maxcon / negative is clamped,
    geom ids outside the id2obj table (negative int16, >= 64) are ignored like None-mapped geoms."""
    env = bpg.make('BlocksTouch-v0', num_envs=5, device=DEV)
    env.reset()
    cube0, cube1, finger = 23, 24, 17
    pairs = np.zeros((5, 4, 2), np.int16)
    pairs[0, :2] = [[cube0, cube1], [finger, cube0]]          # normal
    pairs[1, :] = [[cube0, cube1]] * 4                         # ncon claims 100 -> clamped to maxcon 4
    pairs[2, 0] = [-1, cube1]                                  # negative geom id -> ignored
    pairs[3, 0] = [200, cube0]                                 # beyond the table -> ignored
    pairs[4, 0] = [cube0, cube1]                               # ncon negative -> no contacts
    ncon = np.array([2, 100, 1, 1, -7], np.int32)
    obs, r, done, info = env.step(None, contacts=(pairs, ncon))
    ag = obs['achieved_goal'].cpu().numpy().reshape(5, 4, 4)
    assert ag[0, 2, 3] == 1 and ag[0, 0, 2] == 1 and ag[1, 2, 3] == 1
    assert (ag[2] == -1).all() and (ag[3] == -1).all() and (ag[4] == -1).all()
    np.testing.assert_array_equal(info['is_success'].cpu().numpy(), [True, True, False, False, False])


def test_remaining_abi_entry_points():
    """Entry points not exercised elsewhere: packed compute_reward, desired_goal with a row
    stride, synthetic masks from a later step (t0 > 0), float64 achieved_goal with a stride."""
    import ctypes as C
    L = bpg._lib.lib()
    env_id, B = 'ToppleTower-v0', 777
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=5)
    env.reset()
    rng = np.random.RandomState(9)
    for t in range(5):
        env.step(None, contact_masks=legal_masks(rng, 1, B, 6)[0].view(np.int32))
    # packed reward == float reward of the float views == oracle
    out = torch.empty(B, dtype=torch.float32, device=DEV)
    bpg._lib.check(L.bpg_compute_reward_packed(C.c_void_p(env.state_now.data_ptr()), C.c_void_p(env.state_ever.data_ptr()),
                                               env.info.want_touch, env.info.want_never, B,
                                               C.c_void_p(out.data_ptr()), 0, None))
    obs = env._get_obs()
    ref = O.compute_reward(obs['achieved_goal'].cpu().numpy(), obs['desired_goal'].cpu().numpy())
    np.testing.assert_array_equal(bits(out), ref.view(np.uint32))
    np.testing.assert_array_equal(bits(bpg.compute_reward(obs['achieved_goal'], obs['desired_goal'])), ref.view(np.uint32))
    # desired_goal / float64 achieved_goal into strided slots
    slot = torch.full((B, 3, 36), 7.0, dtype=torch.float32, device=DEV)
    bpg._lib.check(L.bpg_get_desired_goal_f32(env._h, C.c_void_p(slot[:, 1].data_ptr()), 3 * 36, None))
    assert torch.equal(slot[:, 1], obs['desired_goal']) and (slot[:, 0] == 7).all() and (slot[:, 2] == 7).all()
    slot64 = torch.full((B, 2, 36), 7.0, dtype=torch.float64, device=DEV)
    bpg._lib.check(L.bpg_get_achieved_goal_f64(env._h, C.c_void_p(slot64[:, 1].data_ptr()), 2 * 36, None))
    assert torch.equal(slot64[:, 1], obs['achieved_goal'].double()) and (slot64[:, 0] == 7).all()
    # synthetic masks for steps 5..12 == what stepping with the in-kernel source consumes next
    m = torch.empty((8, B), dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks(env._h, 5, 8, C.c_void_p(m.data_ptr()), None))
    np.testing.assert_array_equal(m.cpu().numpy().view(np.uint32), O.synth_masks(5, np.arange(B), 13, 6)[5:])
    twin = bpg.make(env_id, num_envs=B, device=DEV, seed=5)
    twin.load_state_dict(env.state_dict())
    for t in range(8):
        _, r1, _, _ = env.step(None)                               # in-kernel synthetic source at elapsed = 5 + t
        _, r2, _, _ = twin.step(None, contact_masks=m[t])
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32))
    assert torch.equal(env.state_ever, twin.state_ever)
