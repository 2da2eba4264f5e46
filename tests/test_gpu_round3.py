"""Round-3 GPU tests (through the C-ABI, bit-exact unless a tolerance is written in the test):
the policy-gradient return produced by the rollout call itself (step-ok words, no reward stream),
the lane-per-env return kernels (plain -1.0 / -0.0 rewards and general rewards in one launch), the
O(T) scan form, raw masks that carry pairs of absent objects (kept exactly as the reference keeps
them), the sharded success counter, and the reference-owned seed arithmetic."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from oracle import bpg_oracle as O
from util import GOLDEN, VARIANTS, ENV_IDS, short, legal_masks, fixture_masks, valid_by_n

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = bpg._lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731


def i32(a):
    return torch.as_tensor(np.ascontiguousarray(a).view(np.int32), device=DEV)


def dims_of(env_id):
    v = VARIANTS[env_id]
    return dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)


def oracle_G(reward_tb, gamma):
    """(T, B) float32 rewards -> (B, T) float64 returns, the reference's loop (oracle/bpg_oracle.c)."""
    r = np.asarray(reward_tb)
    return np.stack([O.discounted_return(r[:, b].astype(np.float64), gamma) for b in range(r.shape[1])])


def same_bits(a, b):
    np.testing.assert_array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


# ------------------------------------------------------------------ return kernels
@pytest.mark.parametrize('T', [50, 25, 10])
def test_lane_return_kernel_plain_and_general_rewards_in_one_launch(T):
    """T = 50 / 25 / 10 run one env per lane.  Waves whose rewards are all -1.0f / -0.0f take the conditional-
    subtraction path, any other wave the general product + sum: both must give the reference's bits, in the
    same launch, including +0.0f (NOT in the plain set), a partial last wave and padded rows."""
    rng = np.random.RandomState(T)
    B = 64 * 5 + 37
    r = np.where(rng.rand(T, B) < 0.6, -1.0, -0.0).astype(np.float32)
    r[:, 64:128] = rng.normal(size=(T, 64)).astype(np.float32)            # a wave of arbitrary floats
    r[3, 130] = 0.0                                                        # +0.0f: one odd value in a plain wave
    r[:, 200] = np.float32(-0.5)                                           # ... and one odd env in another
    r[:, 320:] = rng.choice(np.array([-1.0, -0.0, 2.5], np.float32), size=(T, B - 320))   # the partial wave
    gamma = 1.0 - 1.0 / 50
    want = oracle_G(r, gamma)
    same_bits(bpg.discounted_return(torch.as_tensor(r, device=DEV), gamma).cpu().numpy(), want)
    padded = bpg.time_major_empty(T, B, torch.float32, DEV)
    big = torch.empty((T, B + 24), dtype=torch.float32, device=DEV)[:, :B]
    for dst in (padded, big):
        dst.copy_(torch.as_tensor(r))
        same_bits(bpg.discounted_return(dst, gamma).cpu().numpy(), want)
    # all-plain input: every wave on the fast path
    r2 = np.where(rng.rand(T, B) < 0.9, -1.0, -0.0).astype(np.float32)
    same_bits(bpg.discounted_return(torch.as_tensor(r2, device=DEV), gamma).cpu().numpy(), oracle_G(r2, gamma))


def test_scan_return_is_within_its_stated_bound():
    """bpg_discounted_return_scan: G[t] = fma(gamma, G[t+1], r[t]); include/bpgym.h promises
    |G_scan - G_ref| <= 4 T 2^-53 sum_k gamma^k |r[t+k]| -- checked against the oracle (the reference's
    order) for env rewards and for arbitrary floats, register (T = 50) and LDS (other T) forms."""
    rng = np.random.RandomState(5)
    for T, B in [(50, 300), (50, 64), (7, 70), (100, 65), (319, 3)]:
        gamma = 1.0 - 1.0 / T
        for kind in ('env', 'float'):
            r = (np.where(rng.rand(T, B) < 0.7, -1.0, -0.0) if kind == 'env' else rng.normal(size=(T, B))).astype(np.float32)
            got = bpg.discounted_return_scan(torch.as_tensor(r, device=DEV), gamma).cpu().numpy()
            want = oracle_G(r, gamma)
            pw = gamma ** np.arange(T)
            bound = np.stack([[4 * T * 2.0 ** -53 * float(np.sum(pw[:T - t] * np.abs(r[t:, b].astype(np.float64))))
                               for t in range(T)] for b in range(B)])
            assert np.all(np.abs(got - want) <= bound), (T, kind, float(np.max(np.abs(got - want) - bound)))
    assert L.bpg_discounted_return_scan(None, 4, 4, 5, 0.9, None, 0, None) != 0
    x = torch.zeros((320, 4), device=DEV)
    with pytest.raises(bpg._lib.BpgError):
        bpg.discounted_return_scan(x, 0.9)


# ------------------------------------------------------------------ G from the rollout call
def _rollout_with_G(env, T, gamma, masks=None, num_grey=None, reset_first=1, want=('reward', 'flags'), episode=False):
    B = env.num_envs
    out = {}
    G = torch.full((B, T), float('nan'), dtype=torch.float64, device=DEV)
    args = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_G=G.data_ptr(), gamma=gamma)
    keep = []
    if masks is not None:
        args.d_masks, args.ld_masks = masks.data_ptr(), int(masks.stride(0)) if T > 1 else B
    if num_grey is not None:
        ng = torch.as_tensor(np.asarray(num_grey, np.int32), device=DEV)
        keep.append(ng)
        args.d_num_grey = ng.data_ptr()
    if 'reward' in want:
        out['reward'] = bpg.time_major_empty(T, B, torch.float32, DEV)
        args.d_reward_tb, args.ld_out = out['reward'].data_ptr(), int(out['reward'].stride(0))
    if 'flags' in want:
        out['flags'] = torch.empty((T, int(L.bpg_row_stride(B))), dtype=torch.uint8, device=DEV)[:, :B]
        args.d_flags_tb, args.ld_out = out['flags'].data_ptr(), int(out['flags'].stride(0))
    if episode:
        out['ag'] = torch.empty((B, T + 1, env.dimg), dtype=torch.float32, device=DEV)
        args.d_ep_ag = out['ag'].data_ptr()
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(args), None))
    torch.cuda.synchronize()
    out['G'] = G
    return out


@pytest.mark.parametrize('env_id', ['BlocksTouchVariation-v0', 'ToppleTower-v0', 'BlocksTouch-v0'])
@pytest.mark.parametrize('T', [50, 25, 10, 7, 33, 64, 1])
def test_rollout_call_produces_the_pg_return(env_id, T):
    """bpg_rollout(d_G): G from the step-ok words of the launch == the reference's loop over the step rewards
    the same call wrote (those are pinned to the oracle elsewhere), for every horizon class (lane kernels 50 /
    25 / 10, the LDS form for the others), odd batches, with and without the in-launch reset, masks and the
    synthetic source, stream path and episode path."""
    rng = np.random.RandomState(T)
    gamma = 1.0 - 1.0 / max(T, 2)
    for B in (193, 2048, 1):
        env = bpg.make(env_id, num_envs=B, device=DEV, seed=3)
        ng = rng.randint(3, size=B)
        nobj = (4 + ng) if env.info.is_variation else env.num_objs
        m = i32(legal_masks(rng, T, B, nobj, density=2))
        for kw in (dict(masks=m, num_grey=ng), dict(masks=m, num_grey=ng, reset_first=0), dict(masks=None),
                   dict(masks=m, num_grey=ng, episode=True), dict(masks=m, num_grey=ng, want=('reward',))):
            o = _rollout_with_G(env, T, gamma, **kw)
            r = o['reward'].cpu().numpy()
            assert set(np.unique(r.view(np.uint32))) <= {0xBF800000, 0x80000000}
            same_bits(o['G'].cpu().numpy(), oracle_G(r, gamma))
            if 'flags' in o:      # bit 2 of the flags byte is the step reward
                np.testing.assert_array_equal((o['flags'].cpu().numpy() >> 2) & 1, (r.view(np.uint32) == 0x80000000))
    # no stream at all: G alone, from the same state
    env = bpg.make(env_id, num_envs=130, device=DEV, seed=3)
    env.reset(num_grey=np.arange(130) % 3)
    sd = env.state_dict()
    o = _rollout_with_G(env, T, gamma, masks=None, reset_first=0, want=('reward',))
    env.load_state_dict(sd)
    o2 = _rollout_with_G(env, T, gamma, masks=None, reset_first=0, want=())
    same_bits(o2['G'].cpu().numpy(), o['G'].cpu().numpy())


def test_rollout_return_at_every_launch_shape():
    """The step-ok words are written by every launch shape of the fused rollout (1 / 2 envs per lane, pipelined /
    plain loop, 256- / 1024-lane blocks) and by the contacts-in-kernel forms."""
    T, gamma = 50, 0.98
    for B in (65536, 131074, 262144, 524288 + 2, 131073):
        env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=11)
        ng = (np.arange(B) * 5) % 3
        m = bpg.time_major_empty(T, B, torch.int32, DEV)
        bpg._lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), int(m.stride(0)), None))
        env.reset(num_grey=ng)
        for kw in (dict(masks=m, num_grey=ng), dict(masks=None)):
            o = _rollout_with_G(env, T, gamma, want=('reward', 'flags'), **kw)
            r = o['reward'].cpu().numpy()
            G = o['G'].cpu().numpy()
            idx = np.r_[0:70, B // 2 - 40:B // 2 + 40, B - 70:B]
            same_bits(G[idx], oracle_G(r[:, idx], gamma))
            # every row: G[:, T-1] is the last reward itself, and G[:, 0] <= ... plain sanity over the whole batch
            np.testing.assert_array_equal(G[:, T - 1].view(np.uint64), r[T - 1].astype(np.float64).view(np.uint64))
            assert np.isfinite(G).all()


def test_rollout_return_for_long_horizons_goes_through_the_reward_stream():
    T, B, gamma = 100, 70, 0.99
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=1, max_episode_steps=200)
    o = _rollout_with_G(env, T, gamma, masks=None, want=('reward',))
    same_bits(o['G'].cpu().numpy(), oracle_G(o['reward'].cpu().numpy(), gamma))
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_G=G.data_ptr(), gamma=gamma)
    assert L.bpg_rollout_v2(env._h, C.byref(a), None) == bpg._lib.ERR_ARG and b'd_reward_tb' in L.bpg_last_error()
    a = bpg._lib.RolloutArgs(T=50, reset_first=1, d_G=G.data_ptr() + 4, gamma=gamma)
    assert L.bpg_rollout_v2(env._h, C.byref(a), None) == bpg._lib.ERR_ARG


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation', 'ToppleTower', 'GripperTouch', 'BlocksTouchChoose'])
def test_golden_pg_return_from_the_fused_call(name):
    """The reference's float64 G (policy_gradient/rollout.py, golden fixtures) from ONE fused call: contacts ->
    masks, episode arrays and G out, no reward stream anywhere."""
    env_id = name + '-v0'
    d = np.load(os.path.join(GOLDEN, 'pg_rollout_%s.npz' % name))
    B, T = int(d['B']), int(d['T'])
    worker = bpg.PGRolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None, None,
                                  dims_of(env_id), None, T, rollout_batch_size=B, gamma=1. - 1. / T)
    for r in range(len(d['modes'])):
        # the recorded contact lists -> masks on the device (env.contacts_to_masks); the oracle's conversion cross-checks it
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        dmasks = worker.env.contacts_to_masks(np.ascontiguousarray(con.transpose(1, 0, 2, 3)),
                                              np.ascontiguousarray(d['r%d_ncon' % r].T), num_grey=d['r%d_num_objs' % r] - 4)
        masks = fixture_masks(env_id, d['r%d_contacts' % r], d['r%d_ncon' % r], d['r%d_model_idx' % r])
        np.testing.assert_array_equal(dmasks.cpu().numpy().view(np.uint32), masks)
        ep = worker.generate_rollouts_fused(masks=dmasks, num_grey=d['r%d_num_objs' % r] - 4)
        assert ep['G'].dtype == np.float64
        same_bits(ep['G'], d['r%d_G' % r])
        np.testing.assert_array_equal(ep['ag'], d['r%d_ag' % r])
        np.testing.assert_array_equal(ep['info_is_success'], d['r%d_info_is_success' % r])
    # stream path on a fresh env, first golden rollout
    env = bpg.make(env_id, num_envs=B, device=DEV)
    masks = fixture_masks(env_id, d['r0_contacts'], d['r0_ncon'], d['r0_model_idx'])
    o = _rollout_with_G(env, T, 1. - 1. / T, masks=i32(masks), num_grey=d['r0_num_objs'] - 4, want=('flags',))
    same_bits(o['G'].cpu().numpy(), d['r0_G'])


# ------------------------------------------------------------------ raw masks with pairs of absent objects
def test_masks_with_pairs_of_absent_objects_behave_as_in_the_reference():
    """fetch_env.py:154-157 demotes 1 -> 0 only for i, j < num_objs, while Variation's matrix is always 6 x 6:
    a raw mask bit for a pair with an object >= num_objs (no id2obj table produces one) is set and then stays 1.
    Per-step kernels, every fused launch shape and the episode writer must keep it exactly as the oracle's
    matrix form does."""
    rng = np.random.RandomState(77)
    env_id, T = 'BlocksTouchVariation-v0', 12
    for B in (193, 131074):
        ng = rng.randint(3, size=B)
        m = rng.randint(0, 1 << 21, size=(T, B)).astype(np.uint32) & rng.randint(0, 1 << 21, size=(T, B)).astype(np.uint32)
        illegal = ~valid_by_n()[4 + ng] & 0x1FFFFF
        assert (m & illegal[None, :]).any()
        nb = min(B, 193)
        ref, _ = O.generate_rollouts(env_id, nb, T, m[:, :nb], num_grey=ng[:nb])
        # fused, stream + episode path
        worker = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None,
                                    dims_of(env_id), None, T, rollout_batch_size=B)
        ep = worker.generate_rollouts_fused(masks=i32(m), num_grey=ng, want_reward=True)
        np.testing.assert_array_equal(ep['ag'][:nb], ref['ag'])
        np.testing.assert_array_equal(ep['info_is_success'][:nb], ref['info_is_success'])
        np.testing.assert_array_equal(ep['reward_tb'][:, :nb].view(np.uint32), ref['reward'].view(np.uint32))
        # stream path (k_rollout shapes) == single steps (k_step_vec4) == the same rewards
        env = bpg.make(env_id, num_envs=B, device=DEV)
        rew = bpg.time_major_empty(T, B, torch.float32, DEV)
        dm = i32(m)
        ngt = torch.as_tensor(ng.astype(np.int32), device=DEV)
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ngt.data_ptr(), d_masks=dm.data_ptr(),
                                 d_reward_tb=rew.data_ptr(), ld_out=int(rew.stride(0)))
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        np.testing.assert_array_equal(rew.cpu().numpy()[:, :nb].view(np.uint32), ref['reward'].view(np.uint32))
        env2 = bpg.make(env_id, num_envs=B, device=DEV)
        env2.reset(num_grey=ng)
        for t in range(T):
            obs, r, done, info = env2.step(np.zeros((B, 4), np.float32), contact_masks=dm[t])
            np.testing.assert_array_equal(r.cpu().numpy().view(np.uint32), rew[t].cpu().numpy().view(np.uint32))
        assert torch.equal(env.state_now, env2.state_now) and torch.equal(env.state_ever, env2.state_ever)
        np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy()[:nb], ref['ag'][:, T])


# ------------------------------------------------------------------ sharded success counter
def test_success_count_is_exact_at_every_grid_size_and_over_repeated_launches():
    """One returning atomic per block on a sharded counter, the last arrival of a shard forwards the sum and
    clears the shard: the count must equal the popcount of the final success flags for every grid size
    (fewer blocks than shards, non-multiples, thousands of blocks) and stay exact launch after launch."""
    T = 5
    for B in (64, 300, 4096, 65536, 70001, 262144, 1 << 20):
        env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=B)
        cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
        fl = bpg.time_major_empty(T, B, torch.uint8, DEV)
        for it in range(3):
            a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_flags_tb=fl.data_ptr(), ld_out=int(fl.stride(0)),
                                     d_success_count=cnt.data_ptr() + 8 * (it % 2), d_zero_count=cnt.data_ptr() + 8 * ((it + 1) % 2))
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
            torch.cuda.synchronize()
            want = int((fl[T - 1] & 1).sum().item())
            assert int(cnt[it % 2].item()) == want and int(cnt[(it + 1) % 2].item()) == 0, (B, it)
            assert int(env.success_count().item()) == want
        # the episode writer counts through the same shards
        w = bpg.RolloutStudent(lambda num_envs: bpg.make('BlocksTouchVariation-v0', num_envs=num_envs, device=DEV, seed=B),
                               None, dims_of('BlocksTouchVariation-v0'), None, T, rollout_batch_size=min(B, 70001), output='torch')
        ep = w.generate_rollouts_fused()
        assert w.success_history[-1] == pytest.approx(float(ep['info_is_success'][:, -1, 0].double().mean().item()), abs=1e-12)


# ------------------------------------------------------------------ reference-owned seed arithmetic
def _draws(env, steps=4):
    env.reset()
    ng = (env.num_objs_per_env() - 4).cpu().numpy()
    m = torch.empty((steps, env.num_envs), dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks(env._h, 0, steps, P(m), None))
    xy, _ = env.randomize_objects()
    return ng, m.cpu().numpy().view(np.uint32), xy.nan_to_num().cpu().numpy()


def test_env_i_seeded_s_draws_what_env_0_seeded_s_plus_1000_i_draws():
    """RolloutStudent.seed (rollout.py:206-210): `env.seed(seed + 1000 * idx)`.  The arithmetic is the
    reference's own and is mirrored on the global env index: env i of a batch seeded s draws exactly what env 0
    of a batch seeded s + 1000 i draws (num_grey, synthetic contacts, cube placement), also across shards
    (env_id_base), across the 64-bit wrap, and as the oracle states the key.  What gym does with the number
    (SHA-512 -> MT19937) is third-party: the draws themselves are the build-defined generator's."""
    env_id, B, s = 'BlocksTouchVariation-v0', 300, 123456789
    ng, m, xy = _draws(bpg.make(env_id, num_envs=B, device=DEV, seed=s))
    for i in (0, 1, 7, 64, 299):
        one = _draws(bpg.make(env_id, num_envs=1, device=DEV, seed=s + 1000 * i))
        assert one[0][0] == ng[i]
        np.testing.assert_array_equal(one[1][:, 0], m[:, i])
        np.testing.assert_array_equal(one[2][0].view(np.uint64), xy[i].view(np.uint64))
    # a shard that starts at global index 100 == the same envs of the whole batch
    part = _draws(bpg.make(env_id, num_envs=50, device=DEV, seed=s, env_id_base=100))
    np.testing.assert_array_equal(part[0], ng[100:150])
    np.testing.assert_array_equal(part[1], m[:, 100:150])
    np.testing.assert_array_equal(part[2].view(np.uint64), xy[100:150].view(np.uint64))
    # 64-bit wrap of seed + 1000 * i
    top = 2 ** 64 - 2500
    w = _draws(bpg.make(env_id, num_envs=8, device=DEV, seed=top))
    w5 = _draws(bpg.make(env_id, num_envs=1, device=DEV, seed=(top + 5000) % 2 ** 64))
    assert w5[0][0] == w[0][5] and (w5[1][:, 0] == w[1][:, 5]).all()
    # the oracle's statement of the same key (oracle/bpg_oracle.c: orc_env_key)
    OL = O.lib()
    assert [OL.orc_num_grey_draw(s, i, 0) for i in range(B)] == list(ng)
    np.testing.assert_array_equal(m, O.synth_masks(s, np.arange(B), 4, 4 + ng))
    # reference_launch: rank r seeds its worker with seed + 10^6 r (train.py:132-133) and numbers its envs from 0
    from blockpuzzle_gym_amd import distributed as D
    assert D.rank_seed(7, 3) == 3000007
    r1 = _draws(D.make_sharded(env_id, 40, 1, 2, device=DEV, seed=s, reference_launch=True))
    e = _draws(bpg.make(env_id, num_envs=1, device=DEV, seed=s + 1000000 + 1000 * 13))
    assert e[0][0] == r1[0][13] and (e[1][:, 0] == r1[1][:, 13]).all()


# ------------------------------------------------------------------ compute_reward on rows that are not ternary
def test_compute_reward_follows_numpys_summation_order_in_both_precisions():
    """tests/golden/compute_reward_order.npz: the reference's compute_reward on float32 and float64 rows whose exact
    sum equals count_nonzero(g) but where the ORDER of the float additions decides d == c.  The float32 and float64
    entry points must reproduce NumPy's outcome row by row (a sequential sum gets ~100 of these rows wrong), for the
    row-by-row and the broadcast-goal forms, through the C-ABI and through the Python surface."""
    d = np.load(os.path.join(GOLDEN, 'compute_reward_order.npz'))
    for key in sorted(k[:-3] for k in d.files if k.endswith('_ag')):
        ag, g, want = d[key + '_ag'], d[key + '_g'], d[key + '_rbits']
        got = bpg.compute_reward(ag, g)
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got.view(np.uint32), want, err_msg=key)
        tg = bpg.compute_reward(torch.as_tensor(ag, device=DEV), torch.as_tensor(g, device=DEV))
        np.testing.assert_array_equal(tg.cpu().numpy().view(np.uint32), want, err_msg=key)
        # one goal row broadcast over many ag rows: compare with NumPy live (numpy is on the box; the order is
        # pinned by the fixture above)
        rows = np.repeat(ag[:5], 30, axis=0)
        rows[::3] *= ag.dtype.type(0.5)
        for r in range(3):
            ref = -(np.sum(rows * g[r], axis=-1) != np.count_nonzero(g[r])).astype(np.float32)
            np.testing.assert_array_equal(bpg.compute_reward(rows, g[r]).view(np.uint32), ref.view(np.uint32))
    # mixed precision follows NumPy's promotion: float32 * float64 (or * int64) is a float64 sum
    ag32, g64 = d['float32_36_ag'], d['float32_36_g'].astype(np.float64)
    ref = -(np.sum(ag32 * g64, axis=-1) != np.count_nonzero(g64, axis=-1)).astype(np.float32)
    np.testing.assert_array_equal(bpg.compute_reward(ag32, g64).view(np.uint32), ref.view(np.uint32))
    # the ternary rows of the env: both precisions agree with the oracle
    rng = np.random.RandomState(0)
    ag = rng.randint(-1, 2, size=(1000, 36)).astype(np.float64)
    g = rng.randint(-1, 2, size=(1000, 36)) * (rng.rand(1000, 36) < 0.2)
    ag[::2] = np.where(g[::2] != 0, g[::2], ag[::2])
    want = O.compute_reward(ag, g.astype(np.float64)).view(np.uint32)
    np.testing.assert_array_equal(bpg.compute_reward(ag, g).view(np.uint32), want)                       # float64 * int64
    np.testing.assert_array_equal(bpg.compute_reward(ag.astype(np.float32), g.astype(np.float32)).view(np.uint32), want)
    # argument checks of the float64 entry point
    x = torch.zeros((4, 257), dtype=torch.float64, device=DEV)
    o = torch.zeros(4, dtype=torch.float32, device=DEV)
    assert L.bpg_compute_reward_f64(P(x), P(x), 4, 257, P(o), 0, None) == bpg._lib.ERR_ARG
    assert L.bpg_compute_reward_f64(P(x), P(x), 4, 256, P(o), 0, None) == 0
    assert L.bpg_compute_reward_f64(C.c_void_p(x.data_ptr() + 4), P(x), 4, 16, P(o), 0, None) == bpg._lib.ERR_ARG    # 8-byte alignment


def test_rollout_with_return_is_capturable_in_a_hip_graph():
    """bpg_rollout(d_G) only enqueues launches (rollout + expansion of the step-ok words; no allocation, no host
    sync, the gamma**k table travels in the kernel arguments): a captured graph replays to the same bits."""
    T, B, gamma = 50, 4096, 0.98
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=9)
    ng = torch.as_tensor(((np.arange(B) * 3) % 3).astype(np.int32), device=DEV)
    m = bpg.time_major_empty(T, B, torch.int32, DEV)
    env.reset(num_grey=ng)
    bpg._lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), int(m.stride(0)), None))
    G = torch.zeros((B, T), dtype=torch.float64, device=DEV)
    fl = bpg.time_major_empty(T, B, torch.uint8, DEV)
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), ld_masks=int(m.stride(0)),
                             d_flags_tb=fl.data_ptr(), ld_out=int(fl.stride(0)), d_G=G.data_ptr(), gamma=gamma)
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    torch.cuda.synchronize()
    want = G.clone()
    s = torch.cuda.Stream(DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), C.c_void_p(s.cuda_stream)))
    for _ in range(3):
        G.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(G.view(torch.int64), want.view(torch.int64))
    f = fl.cpu().numpy()
    r = np.where((f >> 2) & 1, np.float32(-0.0), np.float32(-1.0)).astype(np.float32)
    same_bits(want.cpu().numpy()[:200], oracle_G(r[:, :200], gamma))


# ------------------------------------------------------------------ several rollouts in one launch
def _multi_streams(K, T, B, outs):
    ld = int(L.bpg_row_stride(B))
    r = torch.full((K * T, ld), float('nan'), dtype=torch.float32, device=DEV)[:, :B] if 'reward' in outs else None
    f = torch.full((K * T, ld), 0xEE, dtype=torch.uint8, device=DEV)[:, :B] if 'flags' in outs else None
    return r, f, ld


@pytest.mark.parametrize('env_id,explicit', [('BlocksTouchVariation-v0', True), ('BlocksTouchVariation-v0', False),
                                             ('ToppleTower-v0', False)])
@pytest.mark.parametrize('B', [300, 65536, 100003, 262144, 1 << 20])
def test_k_rollouts_in_one_launch_equal_k_launches(env_id, explicit, B):
    """rollouts = K (the evaluator's loop, train.py:63-64: generate_rollouts K times with nothing between the
    calls): one launch must leave exactly what K launches leave -- reward and flag rows of every rollout, the K
    success counts, the final state, and the generator position (the next draw after the K resets) -- at every
    launch shape, with explicit draws and with the generator's, with all / some / none of the output streams."""
    K, T = 3, 50 if B <= 262144 else 20
    gen = torch.Generator(device=DEV).manual_seed(B + K)
    masks = bpg.time_major_empty(K * T, B, torch.int32, DEV)
    src = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=B)
    bpg._lib.check(L.bpg_synth_masks_ld(src._h, 0, K * T, P(masks), int(masks.stride(0)), None))
    noise = torch.randint(0, 1 << 21, (K * T, B), device=DEV, generator=gen, dtype=torch.int32)
    for _ in range(6):                                   # ~1 % of the bits, legal or not: raw contact words
        noise &= torch.randint(0, 1 << 21, (K * T, B), device=DEV, generator=gen, dtype=torch.int32)
    masks ^= noise
    ldm = int(masks.stride(0))
    ng = torch.randint(0, 3, (K, B), device=DEV, generator=gen, dtype=torch.int32) if explicit else None
    for outs in (('reward', 'flags'), ('flags',), ()):
        one = bpg.make(env_id, num_envs=B, device=DEV, seed=11)
        many = bpg.make(env_id, num_envs=B, device=DEV, seed=11)
        r1, f1, ldo = _multi_streams(K, T, B, outs)
        rk, fk, _ = _multi_streams(K, T, B, outs)
        c1 = torch.zeros(K, dtype=torch.int64, device=DEV)
        ck = torch.zeros(K, dtype=torch.int64, device=DEV)
        for k in range(K):
            a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_masks=masks[k * T:].data_ptr(), ld_masks=ldm, ld_out=ldo if outs else 0,
                                     d_num_grey=None if ng is None else ng[k].data_ptr(),
                                     d_reward_tb=None if r1 is None else r1[k * T:].data_ptr(),
                                     d_flags_tb=None if f1 is None else f1[k * T:].data_ptr(),
                                     d_success_count=c1[k:].data_ptr())
            bpg._lib.check(L.bpg_rollout_v2(one._h, C.byref(a), None))
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_masks=masks.data_ptr(), ld_masks=ldm, ld_out=ldo if outs else 0,
                                 d_num_grey=None if ng is None else ng.data_ptr(),
                                 d_reward_tb=None if rk is None else rk.data_ptr(),
                                 d_flags_tb=None if fk is None else fk.data_ptr(), d_success_count=ck.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(many._h, C.byref(a), None))
        torch.cuda.synchronize()
        assert torch.equal(c1, ck), (outs, c1.tolist(), ck.tolist())
        if r1 is not None:
            assert torch.equal(r1.view(torch.int32), rk.view(torch.int32)) and not torch.isnan(rk).any()
        if f1 is not None:
            assert torch.equal(f1, fk)
            assert [int((fk[k * T + T - 1] & 1).sum().item()) for k in range(K)] == ck.tolist()
        s1, sk = one.state_dict(), many.state_dict()
        for key in s1:
            assert torch.equal(s1[key], sk[key]), key
        assert len(set(ck.tolist())) > 1 or B < 1000, ck.tolist()    # the rollouts really differ
        # and the next reset draws the same thing on both
        one.reset(), many.reset()
        assert torch.equal(one.num_objs_per_env(), many.num_objs_per_env())


def test_k_rollouts_from_the_synthetic_source_and_the_argument_checks():
    B, K, T = 70001, 4, 50
    one = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=5)
    many = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=5)
    f1, fk = torch.empty((K * T, B), dtype=torch.uint8, device=DEV), torch.empty((K * T, B), dtype=torch.uint8, device=DEV)
    c1, ck = torch.zeros(K, dtype=torch.int64, device=DEV), torch.zeros(K, dtype=torch.int64, device=DEV)
    for k in range(K):
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_flags_tb=f1[k * T:].data_ptr(), ld_out=B, d_success_count=c1[k:].data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(one._h, C.byref(a), None))
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_flags_tb=fk.data_ptr(), ld_out=B, d_success_count=ck.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(many._h, C.byref(a), None))
    torch.cuda.synchronize()
    assert torch.equal(f1, fk) and torch.equal(c1, ck)
    # (the synthetic source times its goal contacts by the env key alone, so the K flag blocks are equal by
    # construction; the draws show in the final state below)
    for key, v in one.state_dict().items():
        assert torch.equal(v, many.state_dict()[key]), key
    # K > 1 without the reset, with episode arrays, with d_G, or negative: refused before any launch
    before = many.state_dict()
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)
    ag = torch.empty((B, T + 1, many.dimg), dtype=torch.float32, device=DEV)
    for bad in (dict(reset_first=0, rollouts=2), dict(reset_first=1, rollouts=2, d_G=G.data_ptr(), gamma=0.98),
                dict(reset_first=1, rollouts=2, d_ep_ag=ag.data_ptr()), dict(reset_first=1, rollouts=-1),
                dict(reset_first=1, rollouts=5000)):
        with pytest.raises(bpg._lib.BpgError) as e:
            bpg._lib.check(L.bpg_rollout_v2(many._h, C.byref(bpg._lib.RolloutArgs(T=T, **bad)), None))
        assert e.value.code == bpg._lib.ERR_ARG
    torch.cuda.synchronize()
    for key, v in many.state_dict().items():
        assert torch.equal(v, before[key])


@pytest.mark.parametrize('env_id', ['BlocksTouchVariation-v0', 'BlocksTouch-v0'])
def test_evaluator_loop_in_one_launch_keeps_the_success_history_of_k_calls(env_id):
    """train.py:62-64: evaluator.clear_history(); for _ in range(n_test_rollouts): evaluator.generate_rollouts(...)."""
    B, K, T = 4096, 10, 50
    mk = lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=3)   # noqa: E731
    w1 = bpg.RolloutStudent(mk, None, dims_of(env_id), None, T, rollout_batch_size=B, output='torch')
    wk = bpg.RolloutStudent(mk, None, dims_of(env_id), None, T, rollout_batch_size=B, output='torch')
    masks = bpg.time_major_empty(K * T, B, torch.int32, DEV)
    bpg._lib.check(L.bpg_synth_masks_ld(w1.env._h, 0, K * T, P(masks), int(masks.stride(0)), None))
    w1.clear_history(), wk.clear_history()
    for k in range(K):
        w1.generate_rollouts_fused(masks=masks[k * T:(k + 1) * T])
    fl = bpg.time_major_empty(K * T, B, torch.uint8, DEV)
    rates = wk.evaluate_rollouts_fused(masks, n_rollouts=K, flags_tb=fl)
    assert list(w1.success_history) == list(wk.success_history) == rates and len(rates) == K
    assert w1.n_episodes == wk.n_episodes
    assert w1.logs('test') == wk.logs('test')
    assert torch.equal(w1.env.observation, wk.env.observation)          # Variation: block counts / colours of the last draw
    for key, v in w1.env.state_dict().items():
        assert torch.equal(v, wk.env.state_dict()[key]), key
    with pytest.raises(ValueError):
        wk.evaluate_rollouts_fused(masks, n_rollouts=K + 1)


# ------------------------------------------------------------------ expert input trimming (pg worker)
class _NamedPolicy:
    def __init__(self, env_id):
        self.kwargs = {'info': {'env_name': env_id}}           # policy_gradient/rollout.py:53, :117


def _pg_worker(env_id, B=2, output='numpy', expert=None, policy=None, **kw):
    return bpg.PGRolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=1), policy or _NamedPolicy(env_id),
                                expert, dims_of(env_id), None, 50, rollout_batch_size=B, gamma=0.98, output=output, **kw)


TRIM_ENVS = dict(variation='BlocksTouchVariation-v0', choose='BlocksTouchChoose-v0', tower='ToppleTower-v0', same='BlocksTouch-v0')


@pytest.mark.parametrize('name', sorted(TRIM_ENVS))
def test_trim_equals_the_references_trim_on_its_own_observations(name):
    """policy_gradient/rollout.py:105-171: golden inputs (observations of the reference envs) and outputs of the
    reference's trim, values and dtypes; numpy in -> numpy out, device tensors in -> device tensors out."""
    d = np.load(os.path.join(GOLDEN, 'trim.npz'))
    dimo, dimg, num_objs = (int(x) for x in d[name + '_dims'])
    w = _pg_worker(TRIM_ENVS[name])
    o, g, ag = d[name + '_o'], d[name + '_g'], d[name + '_ag']
    for as_torch in (False, True):
        args = [torch.as_tensor(x, device=DEV) for x in (o, g, ag)] if as_torch else [o, g, ag]
        out = w.trim(*args, dimo, dimg, num_objs=num_objs)
        if name == 'same':
            assert all(a is b for a, b in zip(out, args))
            continue
        for got, key in zip(out, ('o_out', 'g_out', 'ag_out')):
            assert torch.is_tensor(got) == as_torch
            got = got.cpu().numpy() if as_torch else got
            want = d['%s_%s' % (name, key)]
            assert got.dtype == want.dtype and got.shape == want.shape
            np.testing.assert_array_equal(got, want, err_msg=key)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 70001])
def test_trim_kernel_equals_the_oracle_on_random_rows(n):
    """Every block size edge, colours in any slots (the two coloured blocks are taken in SLOT order whatever their
    colours), one-hots that are not one-hot (np.argmax: first maximum, NaN wins), other goal sizes."""
    rng = np.random.RandomState(n)
    # Variation: 87 -> 40, 36 -> 16
    o = rng.standard_normal((n, 87)).astype(np.float32)
    o[:, 0] = rng.randint(2, 5, n)
    for i in range(n):
        slots = rng.permutation(4)
        for k, j in enumerate(slots):
            c = np.zeros(4, np.float32)
            if k < 2:
                c[rng.randint(2, 4)] = 1.0                    # GREEN or BLUE, both may be the same colour
            elif rng.rand() < 0.5:
                c[rng.randint(0, 2)] = 1.0                    # GREY / RED; else all zeros (absent block): argmax 0 = GREY
            o[i, 11 + 19 * j + 15:11 + 19 * j + 19] = c
    if n >= 64:                                               # soft "one-hots": ties and NaNs
        o[5, 11 + 15:11 + 19] = [0.5, 0.5, 0.5, 0.5]          # first maximum -> GREY
        o[5, 11 + 19 + 15:11 + 19 + 19] = [0.1, 0.2, 0.7, 0.7]    # -> GREEN
        o[5, 11 + 38 + 15:11 + 38 + 19] = [0.1, 0.2, np.nan, 9.0]  # NaN counts as the maximum -> GREEN
        o[5, 11 + 57 + 15:11 + 57 + 19] = [np.nan, 0.2, 3.0, 9.0]  # -> GREY
    g = rng.randint(-1, 2, (n, 36)).astype(np.float32)
    ag = rng.randint(-1, 2, (n, 36)).astype(np.float32)
    w = _pg_worker('BlocksTouchVariation-v0')
    got = w.trim(o, g, ag, 40, 16)
    want = O.trim(o, g, ag, 40, 16, True)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype
        np.testing.assert_array_equal(a, b)
    # one coloured block too many / too few anywhere in the batch: NumPy's ValueError (:166)
    for row, paint in ((n - 1, [0, 0, 1, 0]), (0, [1, 0, 0, 0])):
        bad = o.copy()
        slot = [j for j in range(4) if (np.argmax(bad[row, 11 + 19 * j + 15:11 + 19 * j + 19]) >= 2) == (paint[0] == 0)]
        slot = [j for j in range(4) if j not in slot][0]
        bad[row, 11 + 19 * slot + 15:11 + 19 * slot + 19] = paint
        with pytest.raises(ValueError):
            O.trim(bad, g, ag, 40, 16, True)
        with pytest.raises(ValueError):
            w.trim(bad, g, ag, 40, 16)
    # prefix trimming with other geometries: 6 x 6 -> 3 x 3, 5 x 5 -> 4 x 4, 5 x 5 -> 2 x 2
    for env_id, dimo_in, side, dimo, num_objs in (('ToppleTower-v0', 70, 6, 25, 3), ('BlocksTouchChoose-v0', 55, 5, 40, 4),
                                                 ('BlocksTouchChoose-v0', 55, 5, 13, 2)):
        o2 = rng.standard_normal((n, dimo_in)).astype(np.float32)
        g2 = rng.standard_normal((n, side * side)).astype(np.float32)
        ag2 = rng.standard_normal((n, side * side)).astype(np.float32)
        got = _pg_worker(env_id).trim(o2, g2, ag2, dimo, num_objs * num_objs, num_objs=num_objs)
        want = O.trim(o2, g2, ag2, dimo, num_objs * num_objs, False, num_objs=num_objs)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype
            np.testing.assert_array_equal(a, b)


def test_trim_refuses_what_the_reference_asserts_on():
    w = _pg_worker('BlocksTouchVariation-v0')
    o, g, ag = np.zeros((4, 87), np.float32), np.zeros((4, 36), np.float32), np.zeros((4, 36), np.float32)
    with pytest.raises(AssertionError):                       # 9 kept goal entries, not 16 (:146)
        w.trim(o, g, ag, 40, 16, num_objs=3)
    with pytest.raises(AssertionError):                       # o_ would be 40 wide, not 41 (:171)
        w.trim(o, g, ag, 41, 16)
    with pytest.raises(AssertionError):                       # the unbatched branch is `assert(False)` (:114-115)
        w.trim(o[0], g[0], ag[0], 40, 16)
    L_ = bpg._lib
    t = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=DEV)   # noqa: E731
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for bad in (dict(dimg_out=9), dict(dimo_out=41), dict(dimo_in=80), dict(flag=None), dict(n=-1)):
        a = dict(n=4, dimo_in=87, dimg_in=36, dimo_out=40, dimg_out=16, flag=flag)
        a.update(bad)
        rc = L.bpg_trim_expert_inputs(P(t(4, 87)), P(t(4, 36)), P(t(4, 36)), a['n'], a['dimo_in'], a['dimg_in'], a['dimo_out'],
                                      a['dimg_out'], 4, 1, P(t(4, 40, dt=torch.float64)), P(t(4, 16)), P(t(4, 16)),
                                      None if a['flag'] is None else P(a['flag']), 0, None)
        assert rc == L_.ERR_ARG, bad


@pytest.mark.parametrize('name,is_variation', [('BlocksTouchVariation', True), ('BlocksTouchChoose', False)])
def test_golden_pg_rollout_with_an_expert_in_charge(name, is_variation, monkeypatch):
    """policy_gradient/rollout.py:174-290 with an expert (:196-209): which of expert / policy acts is the reference's
    coin flip (replayed from the fixture: the reference's placement loops consume the same global stream, quirk Q7),
    the expert is handed trim(o, g, ag), its RAW action is stored, sigma is float64 zeros, G is the reference's."""
    env_id = name + '-v0'
    d = np.load(os.path.join(GOLDEN, 'pg_expert_rollout_%s.npz' % name))
    B, T = int(d['B']), int(d['T'])
    dimo, dimg = (int(x) for x in d['expert_dims'])
    state = {}

    class Policy(_NamedPolicy):
        def get_actions(self, o, ag, g, exploit=False):
            u = state['policy_u'].pop(0)
            return u, u.copy(), np.ones_like(u)               # sigma is dropped when there is an expert (:207-209)

    class Expert:
        def __init__(self):
            self.dimo, self.dimg = dimo, dimg

        def get_actions(self, o, ag, g, compute_raw=False):
            assert compute_raw
            k = state['seen']
            state['seen'] += 1
            assert o.shape == (B, dimo) and o.dtype == state['saw_o'].dtype
            np.testing.assert_array_equal(ag, state['saw_ag'][k])
            np.testing.assert_array_equal(g, state['saw_g'][k])
            assert ag.dtype == np.float32 and g.dtype == np.float32
            state['o_seen'].append(o)
            u = state['expert_u'][k]
            return u, u * np.float32(0.5)

    class Log:
        lines = []

        def info(self, msg):
            self.lines.append(msg)

    src = bpg.ScriptedContacts()
    w = bpg.PGRolloutStudent(lambda: bpg.make(env_id, device=DEV), Policy(env_id), Expert(), dims_of(env_id), Log(), T,
                             rollout_batch_size=B, gamma=1. - 1. / T, beta_final=0.35, annealing_coeff=2.0)
    w.env.contact_source = src
    for r in range(int(d['n_rollouts'])):
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        src.pairs = np.ascontiguousarray(con.transpose(1, 0, 2, 3))
        src.ncon = np.ascontiguousarray(d['r%d_ncon' % r].T)
        src.t = 0
        flips = [0.0 if c else 1.0 for c in d['r%d_expert_in_charge' % r]]
        monkeypatch.setattr(np.random, 'rand', lambda: flips.pop(0))
        state.update(policy_u=list(d['r%d_policy_u' % r]), expert_u=d['r%d_expert_u' % r], seen=0, o_seen=[],
                     saw_o=d['r%d_expert_saw_o' % r], saw_ag=d['r%d_expert_saw_ag' % r], saw_g=d['r%d_expert_saw_g' % r])
        assert w.beta() == float(d['r%d_beta' % r])
        ep = w.generate_rollouts(num_grey=d['r%d_num_objs' % r] - 4)
        w.anneal()
        assert Log.lines[-1] == 'Beta = {}'.format(w.beta())          # policy_gradient/rollout.py:292-293
        assert not flips and not state['policy_u'] and state['seen'] == len(state['expert_u'])
        np.testing.assert_array_equal(ep['u'], d['r%d_u' % r])
        np.testing.assert_array_equal(ep['G'].view(np.uint64), d['r%d_G' % r].view(np.uint64))
        np.testing.assert_array_equal(ep['ag'], d['r%d_ag' % r])
        np.testing.assert_array_equal(ep['info_is_success'], d['r%d_info_is_success' % r])
        assert ep['sigma'].dtype == d['r%d_sigma' % r].dtype and not ep['sigma'].any() and ep['sigma'].shape == d['r%d_sigma' % r].shape
        assert w.success_history[-1] == pytest.approx(float(d['r%d_success_rate' % r]), abs=1e-7)
        # what the expert saw of the observation: the restatement applied to this episode's own o (no physics
        # backend behind the contacts-in boundary, so the feature values are the env's, not the fixture's)
        for k, t in enumerate(np.flatnonzero(d['r%d_expert_in_charge' % r])):
            want = O.trim(ep['o'][:, t], ep['g'][:, t], ep['ag'][:, t], dimo, dimg, is_variation)[0]
            np.testing.assert_array_equal(state['o_seen'][k], want)


def test_k_rollouts_in_one_launch_at_4m_envs_with_rows_a_power_of_two_apart():
    """The launch shape of 4 M envs with contiguous rows (2 envs per lane x 256 lanes, uncapped grid) and the
    generator's draws: one launch of K = 2 rollouts == two launches, final state and counts included."""
    B, K, T = 1 << 22, 2, 10
    src = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=9)
    masks = torch.empty((K * T, B), dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_synth_masks_ld(src._h, 0, K * T, P(masks), B, None))
    del src
    one = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=2)
    many = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV, seed=2)
    f1 = torch.empty((K * T, B), dtype=torch.uint8, device=DEV)
    fk = torch.empty((K * T, B), dtype=torch.uint8, device=DEV)
    r1 = torch.empty((K * T, B), dtype=torch.float32, device=DEV)
    rk = torch.empty((K * T, B), dtype=torch.float32, device=DEV)
    c1, ck = torch.zeros(K, dtype=torch.int64, device=DEV), torch.zeros(K, dtype=torch.int64, device=DEV)
    for k in range(K):
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_masks=masks[k * T:].data_ptr(), ld_masks=B, ld_out=B,
                                 d_reward_tb=r1[k * T:].data_ptr(), d_flags_tb=f1[k * T:].data_ptr(), d_success_count=c1[k:].data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(one._h, C.byref(a), None))
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_masks=masks.data_ptr(), ld_masks=B, ld_out=B,
                             d_reward_tb=rk.data_ptr(), d_flags_tb=fk.data_ptr(), d_success_count=ck.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(many._h, C.byref(a), None))
    torch.cuda.synchronize()
    assert torch.equal(c1, ck) and torch.equal(f1, fk) and torch.equal(r1.view(torch.int32), rk.view(torch.int32))
    assert [int((fk[k * T + T - 1] & 1).sum().item()) for k in range(K)] == ck.tolist()
    for key, v in one.state_dict().items():
        assert torch.equal(v, many.state_dict()[key]), key


def test_graphed_pg_rollout_equals_eager():
    """PGRolloutStudent.generate_rollouts_graphed: reset, T x (policy, step + reward row), the return kernel in one
    hipGraph == the eager loop, replay after replay (fresh explicit draws per replay), G bit for bit."""
    env_id, B, T = 'BlocksTouchVariation-v0', 2048, 50
    rng = np.random.RandomState(6)
    masks = i32(legal_masks(rng, T, B, 4, density=3))

    class Policy:
        device_tensors = True
        kwargs = {'info': {'env_name': env_id}}

        def get_actions(self, o, ag, g, exploit=False):
            u = torch.tanh(ag[:, :4] * 0.5 + o[:, :4]) * (0.5 if exploit else 1.0)
            return u, u * 2.0, torch.abs(u) + 1.0

    def worker():
        src = bpg.ScriptedContacts(masks=masks)
        w = bpg.PGRolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src), Policy(), None,
                                 dims_of(env_id), None, T, rollout_batch_size=B, gamma=1. - 1. / T, output='torch')
        return w, src
    wg, _ = worker()
    we, src_e = worker()
    for r, exploit in enumerate((False, False, True, True)):          # a change of `exploit` re-captures
        ng = rng.randint(3, size=B).astype(np.int32)
        # the global NumPy stream: the reference draws np.random.rand() once per step (policy_gradient/rollout.py:196);
        # capture (r = 0, 2) and replay (r = 1, 3) must both leave it where the eager rollout leaves it
        np.random.seed(100 + r)
        epg = wg.generate_rollouts_graphed(exploit=exploit, num_grey=ng)
        after_graphed = np.random.rand()
        src_e.t = 0
        np.random.seed(100 + r)
        epe = we.generate_rollouts(exploit=exploit, num_grey=ng)
        assert np.random.rand() == after_graphed
        np.random.seed(100 + r)
        np.random.rand(T)
        assert np.random.rand() == after_graphed                       # = T draws of the reference's loop
        assert set(epg) == set(epe) and 'G' in epg and 'sigma' in epg
        if r == 3:                                   # ... and with the per-env episode return the step launches accumulate
            src_e.t = 0
            epr = we.generate_rollouts(exploit=exploit, num_grey=ng, episode_return=True)
            assert torch.equal(epr['G'], epe['G'])
            wf = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), None, dims_of(env_id), None,
                                    T, rollout_batch_size=B, output='torch')
            epf = wf.generate_rollouts_fused(masks=masks, num_grey=ng, episode_return=True)
            assert torch.equal(epr['episode_return'].view(torch.int32), epf['episode_return'].view(torch.int32))
            epgr = wg.generate_rollouts_graphed(exploit=exploit, num_grey=ng, episode_return=True)
            assert torch.equal(epgr['episode_return'].view(torch.int32), epf['episode_return'].view(torch.int32))
        for k in epe:
            assert epg[k].dtype == epe[k].dtype and epg[k].shape == epe[k].shape, k
            assert torch.equal(epg[k], epe[k]), k
        assert wg.success_history[-1] == we.success_history[-1]
        # the oracle's return on the rewards the episode implies (-0.0 on steps whose success condition held)
        G = epg['G'].cpu().numpy()
        assert G.dtype == np.float64 and (G <= 0).all()
    with pytest.raises(NotImplementedError):
        w2, _ = worker()
        w2.expert = object()
        w2.generate_rollouts_graphed()


def test_trim_with_rows_that_start_off_a_16_byte_boundary():
    """The staged observation rows use 16-byte loads only when the array starts on a 16-byte boundary."""
    n = 333
    rng = np.random.RandomState(3)
    o = rng.standard_normal((n, 87)).astype(np.float32)
    for j in range(4):
        o[:, 11 + 19 * j + 15:11 + 19 * j + 19] = [0, 0, 1, 0] if j in (1, 3) else [1, 0, 0, 0]
    g = rng.standard_normal((n, 36)).astype(np.float32)
    ag = rng.standard_normal((n, 36)).astype(np.float32)
    want = O.trim(o, g, ag, 40, 16, True)
    for off in (1, 2, 3):
        dev = lambda a: torch.cat([torch.zeros(off, device=DEV), torch.as_tensor(a, device=DEV).reshape(-1)])[off:].view(a.shape)   # noqa: E731
        do, dg, dag = dev(o), dev(g), dev(ag)
        assert do.data_ptr() % 16 != 0
        o_ = torch.empty((n, 40), dtype=torch.float64, device=DEV)
        g_, ag_ = torch.empty((n, 16), device=DEV), torch.empty((n, 16), device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_trim_expert_inputs(P(do), P(dg), P(dag), n, 87, 36, 40, 16, 4, 1, P(o_), P(g_), P(ag_), P(flag), 0, None))
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        for got, exp in zip((o_, g_, ag_), want):
            np.testing.assert_array_equal(got.cpu().numpy(), exp)


def test_random_rollouts_example_runs():
    """examples/random_rollouts.py: the gym-style single env, the batched env, both rollout workers' call forms and
    the evaluator loop in one launch -- as a user would run it."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'random_rollouts.py'), '4096'], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'evaluator: 10 rollouts in one launch' in out.stdout and 'fused success rate' in out.stdout


def test_device_resident_expert_gets_trimmed_device_tensors_without_a_sync_per_step():
    """output='torch': the expert is handed device tensors (o_ float64 for Variation), the bad-row flag of trim is read
    once at the end of the rollout -- and a row that breaks the expert's assumption still raises ValueError."""
    env_id, B, T = 'BlocksTouchVariation-v0', 300, 50
    seen = []

    class Policy(_NamedPolicy):
        device_tensors = True

        def get_actions(self, o, ag, g, exploit=False):
            u = torch.zeros((o.shape[0], 4), device=o.device)
            return u, u, u

    class Expert:
        dimo, dimg = 40, 16

        def get_actions(self, o, ag, g, compute_raw=False):
            assert o.is_cuda and o.dtype == torch.float64 and o.shape == (B, 40)
            assert ag.is_cuda and ag.dtype == torch.float32 and ag.shape == (B, 16) and g.shape == (B, 16)
            seen.append(o)
            u = torch.zeros((B, 4), device=o.device)
            return u, u

    w = bpg.PGRolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, seed=1), Policy(env_id), Expert(),
                             dims_of(env_id), None, T, rollout_batch_size=B, gamma=0.98, beta_final=1.0, annealing_coeff=1.0,
                             output='torch')
    ep = w.generate_rollouts()
    assert len(seen) == T and ep['G'].shape == (B, T) and ep['sigma'].dtype == torch.float64
    # the expert saw the trimmed form of the episode's own observations
    want = O.trim(ep['o'][:, 3].cpu().numpy(), ep['g'][:, 3].cpu().numpy(), ep['ag'][:, 3].cpu().numpy(), 40, 16, True)[0]
    np.testing.assert_array_equal(seen[3].cpu().numpy(), want)
    # a third coloured block in some row: the deferred form raises nothing at the call, the flag stays set for the read
    # at the end of the rollout; the immediate form raises NumPy's ValueError
    o = ep['o'][:, 0].clone()
    o[7, 11 + 19 * 2 + 15:11 + 19 * 2 + 19] = torch.tensor([0., 0., 0., 1.], device=DEV)
    w._trim_bad.zero_()
    w.trim(o, ep['g'][:, 0], ep['ag'][:, 0], 40, 16, _defer_check=True)
    w.trim(ep['o'][:, 0], ep['g'][:, 0], ep['ag'][:, 0], 40, 16, _defer_check=True)      # a good call does not clear it
    assert int(w._trim_bad.item()) == 1
    with pytest.raises(ValueError):
        w.trim(o, ep['g'][:, 0], ep['ag'][:, 0], 40, 16)


@pytest.mark.parametrize('B,K,T', [(1, 4096, 1), (3, 2, 1023), (65, 7, 64), (2, 3, 65)])
def test_k_rollouts_at_the_edges_of_the_argument_ranges_against_the_oracle(B, K, T):
    """One env, the largest K, the longest horizon (the elapsed counter saturates at 1023), odd tiny batches: every
    rollout of the launch against one oracle env per slot (reset with the r-th draw, T steps on the r-th rows)."""
    env_id = 'BlocksTouchVariation-v0'
    rng = np.random.RandomState(B * 7 + K)
    ng = rng.randint(3, size=(K, B)).astype(np.int32)
    masks = np.concatenate([legal_masks(rng, T, B, 4 + ng[k], density=2) for k in range(K)])
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=1, max_episode_steps=50)
    dm, dng = i32(masks), torch.as_tensor(ng, device=DEV)
    rew = torch.empty((K * T, B), dtype=torch.float32, device=DEV)
    flg = torch.empty((K * T, B), dtype=torch.uint8, device=DEV)
    cnt = torch.zeros(K, dtype=torch.int64, device=DEV)
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_masks=dm.data_ptr(), d_reward_tb=rew.data_ptr(), d_flags_tb=flg.data_ptr(),
                             d_num_grey=dng.data_ptr(), d_success_count=cnt.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    torch.cuda.synchronize()
    rew, flg = rew.cpu().numpy().view(np.uint32), flg.cpu().numpy()
    oenvs = [O.OracleEnv(env_id) for _ in range(B)]
    for k in (range(K) if K <= 8 else (0, 1, K // 2, K - 1)):
        succ = 0
        for b, e in enumerate(oenvs):
            e.reset(num_grey=int(ng[k, b]))
            for t in range(T):
                ag, r, done, ok = e.step_mask(int(masks[k * T + t, b]))
                assert rew[k * T + t, b] == np.float32(r).view(np.uint32)
                assert (flg[k * T + t, b] & 3) == (int(ok) | (int(done) << 1)), (k, t, b)
            succ += int(e.e.has_succeeded)
        assert int(cnt[k].item()) == succ


def test_step_with_in_kernel_contacts_equals_the_same_contacts_read_from_memory():
    """bpg_step_synth (contacts drawn inside the step launch, keyed by seed, env and the env's elapsed step) ==
    bpg_synth_masks for the same steps fed through bpg_step_masks: state, reward bits and flags, over a reset;
    bpg_num_envs reports the handle's batch size."""
    env_id, B, T = 'BlocksTouchVariation-v0', 70001, 12
    a = bpg.make(env_id, num_envs=B, device=DEV, seed=21)
    b = bpg.make(env_id, num_envs=B, device=DEV, seed=21)
    assert int(L.bpg_num_envs(a._h)) == B
    ng = torch.as_tensor(np.random.RandomState(0).randint(3, size=B).astype(np.int32), device=DEV)
    for episode in range(2):
        a.reset(num_grey=ng), b.reset(num_grey=ng)
        m = torch.empty((T, B), dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_synth_masks(b._h, 0, T, P(m), None))
        for t in range(T):
            ra, fa = torch.empty(B, device=DEV), torch.empty(B, dtype=torch.uint8, device=DEV)
            rb, fb = torch.empty(B, device=DEV), torch.empty(B, dtype=torch.uint8, device=DEV)
            bpg._lib.check(L.bpg_step_synth(a._h, P(ra), P(fa), None))
            bpg._lib.check(L.bpg_step_masks(b._h, P(m[t]), P(rb), P(fb), None))
            assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(fa, fb), (episode, t)
        for key, v in a.state_dict().items():
            assert torch.equal(v, b.state_dict()[key]), key
    assert int(a.success_count().item()) == int((fa & 1).sum().item()) > 0


def test_golden_rollout_with_compute_q_keeps_the_references_mean_q():
    """rollout.py:99-107, :168-169, :182-184: with compute_Q the policy returns (u, Q), the worker appends np.mean(Qs)
    to Q_history -- float32, NumPy's mean over the stacked arrays -- and current_mean_Q / logs average the history."""
    env_id = 'BlocksTouch-v0'
    d = np.load(os.path.join(GOLDEN, 'rollout_Q_BlocksTouch.npz'))
    B, T = int(d['B']), int(d['T'])

    class Replay:
        def get_actions(self, o, ag, g, compute_Q=False, **kw):
            assert compute_Q is True
            u, q = self.u[:, self.t], self.q[self.t]
            self.t += 1
            return u, q

    policy, src = Replay(), bpg.ScriptedContacts()
    w = bpg.RolloutStudent(lambda: bpg.make(env_id, device=DEV, contact_source=src), policy, dims_of(env_id), None, T,
                           rollout_batch_size=B, compute_Q=True)
    w.env.contact_source = src
    for r in range(len(d['modes'])):
        con = d['r%d_contacts' % r].copy()
        con[con < 0] = 0
        src.pairs = np.ascontiguousarray(con.transpose(1, 0, 2, 3))
        src.ncon = np.ascontiguousarray(d['r%d_ncon' % r].T)
        src.masks, src.t = None, 0
        policy.u, policy.q, policy.t = d['r%d_u' % r], d['r%d_Q' % r], 0
        ep = w.generate_rollouts()
        np.testing.assert_array_equal(ep['ag'], d['r%d_ag' % r])
        got, want = w.Q_history[-1], d['r%d_Q_history_last' % r]
        assert np.asarray(got).dtype == want.dtype and np.asarray(got).tobytes() == want.tobytes()
        cm = np.asarray(w.current_mean_Q())
        assert cm.dtype == d['r%d_current_mean_Q' % r].dtype and cm.tobytes() == d['r%d_current_mean_Q' % r].tobytes()
        assert dict(w.logs())['worker/mean_Q'] == cm
