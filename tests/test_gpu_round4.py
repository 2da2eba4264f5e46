"""Round-4 GPU tests (through the C-ABI, bit-exact):

* the 2-envs-per-lane launch shapes of the fused rollout -- 256-lane and 1024-lane blocks, grid-stride trips, the
  bench headline's own instantiation -- compared DIRECTLY with the oracle (O.OracleEnv, the matrix-form restatement of
  fetch_env.py:135-167, :275-281) on env windows that cover lane-group, block and grid-stride boundaries;
* bpg_rollout_args.d_episode_return (the per-env undiscounted return, BASELINE's "aggregated rollout return") in every
  form that produces it: in-loop counter, population of the step-ok word, several rollouts per launch, episode writer;
* struct versioning of bpg_rollout_args, a negative gamma, the lazily allocated step-ok words.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from oracle import bpg_oracle as O
from util import VARIANTS, legal_masks, valid_by_n, pack_oracle_env

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ENV_ID = 'BlocksTouchVariation-v0'
L = bpg._lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731


def bits32(t):
    return t.cpu().numpy().view(np.uint32)


def same_bits64(a, b):
    np.testing.assert_array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def oracle_G(reward_tb, gamma):
    r = np.asarray(reward_tb)
    return np.stack([O.discounted_return(r[:, b].astype(np.float64), gamma) for b in range(r.shape[1])])


def windows(B):
    """Env windows of a 2-envs-per-lane launch: the first lanes, the edge between two blocks (512 / 2048 envs), the edge
    between two grid-stride trips (256 x 1024 x 2 = 1024 x 256 x 2 = 524 288 envs), an odd offset in the middle, the
    last lanes."""
    w = [(0, 128), (2048 - 64, 128), ((B // 2) | 1, 129), (B - 130, 130)]
    if B > 524288 + 64:
        w.append((524288 - 64, 128))
    idx = np.unique(np.concatenate([np.arange(lo, lo + n) for lo, n in w]))
    return idx[(idx >= 0) & (idx < B)]


def fill_device_masks(out, ng, gen, raw_every=7):
    """out (T, B) int32 <- contact masks generated on the device: two random words ANDed (about a quarter of the pairs),
    restricted to the pairs of objects present in the env's episode (ng (B,): num_grey) -- except for the envs with
    index % raw_every == 5, which keep raw masks (pairs of absent objects: the reference keeps those set,
    fetch_env.py:154-157 demotes only i, j < num_objs)."""
    T, B = out.shape
    m = torch.randint(0, 1 << 21, (T, B), device=DEV, generator=gen, dtype=torch.int32)
    m &= torch.randint(0, 1 << 21, (T, B), device=DEV, generator=gen, dtype=torch.int32)
    vb = torch.as_tensor(valid_by_n().astype(np.int32), device=DEV)
    valid = vb[(4 + ng).long()]
    raw = (torch.arange(B, device=DEV) % raw_every) == 5
    out.copy_(torch.where(raw[None, :], m, m & valid[None, :]))


@pytest.mark.parametrize('B,pow2_rows', [(262144, False), (524288 + 2, False), (1048576, False), (4194304, True),
                                         (4194304, False)])
def test_two_envs_per_lane_shapes_directly_against_the_oracle(B, pow2_rows):
    """One batch, four launches in a row, every one compared with the oracle on the windows above -- reward bits, flag
    bytes, the undiscounted episode return, G where asked for, and after each launch the packed state:
      A  reset_first = 1, reward + flags + success count + episode return      (in-loop counter, RET)
      B  reset_first = 0 (steps 51..100 of the same episode, per-lane elapsed counters, done set), reward + flags +
         d_G + episode return                                                    (step-ok words, OKB)
      C  rollouts = 3 in one launch, reward + flags + 3 counts + (3, B) returns  (MULTI + RET)
      D  reset_first = 1, reward + flags + count: the bench headline's instantiation, nothing else asked for.
    Rows are bpg_row_stride(B) apart (B + 256 at 1 M and 4 M envs) or, with pow2_rows, exactly B apart (at 4 M envs that
    is the aliased case, for which rollout_geometry picks 256-lane blocks)."""
    T, K, gamma = 50, 3, 0.98
    gen = torch.Generator(device=DEV).manual_seed(B % 9973 + (1 if pow2_rows else 0))
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=3)
    ld = B if pow2_rows else int(L.bpg_row_stride(B))
    assert ld % 2 == 0
    rows = (1 + 1 + K + 1) * T

    def tm(dtype):
        return torch.empty((rows, ld), dtype=dtype, device=DEV)[:, :B]
    ng = torch.randint(0, 3, (2 + K, B), device=DEV, generator=gen, dtype=torch.int32)      # A/B, C x 3, D
    masks = tm(torch.int32)
    for seg, k in enumerate([0, 0] + list(range(1, 2 + K))):          # launches A, B share an episode
        fill_device_masks(masks[seg * T:(seg + 1) * T], ng[k], gen)
    reward, flags = tm(torch.float32), tm(torch.uint8)
    cnt = torch.zeros(8, dtype=torch.int64, device=DEV)
    ret = torch.full((2 + K, B), 7.0, dtype=torch.float32, device=DEV)
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)

    def launch(row0, **kw):
        a = bpg._lib.RolloutArgs(T=T, d_masks=masks[row0].data_ptr(), d_reward_tb=reward[row0].data_ptr(),
                                 d_flags_tb=flags[row0].data_ptr(), ld_masks=ld, ld_out=ld, **kw)
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        return env.state_now.clone(), env.state_ever.clone()

    states = {}
    states['A'] = launch(0, reset_first=1, d_num_grey=ng[0].data_ptr(), d_success_count=cnt[0:].data_ptr(),
                         d_episode_return=ret[0].data_ptr())
    states['B'] = launch(T, reset_first=0, d_G=G.data_ptr(), gamma=gamma, d_episode_return=ret[1].data_ptr())
    states['C'] = launch(2 * T, reset_first=1, rollouts=K, d_num_grey=ng[1].data_ptr(), d_success_count=cnt[1:].data_ptr(),
                         d_episode_return=ret[2].data_ptr())
    states['D'] = launch((2 + K) * T, reset_first=1, d_num_grey=ng[1 + K].data_ptr(), d_success_count=cnt[4:].data_ptr())
    torch.cuda.synchronize()

    # whole-batch identities on the device: counts against the flags of each rollout's last step, returns against rewards
    for c, row in zip(cnt[:5].tolist(), [T - 1, 3 * T - 1, 4 * T - 1, 5 * T - 1, 6 * T - 1]):   # A, C x 3, D
        assert c == int((flags[row] & 1).sum()) and 0 < c <= B
    for k, row0 in ((0, 0), (1, T), (2, 2 * T), (3, 3 * T), (4, 4 * T)):
        fails = (reward[row0:row0 + T] != 0).sum(0)                                  # -1.0 != 0, -0.0 == 0
        assert torch.equal(ret[k], -(fails.float()))
        assert bool(torch.signbit(ret[k]).all())                                     # -0.0f where nothing failed

    # the oracle on the windows
    idx = windows(B)
    di = torch.as_tensor(idx, device=DEV)
    m_w = masks[:, di].cpu().numpy().view(np.uint32)
    r_w, f_w = bits32(reward[:, di].contiguous()), flags[:, di].cpu().numpy()
    ret_w, ng_w, G_w = ret[:, di].cpu().numpy(), ng[:, di].cpu().numpy(), G[di].cpu().numpy()
    st_w = {k: (v[0][di].cpu().numpy().view(np.uint32), v[1][di].cpu().numpy().view(np.uint32)) for k, v in states.items()}
    exp_r = np.zeros_like(r_w)
    exp_f = np.zeros_like(f_w)
    for j in range(len(idx)):
        e = O.OracleEnv(ENV_ID)
        row = 0

        def steps(n):
            nonlocal row
            for _ in range(n):
                _, r, done, succ = e.step_mask(int(m_w[row, j]))
                exp_r[row, j] = np.float32(r).view(np.uint32)
                exp_f[row, j] = int(succ) | (int(done) << 1) | (int(r == 0) << 2)
                row += 1
        e.reset(num_grey=int(ng_w[0, j]))
        steps(T)
        assert (st_w['A'][0][j], st_w['A'][1][j]) == pack_oracle_env(e), ('A', idx[j])
        steps(T)                                                                     # B: no reset
        assert (st_w['B'][0][j], st_w['B'][1][j]) == pack_oracle_env(e), ('B', idx[j])
        for k in range(K):
            e.reset(num_grey=int(ng_w[1 + k, j]))
            steps(T)
        assert (st_w['C'][0][j], st_w['C'][1][j]) == pack_oracle_env(e), ('C', idx[j])
        e.reset(num_grey=int(ng_w[1 + K, j]))
        steps(T)
        assert (st_w['D'][0][j], st_w['D'][1][j]) == pack_oracle_env(e), ('D', idx[j])
    np.testing.assert_array_equal(r_w, exp_r)
    np.testing.assert_array_equal(f_w, exp_f)
    assert (exp_f[T:2 * T] & 2).all() and not (exp_f[:T - 1] & 2).any()            # launch B runs behind the TimeLimit
    rew_f = exp_r.view(np.float32)
    for k in range(2 + K):
        want = O.episode_return(rew_f[k * T:(k + 1) * T])
        np.testing.assert_array_equal(ret_w[k].view(np.uint32), want.view(np.uint32))
    same_bits64(G_w, oracle_G(rew_f[T:2 * T], gamma))


@pytest.mark.parametrize('env_id', ['GripperTouch-v0', 'BlocksTouchChoose-v0', 'BlocksTouchVariation-v0', 'ToppleTower-v0'])
@pytest.mark.parametrize('B', [1, 1003, 4096, 131072 + 2])
def test_episode_return_of_every_launch_form_vs_oracle(env_id, B):
    """d_episode_return == r[0] + r[1] + ... of the oracle's step rewards (bit-exact, -0.0f when every step succeeded):
    stream path with and without the reward stream, with d_G (taken from the step-ok word), with the synthetic source,
    from the episode writer, and for T = 70 (> 64: no step-ok word exists)."""
    rng = np.random.RandomState(B + len(env_id))
    v = VARIANTS[env_id]
    for T in (50, 70):
        ng = rng.randint(3, size=B).astype(np.int32)
        nobj = 4 + ng if 'Variation' in env_id else np.full(B, v['num_objs'])
        masks = legal_masks(rng, T, B, nobj, density=2)
        # make a few envs succeed at every step and a few never, whatever the goal of the variant
        info = bpg._lib.variant_info(L.bpg_variant_from_id(env_id.encode()))
        masks[:, ::5] |= np.uint32(info.want_touch)
        masks[:, ::5] &= ~np.uint32(info.want_never)
        nB = min(B, 600)                                                  # oracle envs (the rest: device-side identity)
        ref, _ = O.generate_rollouts(env_id, nB, T, masks[:, :nB], num_grey=ng[:nB])
        want = ref['episode_return']
        assert want.dtype == np.float32 and (np.signbit(want)).all()
        if B >= 5:
            assert (want == 0).any() and (want < 0).any()
        dm = torch.as_tensor(masks.view(np.int32), device=DEV)
        dng = torch.as_tensor(ng, device=DEV)

        def run(**kw):
            env = bpg.make(env_id, num_envs=B, device=DEV, seed=1)
            ret = torch.full((B,), 3.0, dtype=torch.float32, device=DEV)
            a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=dng.data_ptr(), d_masks=dm.data_ptr(),
                                     d_episode_return=ret.data_ptr(), **kw)
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
            return ret
        rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
        fl = torch.empty((T, B), dtype=torch.uint8, device=DEV)
        G = torch.empty((B, T), dtype=torch.float64, device=DEV)
        ag = torch.empty((B, T + 1, v['dimg']), dtype=torch.float32, device=DEV)
        forms = [dict(), dict(d_reward_tb=rew.data_ptr()), dict(d_flags_tb=fl.data_ptr()),
                 dict(d_reward_tb=rew.data_ptr(), d_flags_tb=fl.data_ptr()), dict(d_ep_ag=ag.data_ptr())]
        if T <= 64:
            forms += [dict(d_G=G.data_ptr(), gamma=0.98), dict(d_G=G.data_ptr(), gamma=0.98, d_ep_ag=ag.data_ptr())]
        else:
            forms += [dict(d_G=G.data_ptr(), gamma=0.99, d_reward_tb=rew.data_ptr())]
        for kw in forms:
            got = run(**kw)
            np.testing.assert_array_equal(got[:nB].cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=str(sorted(kw)))
            if 'd_reward_tb' in kw:
                assert torch.equal(got, -((rew != 0).sum(0).float()))
    # synthetic source: the return equals the sum of the reward stream the same launch wrote
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=5)
    T = 50
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    ret = torch.empty(B, dtype=torch.float32, device=DEV)
    for reset_first in (1, 0):
        a = bpg._lib.RolloutArgs(T=T, reset_first=reset_first, d_reward_tb=rew.data_ptr(), d_episode_return=ret.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        assert torch.equal(ret, -((rew != 0).sum(0).float())) and bool(torch.signbit(ret).all())


def test_worker_returns_the_episode_return():
    """RolloutStudent.generate_rollouts_fused(episode_return=True) / evaluate_rollouts_fused(episode_return=(K, B)): the
    worker-level view of the same array."""
    B, T, K = 3000, 50, 4
    rng = np.random.RandomState(2)
    w = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=DEV), None,
                           dict(o=87, u=4, g=36, info_is_success=1), None, T, rollout_batch_size=B)
    ng = rng.randint(3, size=(K, B)).astype(np.int32)
    masks = np.concatenate([legal_masks(rng, T, B, 4 + ng[k], density=2) for k in range(K)])
    ep = w.generate_rollouts_fused(masks=torch.as_tensor(masks[:T].view(np.int32), device=DEV), num_grey=ng[0],
                                   want_reward=True, episode_return=True)
    ref, _ = O.generate_rollouts(ENV_ID, B, T, masks[:T], num_grey=ng[0])
    np.testing.assert_array_equal(ep['episode_return'].view(np.uint32), ref['episode_return'].view(np.uint32))
    np.testing.assert_array_equal(ep['reward_tb'].view(np.uint32), ref['reward'].view(np.uint32))
    ret = torch.empty((K, B), dtype=torch.float32, device=DEV)
    rates = w.evaluate_rollouts_fused(torch.as_tensor(masks.view(np.int32), device=DEV), num_grey=ng, episode_return=ret)
    for k in range(K):
        ref, _ = O.generate_rollouts(ENV_ID, B, T, masks[k * T:(k + 1) * T], num_grey=ng[k])
        np.testing.assert_array_equal(ret[k].cpu().numpy().view(np.uint32), ref['episode_return'].view(np.uint32))
        assert rates[k] == pytest.approx(ref['success_rate'], abs=1e-6)


def test_rollout_args_struct_versioning():
    """struct_size (include/bpgym.h): unset -> BPG_ERR_ARG; a caller built before d_episode_return existed (120 bytes)
    works and the field is taken as NULL even if the bytes behind its struct are garbage; a caller with MORE bytes than
    this library knows is accepted only if they are zero."""
    B, T = 512, 10
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
    fl = torch.zeros((T, B), dtype=torch.uint8, device=DEV)
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_flags_tb=fl.data_ptr())
    assert a.struct_size == bpg._lib.args_tag(C.sizeof(bpg._lib.RolloutArgs))
    a.struct_size = 0
    assert L.bpg_rollout_v2(env._h, C.byref(a), None) == bpg._lib.ERR_ARG and b'struct_size' in L.bpg_last_error()
    # a struct of a layout without the tagged size word (the round-3 one had T there: 120, 128, ... passed as sizes in round 4)
    for stale in (120, 128, 136):
        a.struct_size = stale
        assert L.bpg_rollout_v2(env._h, C.byref(a), None) == bpg._lib.ERR_ARG
    a.struct_size = bpg._lib.args_tag(bpg._lib.ROLLOUT_ARGS_MIN_SIZE)
    a.d_episode_return = 0xdead0000                      # behind the caller's struct: must not be read
    assert L.bpg_rollout_v2(env._h, C.byref(a), None) == 0
    torch.cuda.synchronize()
    assert int((fl[T - 1] & 2).sum()) == 0 and int(fl.sum()) >= 0

    class Bigger(C.Structure):
        _fields_ = [('base', bpg._lib.RolloutArgs), ('future_field', C.c_void_p)]
    big = Bigger()
    big.base = bpg._lib.RolloutArgs(T=T, reset_first=1, d_flags_tb=fl.data_ptr())
    big.base.struct_size = bpg._lib.args_tag(C.sizeof(Bigger))
    fn = L.bpg_rollout_v2
    rc = fn(env._h, C.cast(C.pointer(big), C.POINTER(bpg._lib.RolloutArgs)), None)
    assert rc == 0
    big.future_field = 1
    rc = fn(env._h, C.cast(C.pointer(big), C.POINTER(bpg._lib.RolloutArgs)), None)
    assert rc == bpg._lib.ERR_UNSUPPORTED
    torch.cuda.synchronize()


@pytest.mark.parametrize('gamma', [-0.5, -0.0, 1e200, 0.0, 1.0, 1.5])
def test_return_for_gammas_outside_the_subtractable_set(gamma):
    """With gamma < 0 (or -0.0) the reference's `+= gamma**k * r` turns -0.0 accumulators into +0.0, with overflowing
    powers inf * -0.0 is NaN: the conditional-subtract shortcut does not apply, and both entry points must take the
    general product + sum.  bpg_rollout then needs the reward stream for d_G."""
    T, B = 50, 64 * 3 + 5
    rng = np.random.RandomState(4)
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=2)
    ng = rng.randint(3, size=B).astype(np.int32)
    masks = legal_masks(rng, T, B, 4 + ng, density=2)
    masks[:, ::4] |= np.uint32(env.info.want_touch)
    dm, dng = torch.as_tensor(masks.view(np.int32), device=DEV), torch.as_tensor(ng, device=DEV)
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)
    with np.errstate(over='ignore'):
        subtractable = gamma >= 0 and not np.signbit(gamma) and np.isfinite(np.float64(gamma) ** (T - 1))
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=dng.data_ptr(), d_masks=dm.data_ptr(), d_G=G.data_ptr(), gamma=gamma)
    rc = L.bpg_rollout_v2(env._h, C.byref(a), None)
    assert (rc == 0) == bool(subtractable), (gamma, rc)
    a.d_reward_tb = rew.data_ptr()
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    with np.errstate(all='ignore'):
        want = oracle_G(rew.cpu().numpy(), gamma)
    for got in (G.cpu().numpy(), bpg.discounted_return(rew, gamma).cpu().numpy()):
        # an invalid operation yields the hardware's default NaN (sign bit set on x86, clear on gfx950): NaNs are
        # compared by position, everything else bit for bit
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        same_bits64(np.where(np.isnan(want), 0.0, got), np.where(np.isnan(want), 0.0, want))
    if gamma == -0.5:
        assert (want == 0).any() and not np.signbit(want[want == 0]).all()       # the +0.0 the shortcut would have missed
    if gamma == 1e200:
        assert np.isnan(want).any()


def test_step_ok_words_are_allocated_on_first_use_only():
    """A handle that never asks for d_G holds 8 bytes per env (now + ever); the step-ok words (8 more) appear with the
    first d_G call."""
    B = 1 << 22
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    h = C.c_void_p()
    bpg._lib.check(L.bpg_create(C.byref(h), L.bpg_variant_from_id(ENV_ID.encode()), B, 0, 50, 1, 0))

    class Raw:
        _h = h
    env = Raw()
    held = free0 - torch.cuda.mem_get_info()[0]
    assert 8 * B <= held < 8 * B + (24 << 20), held
    a = bpg._lib.RolloutArgs(T=10, reset_first=1)
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    assert free0 - torch.cuda.mem_get_info()[0] == held
    G = torch.empty((B, 10), dtype=torch.float64, device=DEV)
    free1 = torch.cuda.mem_get_info()[0]
    a = bpg._lib.RolloutArgs(T=10, reset_first=1, d_G=G.data_ptr(), gamma=0.9)
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    torch.cuda.synchronize()
    assert 8 * B <= free1 - torch.cuda.mem_get_info()[0] < 8 * B + (8 << 20)
    del G
    bpg._lib.check(L.bpg_destroy(h))


def test_compute_reward_precision_follows_numpys_promotion():
    """bpg.compute_reward picks the arithmetic NumPy would use for `achieved_goal * goal` (fetch_env.py:141):
    float32 x int8 / bool stays float32, float32 x int64 / float64 is float64 -- on rows built so that the two
    precisions DISAGREE about `sum == count` (a float64 call is never narrowed to float32)."""
    rng = np.random.RandomState(5)
    rows = []
    g8 = np.array([3, 3, 3, 0, 0, 0, 0, 0], np.int8)       # three non-zero entries: count_nonzero(g) = 3
    while len(rows) < 40:                                   # d == c in float32 arithmetic, d != c in float64 (or vice versa)
        x = rng.rand(3).astype(np.float32)
        x[2] = np.float32(1.0) - (x[0] + x[1])
        ag = np.concatenate([x, rng.rand(5).astype(np.float32)])
        d32 = np.sum(ag * g8.astype(np.float32))
        d64 = np.sum(ag.astype(np.float64) * g8)
        if (d32 == 3) != (d64 == 3):
            rows.append(ag)
    ag = np.stack(rows)
    g = np.tile(g8.astype(np.int64), (len(rows), 1))
    g[::3, 2] = 0                                           # some rows with another count: != in both precisions
    def numpy_says(a, b):                                   # the reference's expression, NumPy's own promotion
        d = np.sum(a * b, axis=-1)
        c = np.count_nonzero(b, axis=-1)
        return -(d != c).astype(np.float32)
    seen = set()
    for gdt in (np.int8, np.bool_, np.float32, np.int64, np.float64, np.int32):
        for adt in (np.float32, np.float64):
            a, b = ag.astype(adt), g.astype(gdt)
            want = numpy_says(a, b)
            got = bpg.compute_reward(a, b)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg='%s x %s' % (adt, gdt))
            gt = bpg.compute_reward(torch.as_tensor(a, device=DEV), torch.as_tensor(b, device=DEV))
            np.testing.assert_array_equal(bits32(gt), want.view(np.uint32))
            seen.add(want.tobytes())
    assert len(seen) >= 2                                   # the two precisions really answered differently (bool goals: a third)
    with pytest.raises(bpg._lib.BpgError):                  # the header's range, len <= 256, in either precision
        bpg.compute_reward(np.zeros((4, 257)), np.zeros((4, 257)))
    assert bpg.compute_reward(np.zeros((4, 160), np.float32), np.zeros((4, 160), np.int8)).shape == (4,)
    with pytest.raises(NotImplementedError):
        bpg.compute_reward(np.zeros((2, 8), np.float16), np.zeros((2, 8), np.float16))


def test_async_stats_worker_never_waits_and_reports_the_same_history():
    """RolloutStudent(async_stats=True): generate_rollouts* append a lazy entry to success_history instead of reading the
    success count back (no host sync per rollout); the entries resolve on first read to exactly what a synchronous worker
    appended -- fused, per-step eager and hipGraph-replayed rollouts, across the wrap of the ring of pinned slots."""
    B, T = 4096, 50
    rng = np.random.RandomState(11)
    dims = dict(o=87, u=4, g=36, info_is_success=1)

    class Pol:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return ag[:, :4] * 0.25

    def worker(**kw):
        return bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=DEV, seed=4), Pol(), dims, None, T,
                                  rollout_batch_size=B, output='torch', history_len=5, **kw)
    ws, wa = worker(), worker(async_stats=True)
    assert wa._pinned.is_pinned() and wa._pinned.numel() == 64
    n = 0
    for rep in range(30):                       # 150 rollouts: the 64 pinned slots wrap twice, the history (5) many times
        ng = rng.randint(3, size=B).astype(np.int32)
        masks = torch.as_tensor(legal_masks(rng, T, B, 4 + ng, density=3 + rep % 2).view(np.int32), device=DEV)
        for w in (ws, wa):
            w.generate_rollouts_fused(masks=masks, num_grey=ng)
            w.generate_rollouts(num_grey=ng)
            w.generate_rollouts_graphed(num_grey=ng)
            w.generate_rollouts_graphed(num_grey=ng)
            w.generate_rollouts_fused(masks=masks, num_grey=ng, episode_return=True)
        n += 5
        if rep % 7 == 3:                        # a reader in the middle (np.mean, as logs() / current_success_rate do)
            assert wa.current_success_rate() == ws.current_success_rate()
            assert all(isinstance(x, float) for x in wa.success_history)
    from collections import deque
    unresolved = [type(x).__name__ for x in list(deque.__iter__(wa.success_history))]
    assert '_LazyRate' in unresolved            # the last rollouts have not been read back yet
    assert list(wa.success_history) == list(ws.success_history) and len(wa.success_history) == 5
    assert wa.success_history[-1] == ws.success_history[-1] and wa.logs() == ws.logs()
    assert wa.n_episodes == ws.n_episodes == n * B
    assert 0 < np.mean(ws.success_history) <= 1


@pytest.mark.parametrize('env_id', ['BlocksTouch-v0', 'BlocksTouchVariation-v0'])
def test_per_step_rollouts_accumulate_the_same_episode_return(env_id):
    """generate_rollouts / generate_rollouts_graphed (episode_return=True): the step launches add each step's reward to a
    per-env accumulator (bpg_step_args.d_return_acc) -- the same bits as the fused launch's d_episode_return and as the
    oracle's r[0] + r[1] + ...; bpg_step_args versioning."""
    B, T = 2051, 50
    rng = np.random.RandomState(9)
    v = VARIANTS[env_id]
    dims = dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)

    class Pol:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return ag[:, :4] * 0.25
    for rep in range(3):
        ng = rng.randint(3, size=B).astype(np.int32)
        nobj = 4 + ng if 'Variation' in env_id else np.full(B, v['num_objs'])
        masks = legal_masks(rng, T, B, nobj, density=2)
        info = bpg._lib.variant_info(L.bpg_variant_from_id(env_id.encode()))
        masks[:, ::6] |= np.uint32(info.want_touch)                      # some envs succeed at every step: -0.0f
        ref, _ = O.generate_rollouts(env_id, B, T, masks, num_grey=ng)
        want = ref['episode_return'].view(np.uint32)
        assert (ref['episode_return'] == 0).any() and np.signbit(ref['episode_return']).all()
        if rep == 0:
            dm = torch.empty((T, B), dtype=torch.int32, device=DEV)      # ONE script buffer: a captured graph holds its rows
            ws = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV,
                                                              contact_source=bpg.ScriptedContacts(masks=dm)),
                                    Pol(), dims, None, T, rollout_batch_size=B, output='torch')
        dm.copy_(torch.as_tensor(masks.view(np.int32), device=DEV))
        for fn in (ws.generate_rollouts, ws.generate_rollouts_graphed):
            ws.env.contact_source.t = 0
            if env_id == 'BlocksTouch-v0':
                # quirk Q1: the base classes keep the matrix over resets -- start every comparison from a fresh state
                ws.env.load_state_dict(bpg.make(env_id, num_envs=B, device=DEV).state_dict())
            ep = fn(num_grey=ng, episode_return=True)
            np.testing.assert_array_equal(bits32(ep['episode_return']), want)
            np.testing.assert_array_equal(ep['info_is_success'].cpu().numpy(), ref['info_is_success'])
            assert 'episode_return' not in fn(num_grey=ng)               # the other graph / no accumulator launches
    # struct versioning of bpg_step_args: unset size refused; unknown trailing bytes must be zero
    env = bpg.make(env_id, num_envs=64, device=DEV)
    a = bpg._lib.StepArgs()
    assert a.struct_size == bpg._lib.args_tag(C.sizeof(bpg._lib.StepArgs)) == bpg._lib.args_tag(bpg._lib.STEP_ARGS_MIN_SIZE)
    assert L.bpg_step_record_v2(env._h, C.byref(a), None) == 0
    a.struct_size = 0
    assert L.bpg_step_record_v2(env._h, C.byref(a), None) == bpg._lib.ERR_ARG

    class Bigger(C.Structure):
        _fields_ = [('base', bpg._lib.StepArgs), ('future_field', C.c_void_p)]
    big = Bigger()
    big.base.struct_size = bpg._lib.args_tag(C.sizeof(Bigger))
    ptr = C.cast(C.pointer(big), C.POINTER(bpg._lib.StepArgs))
    assert L.bpg_step_record_v2(env._h, ptr, None) == 0
    big.future_field = 5
    assert L.bpg_step_record_v2(env._h, ptr, None) == bpg._lib.ERR_UNSUPPORTED
    torch.cuda.synchronize()


@pytest.mark.parametrize('total,S', [(1048576, 4), (524288 + 6, 2), (1003, 3)])
def test_stream_shards_of_one_gpu_equal_one_batch(total, S):
    """distributed.StreamShards: S env shards on one GPU, each on its own stream, keyed by global env index -- their
    rollouts (launched side by side, nothing synchronises them between rollouts) write the slices of ONE set of arrays
    that a single `total`-env handle writes: reward, flags, episode return, success counts, final state."""
    from blockpuzzle_gym_amd import distributed as D
    T = 50
    one = bpg.make(ENV_ID, num_envs=total, device=DEV, seed=21)
    sh = D.StreamShards(ENV_ID, total, S, device=DEV, seed=21)
    assert len(sh) == S and sum(e.num_envs for e, _ in sh.shards) == total
    ld = int(L.bpg_row_stride(total))
    masks = torch.empty((T, ld), dtype=torch.int32, device=DEV)[:, :total]
    ng = (torch.arange(total, device=DEV, dtype=torch.int32) * 5) % 3
    outs = []
    for who in ('one', 'shards'):
        rew = torch.empty((T, ld), dtype=torch.float32, device=DEV)[:, :total]
        flg = torch.empty((T, ld), dtype=torch.uint8, device=DEV)[:, :total]
        ret = torch.empty(total, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(3 * max(S, 1), dtype=torch.int64, device=DEV)
        for it in range(3):                          # three rollouts back to back: the shards never wait for each other
            one.seed(21 + it)
            bpg._lib.check(L.bpg_synth_masks_ld(one._h, 0, T, P(masks), ld, None))
            torch.cuda.synchronize()
            if who == 'one':
                a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(), ld_masks=ld, ld_out=ld,
                                         d_reward_tb=rew.data_ptr(), d_flags_tb=flg.data_ptr(), d_episode_return=ret.data_ptr(),
                                         d_success_count=cnt[it:].data_ptr())
                bpg._lib.check(L.bpg_rollout_v2(one._h, C.byref(a), None))
            else:
                def go(i, env, st):
                    lo = sh.first[i]
                    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng[lo:].data_ptr(), d_masks=masks[:, lo:].data_ptr(),
                                             ld_masks=ld, ld_out=ld, d_reward_tb=rew[:, lo:].data_ptr(), d_flags_tb=flg[:, lo:].data_ptr(),
                                             d_episode_return=ret[lo:].data_ptr(), d_success_count=cnt[it * S + i:].data_ptr())
                    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), C.c_void_p(st.cuda_stream)))
                sh.fork()                            # the masks were written on the current stream
                sh.for_each(go)
                sh.join()
            torch.cuda.synchronize()
        if who == 'one':
            outs.append((rew.clone(), flg.clone(), ret.clone(), cnt[:3].clone(), one.state_now.clone(), one.state_ever.clone()))
        else:
            now = torch.cat([e.state_now for e, _ in sh.shards])
            ever = torch.cat([e.state_ever for e, _ in sh.shards])
            outs.append((rew.clone(), flg.clone(), ret.clone(), cnt[:3 * S].reshape(3, S).sum(1), now, ever))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.float32 else a, b.view(torch.uint8) if b.dtype == torch.float32 else b)
    assert int(outs[0][3][0]) > 0 and bool(torch.signbit(outs[0][2]).all())


@pytest.mark.parametrize('maxcon', [1, 3, 4, 6, 8, 12, 16, 20])
@pytest.mark.parametrize('shift', [0, 2])
def test_contact_list_walk_at_every_length_and_alignment(maxcon, shift):
    """The per-lane walk of a padded contact list row (row_pair_mask: 16-byte steps when maxcon % 4 == 0 and the list is
    16-byte aligned, dword steps otherwise) against the oracle's list -> mask conversion: every ncon from 0 to maxcon
    (lengths that end inside a 16-byte step), entries past ncon filled with pairs that WOULD set bits if they were read,
    list base 16-byte aligned and 4 bytes off -- through bpg_contacts_to_masks, bpg_contacts_to_masks_tb and
    bpg_step_contacts."""
    v = VARIANTS[ENV_ID]
    T, n = 3, 2 * 64 * (maxcon + 1) + 5
    rng = np.random.RandomState(maxcon * 7 + shift)
    num_grey = rng.randint(3, size=n).astype(np.int32)
    ngeom = np.array([len(v['models'][m]['id2obj']) for m in num_grey])
    ncon = ((np.arange(T * n) * 5 + rng.randint(2)) % (maxcon + 1)).astype(np.int32).reshape(T, n)
    ncon[0, :4] = [-3, maxcon + 7, 0, maxcon]                    # clamped to 0 .. maxcon (sanitised input)
    pairs = (rng.randint(0, 1 << 30, size=(T, n, maxcon, 2)) % ngeom[None, :, None, None]).astype(np.int16)
    flat = torch.zeros(pairs.size + 8 + shift, dtype=torch.int16, device=DEV)
    off = (-(flat.data_ptr() // 2) % 8 + shift) % 8 if shift else (-(flat.data_ptr() // 2)) % 8
    dp = flat[off:off + pairs.size].view(T, n, maxcon, 2)
    assert dp.data_ptr() % 16 == (4 if shift else 0)
    dp.copy_(torch.as_tensor(pairs))
    dn = torch.as_tensor(ncon, device=DEV)
    ng = torch.as_tensor(num_grey, device=DEV)
    nc_eff = np.clip(ncon, 0, maxcon)
    exp = np.zeros((T, n), np.uint32)
    for t in range(T):
        for e in range(n):
            exp[t, e] = O.contacts_to_mask(v['models'][int(num_grey[e])]['id2obj'], pairs[t, e, :nc_eff[t, e]])
    env = bpg.make(ENV_ID, num_envs=n, device=DEV, seed=2)
    # whole recorded rollout: the model of env e is the one of the episode about to be drawn (num_grey)
    tb = bpg.time_major_empty(T, n, torch.int32, DEV)
    tb.fill_(-1)
    bpg._lib.check(L.bpg_contacts_to_masks_tb(env._h, P(dp), P(dn), T, maxcon, P(ng), P(tb), int(tb.stride(0)), None))
    np.testing.assert_array_equal(tb.cpu().numpy().view(np.uint32), exp)
    # one step's lists, model from the state
    env.reset(num_grey=ng)
    one = torch.empty(n, dtype=torch.int32, device=DEV)
    bpg._lib.check(L.bpg_contacts_to_masks(env._h, P(dp[1]), P(dn[1]), maxcon, P(one), None))
    np.testing.assert_array_equal(one.cpu().numpy().view(np.uint32), exp[1])
    # the step from lists == the step from the masks just checked
    ref = bpg.make(ENV_ID, num_envs=n, device=DEV, seed=2)
    ref.reset(num_grey=ng)
    r = [torch.empty(n, dtype=torch.float32, device=DEV) for _ in range(2)]
    f = [torch.empty(n, dtype=torch.uint8, device=DEV) for _ in range(2)]
    for t in range(T):
        bpg._lib.check(L.bpg_step_contacts(env._h, P(dp[t]), P(dn[t]), maxcon, P(r[0]), P(f[0]), None))
        bpg._lib.check(L.bpg_step_masks(ref._h, P(tb[t]), P(r[1]), P(f[1]), None))
        assert torch.equal(r[0].view(torch.int32), r[1].view(torch.int32)) and torch.equal(f[0], f[1])
    assert torch.equal(env.state_now, ref.state_now) and torch.equal(env.state_ever, ref.state_ever)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_compute_reward_rows_of_every_length_in_numpys_order(dtype):
    """bpg_compute_reward_f32 / _f64 (rows and one-goal-row forms) against NumPy itself -- np.sum(ag * g, axis=-1) !=
    np.count_nonzero(g), fetch_env.py:141-143, in the arrays' own precision -- on NON-ternary rows whose verdict depends on
    the summation order: every length class of NumPy's pairwise reduction (below 8, the 8-accumulator block up to 128,
    one and two splits up to 256), lengths that are and are not multiples of the 16-byte vector, tiles with ragged tails,
    16-byte aligned and 8-/4-byte offset arrays (the scalar staging path)."""
    rng = np.random.RandomState(17)
    for ln in (1, 2, 5, 7, 8, 9, 16, 25, 36, 37, 64, 100, 127, 128, 129, 130, 160, 200, 255, 256):
        for n in (1, 127, 130, 389):
            g = (rng.randint(1, 4, size=(n, ln)) * (rng.rand(n, ln) < 0.6)).astype(dtype)
            c = np.count_nonzero(g, axis=-1)
            ag = rng.rand(n, ln).astype(dtype)
            # rows whose exact sum is close to the count: the rounding order decides `d != c`
            with np.errstate(divide='ignore', invalid='ignore'):
                scale = np.where(c > 0, c / np.maximum(np.sum(ag.astype(np.float64) * g, axis=-1), 1e-30), 1.0)
            ag = (ag * scale[:, None]).astype(dtype)
            ag[::5] = np.where(g[::5] != 0, 1.0 / np.maximum(g[::5], 1), ag[::5]).astype(dtype)
            want = -(np.sum(ag * g, axis=-1) != c).astype(np.float32)
            got = bpg.compute_reward(ag, g)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg='len %d n %d' % (ln, n))
            g1 = g[0]
            want1 = -(np.sum(ag * g1, axis=-1) != np.count_nonzero(g1)).astype(np.float32)
            np.testing.assert_array_equal(bpg.compute_reward(ag, g1).view(np.uint32), want1.view(np.uint32), err_msg='len %d n %d goal row' % (ln, n))
    # arrays that are element-aligned only: one element past a 16-byte boundary
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    fn = L.bpg_compute_reward_f64 if dtype == np.float64 else L.bpg_compute_reward_f32
    n, ln = 300, 36
    g = (rng.randint(1, 4, size=(n, ln)) * (rng.rand(n, ln) < 0.6)).astype(dtype)
    ag = rng.rand(n, ln).astype(dtype)
    want = -(np.sum(ag * g, axis=-1) != np.count_nonzero(g, axis=-1)).astype(np.float32)
    fa = torch.zeros(n * ln + 1, dtype=tdt, device=DEV)
    fg = torch.zeros(n * ln + 1, dtype=tdt, device=DEV)
    fa[1:].copy_(torch.as_tensor(ag.ravel()))
    fg[1:].copy_(torch.as_tensor(g.ravel()))
    o = torch.empty(n, dtype=torch.float32, device=DEV)
    assert fa[1:].data_ptr() % 16 != 0
    bpg._lib.check(fn(P(fa[1:]), P(fg[1:]), n, ln, P(o), 0, None))
    np.testing.assert_array_equal(bits32(o), want.view(np.uint32))


@pytest.mark.parametrize('T', [1, 2, 7, 16, 31, 32, 33, 40, 49, 51, 63, 64])
def test_return_from_step_ok_words_at_any_horizon(T):
    """bpg_rollout(d_G) at horizons without an instantiation of their own (every T <= 64 except 50 / 25 / 10): the
    register-resident form for 32 or 64 accumulators with the steps t >= T counted as ok and only the first T entries of a
    row stored -- G bit for bit the oracle's float64 accumulation (policy_gradient/rollout.py:255-258) of the rewards the same
    launch wrote, ragged batch (partial wave, partial 16-row tile), G 16-byte aligned and 8 bytes off."""
    B, gamma = 64 * 5 + 37, 1.0 - 1.0 / max(T, 2)
    rng = np.random.RandomState(T)
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=4, max_episode_steps=T)
    ng = torch.as_tensor(rng.randint(3, size=B).astype(np.int32), device=DEV)
    masks = torch.as_tensor(legal_masks(rng, T, B, 6, density=4).astype(np.int32), device=DEV)
    rew = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flat = torch.full((B * T + 2,), 7.0, dtype=torch.float64, device=DEV)
    for off in (0, 1):
        base = flat[off:]
        shift = 0 if (base.data_ptr() % 16 == 0) == (off == 0) else 1         # off 0: aligned, off 1: 8 bytes off
        G = flat[off + shift:off + shift + B * T].view(B, T)
        assert (G.data_ptr() % 16 == 0) == (off == 0)
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(), d_reward_tb=rew.data_ptr(),
                                 d_G=G.data_ptr(), gamma=gamma)
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        torch.cuda.synchronize()
        same_bits64(G.cpu().numpy(), oracle_G(rew.cpu().numpy(), gamma))
        tail = flat[off + shift + B * T:]
        assert tail.numel() == 0 or float(tail.min()) == 7.0                   # nothing written past the last row
        flat.fill_(7.0)
    if T >= 16:
        assert (rew.cpu().numpy() == -1.0).any() and (rew.cpu().numpy() == 0.0).any()


@pytest.mark.parametrize('T', [1, 2, 3, 9, 24, 26, 31, 32, 33, 40, 49, 51, 63, 64])
def test_returns_from_a_reward_stream_at_any_horizon(T):
    """bpg_discounted_return / bpg_discounted_return_scan at horizons without an instantiation of their own, T <= 64: the
    register-resident form with 32 or 64 accumulators in which rows t >= T are neither read nor contribute and only the first
    T entries of a row leave.  Exact form bit for bit the oracle's float64 accumulation -- env rewards (the conditional
    subtraction), arbitrary floats (the general product + sum), batches in which only some waves are plain, a negative gamma
    -- on padded rows and with G 8 bytes off a 16-byte boundary; scan form inside its stated bound.  Nothing is written
    past the last row."""
    rng = np.random.RandomState(100 + T)
    B = 64 * 3 + 21
    gamma = 1.0 - 1.0 / max(T, 2)
    env_r = np.where(rng.rand(T, B) < 0.6, -1.0, -0.0).astype(np.float32)
    gen_r = rng.normal(size=(T, B)).astype(np.float32)
    mixed = env_r.copy()
    mixed[:, 64:128] = gen_r[:, 64:128]                       # the second wave runs the general path, the others the plain one
    mixed[T // 2, -1] = np.float32(0.0)                        # +0.0f is not an env reward either
    for name, r, g in (('env', env_r, gamma), ('general', gen_r, gamma), ('mixed', mixed, gamma), ('negative gamma', env_r, -0.5)):
        want = oracle_G(r, g)
        src = bpg.time_major_empty(T, B, torch.float32, DEV)
        src.copy_(torch.as_tensor(r))
        flat = torch.full((B * T + 3,), 7.0, dtype=torch.float64, device=DEV)
        for off in (0, 1):
            k = off if flat.data_ptr() % 16 == 0 else 1 - off
            G = flat[k:k + B * T].view(B, T)
            assert (G.data_ptr() % 16 == 0) == (off == 0)
            bpg.discounted_return(src, g, out=G)
            torch.cuda.synchronize()
            same_bits64(G.cpu().numpy(), want)
            assert float(flat[k + B * T:].min()) == 7.0 and (k == 0 or float(flat[0]) == 7.0)
            flat.fill_(7.0)
        if g > 0:
            got = bpg.discounted_return_scan(src, g).cpu().numpy()
            pw = g ** np.arange(T)
            bound = np.stack([[4 * T * 2.0 ** -53 * float(np.sum(pw[:T - t] * np.abs(r[t:, b].astype(np.float64))))
                               for t in range(T)] for b in range(B)])
            assert np.all(np.abs(got - want) <= bound), (name, float(np.max(np.abs(got - want) - bound)))
