"""Maximum sizes: batches whose (T, B) streams and episode arrays hold more than 2^31 elements
(64-bit indexing in every kernel of the path), checked through shard invariance -- env b of a
batch keyed by global env ids [0, B) must equal env 0 of a small batch created with
env_id_base = b (the multi-GPU partition relies on exactly this) -- and the
small batches are the ones the oracle tests pin.  Sized for one MI355X (288 GB): <= 60 GB."""
import ctypes as C

import pytest
import torch

import blockpuzzle_gym_amd as bpg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ENV_ID = 'BlocksTouchVariation-v0'
T = 50


def P(t):
    return C.c_void_p(t.data_ptr())


def packed_rollout(env, masks=None, ret=None):
    """reset + T steps in one launch; returns (masks, reward, flags, count).  ret: optional (B,) float32 that receives
    the per-env episode return of the same launch."""
    L, B = bpg._lib.lib(), env.num_envs
    if masks is None:
        masks = torch.empty((T, B), dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_synth_masks(env._h, 0, T, P(masks), None))
    reward = torch.empty((T, B), dtype=torch.float32, device=DEV)
    flags = torch.empty((T, B), dtype=torch.uint8, device=DEV)
    cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
    a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_masks=masks.data_ptr(), d_reward_tb=reward.data_ptr(),
                             d_flags_tb=flags.data_ptr(), d_success_count=cnt.data_ptr(),
                             d_episode_return=None if ret is None else ret.data_ptr())
    bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    return masks, reward, flags, cnt


def offsets(B, K):
    return [0, (B // 2) | 1, B - K]               # first rows, an odd offset in the middle, the last rows


def test_packed_streams_beyond_2_31_elements():
    B, K = 45_000_002, 4096                       # T * B = 2.25e9 elements per stream
    L = bpg._lib.lib()
    big = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=5)
    # the generator's num_grey draw is taken before anything else consumes a reset index
    ret = torch.empty(B, dtype=torch.float32, device=DEV)
    m, r, f, cnt = packed_rollout(big, ret=ret)
    assert int(cnt[0]) == int((f[T - 1] & 1).sum()) == int(big.success_count()) > 0
    # the per-env episode return of the same launch == -(failed steps) of the reward stream it wrote, all 45 M envs
    fails = torch.zeros(B, dtype=torch.int32, device=DEV)
    for t in range(T):
        fails += (r[t] != 0)
    assert torch.equal(ret, -(fails.float())) and bool(torch.signbit(ret).all())
    del fails
    # a second consumer of the same streams: the return of the policy-gradient worker, (B, T) float64
    G = torch.empty((B, T), dtype=torch.float64, device=DEV)
    bpg._lib.check(L.bpg_discounted_return(P(r), B, T, 0.98, P(G), 0, None))
    # slices of the big batch == small batches keyed by the same global env ids
    for off in offsets(B, K):
        small = bpg.make(ENV_ID, num_envs=K, device=DEV, seed=5, env_id_base=off)
        rets = torch.empty(K, dtype=torch.float32, device=DEV)
        ms, rs, fs, cs = packed_rollout(small, ret=rets)
        sl = slice(off, off + K)
        assert torch.equal(ms, m[:, sl]) and torch.equal(rets.view(torch.int32), ret[sl].view(torch.int32))
        assert torch.equal(rs.view(torch.int32), r[:, sl].view(torch.int32))
        assert torch.equal(fs, f[:, sl])
        assert torch.equal(small.state_now, big.state_now[sl]) and torch.equal(small.state_ever, big.state_ever[sl])
        Gs = torch.empty((K, T), dtype=torch.float64, device=DEV)
        bpg._lib.check(L.bpg_discounted_return(P(rs), K, T, 0.98, P(Gs), 0, None))
        assert torch.equal(Gs.view(torch.int64), G[sl].view(torch.int64))
    # G[:, T-1] is the last reward itself
    assert torch.equal(G[:, T - 1].float().view(torch.int32), r[T - 1].view(torch.int32))
    del G
    # per-step kernel on the big batch: continue the episode one step with row 0's masks
    r1 = torch.empty(B, dtype=torch.float32, device=DEV)
    f1 = torch.empty(B, dtype=torch.uint8, device=DEV)
    bpg._lib.check(L.bpg_step_masks(big._h, P(m[0]), P(r1), P(f1), None))
    assert bool(((f1 & 2) != 0).all())            # step 51 of a 50-step TimeLimit: done everywhere
    off = B - K
    small = bpg.make(ENV_ID, num_envs=K, device=DEV, seed=5, env_id_base=off)
    ms, _, _, _ = packed_rollout(small)
    rs1 = torch.empty(K, dtype=torch.float32, device=DEV)
    fs1 = torch.empty(K, dtype=torch.uint8, device=DEV)
    bpg._lib.check(L.bpg_step_masks(small._h, P(ms[0]), P(rs1), P(fs1), None))
    assert torch.equal(rs1.view(torch.int32), r1[off:].view(torch.int32)) and torch.equal(fs1, f1[off:])


def test_episode_arrays_beyond_2_31_elements():
    B, K = 2_200_006, 2048                        # ag: B * 51 * 36 = 4.0e9 elements
    L = bpg._lib.lib()

    def episode(env):
        n = env.num_envs
        masks = torch.empty((T, n), dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_synth_masks(env._h, 0, T, P(masks), None))
        ag = torch.empty((n, T + 1, 36), dtype=torch.float32, device=DEV)
        g = torch.empty((n, T, 36), dtype=torch.float32, device=DEV)
        sc = torch.empty((n, T, 1), dtype=torch.float32, device=DEV)
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_masks=masks.data_ptr(), d_ep_ag=ag.data_ptr(),
                                 d_ep_g=g.data_ptr(), d_ep_success=sc.data_ptr())
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        return ag, g, sc

    big = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=8)
    ag, g, sc = episode(big)
    for off in offsets(B, K):
        small = bpg.make(ENV_ID, num_envs=K, device=DEV, seed=8, env_id_base=off)
        ags, gs, scs = episode(small)
        sl = slice(off, off + K)
        assert torch.equal(ags, ag[sl]) and torch.equal(gs, g[sl]) and torch.equal(scs, sc[sl])
        assert torch.equal(small.state_ever, big.state_ever[sl])
    # the worker's step kernel writing a time slot of the big episode buffer (row stride 51 * 36)
    big2 = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=8)
    masks = torch.empty((1, B), dtype=torch.int32, device=DEV)          # as episode(): drawn before the reset
    bpg._lib.check(L.bpg_synth_masks(big2._h, 0, 1, P(masks), None))
    big2.reset()
    slot = torch.empty_like(ag)
    big2.step(None, contact_masks=masks[0], ag_out=slot[:, 1])
    assert torch.equal(slot[:, 1], ag[:, 1])


def test_return_from_step_ok_words_at_full_size_and_beyond_2_31_elements():
    """bpg_rollout(d_G) at BASELINE's 1 048 576 envs and at 45 M envs (G holds 2.25e9 float64): the rows of the
    rollout call == the rows bpg_discounted_return computes from the reward stream the same call wrote (that
    kernel is pinned to the oracle at small sizes), bit for bit over the WHOLE batch; plus the size-independent
    recurrence |G[t] - (r[t] + gamma G[t+1])| <= 1e-12 and G[:, T-1] == r[T-1] everywhere."""
    L = bpg._lib.lib()
    gamma = 1.0 - 1.0 / T
    for B in (1 << 20, 45_000_002):
        env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=8)
        masks = torch.empty((T, B), dtype=torch.int32, device=DEV)
        bpg._lib.check(L.bpg_synth_masks(env._h, 0, T, P(masks), None))
        reward = torch.empty((T, B), dtype=torch.float32, device=DEV)
        G = torch.empty((B, T), dtype=torch.float64, device=DEV)
        ng = (torch.arange(B, device=DEV, dtype=torch.int32) * 5) % 3
        a = bpg._lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=masks.data_ptr(),
                                 d_reward_tb=reward.data_ptr(), d_G=G.data_ptr(), gamma=gamma)
        bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        # the last column is the last reward; the recurrence holds within rounding, chunk by chunk (memory)
        step = 1 << 22
        for lo in range(0, B, step):
            g = G[lo:lo + step]
            r = reward[:, lo:lo + step].t().double()
            assert torch.equal(g[:, T - 1].view(torch.int64), r[:, T - 1].contiguous().view(torch.int64))
            assert float((g[:, :-1] - (r[:, :-1] + gamma * g[:, 1:])).abs().max()) <= 1e-12
            del g, r
        G2 = torch.empty((B, T), dtype=torch.float64, device=DEV)
        bpg._lib.check(L.bpg_discounted_return(P(reward), B, T, gamma, P(G2), 0, None))
        for lo in range(0, B, step):
            assert torch.equal(G[lo:lo + step].view(torch.int64), G2[lo:lo + step].view(torch.int64))
        del G, G2, masks, reward, env
        torch.cuda.empty_cache()
