"""Stateful differential fuzz: one batch of HIP envs and one oracle env per batch slot are
driven through the same seeded random sequence of API calls -- mask steps (legal masks, and RAW 21-bit masks that
may carry pairs of objects the episode does not have: fetch_env.py:154-157 never demotes those), ragged contact-list
steps (dense and CSR), fused multi-step launches without reset (with the policy-gradient return of the launch,
policy_gradient/rollout.py:255-258), full and partial resets
(with Variation's num_grey draw), state save / restore -- and compared after every call:
achieved_goal, reward bits, the is_success latch, TimeLimit's done.  Bit-exact throughout.
The oracle side follows fetch_env.py:148-167 (_step_callback), :135-143 (compute_reward),
:275-281 (_is_success), :247-255 / :646-679 (_reset_sim) and gym's TimeLimit."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from oracle import bpg_oracle as O
from util import VARIANTS, ENV_IDS, legal_masks

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


class Mirror:
    """B oracle envs + the bookkeeping the comparison needs."""

    def __init__(self, env_id, B):
        self.env_id, self.B, self.v = env_id, B, VARIANTS[env_id]
        self.envs = [O.OracleEnv(env_id) for _ in range(B)]
        self.model = np.zeros(B, np.int32)          # Variation: which of the three sims (num_grey)
        self.variation = 'Variation' in env_id

    def nobj(self):
        return 4 + self.model if self.variation else np.full(self.B, self.v['num_objs'])

    def reset(self, num_grey, sel):
        for b in np.nonzero(sel)[0]:
            self.envs[b].reset(num_grey=int(num_grey[b]))
            if self.variation:
                self.model[b] = num_grey[b]

    def ag(self):
        return np.stack([e.achieved_goal() for e in self.envs])

    def step_masks(self, m):
        out = [e.step_mask(int(x)) for e, x in zip(self.envs, m)]
        return self._pack(out)

    def step_contacts(self, pairs, ncon):
        out = [e.step(self.v['models'][int(mi)]['id2obj'], pairs[b, :ncon[b]])
               for b, (e, mi) in enumerate(zip(self.envs, self.model))]
        return self._pack(out)

    @staticmethod
    def _pack(out):
        return (np.stack([x[0] for x in out]),
                np.array([np.float32(x[1]).view(np.uint32) for x in out]),
                np.array([x[2] for x in out]), np.array([x[3] for x in out]))


def check_step(ret, exp):
    obs, r, done, info = ret
    np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), exp[0].astype(np.float32))
    np.testing.assert_array_equal(bits(r), exp[1])
    np.testing.assert_array_equal(done.cpu().numpy(), exp[2])
    np.testing.assert_array_equal(info['is_success'].cpu().numpy(), exp[3])


@pytest.mark.parametrize('env_id', ENV_IDS)
@pytest.mark.parametrize('B', [193, 1])
def test_random_api_sequences_vs_oracle(env_id, B):
    # BPG_FUZZ_SEED shifts the sequence (soak runs: `for s in 1 2 3 ...; do BPG_FUZZ_SEED=$s pytest tests/test_gpu_fuzz.py`)
    rng = np.random.RandomState(1000 * B + len(env_id) + 7919 * int(os.environ.get('BPG_FUZZ_SEED', '0')))
    L = bpg._lib.lib()
    env = bpg.make(env_id, num_envs=B, device=DEV, seed=3)
    mir = Mirror(env_id, B)
    ng = rng.randint(3, size=B).astype(np.int32)
    obs = env.reset(num_grey=ng)
    mir.reset(ng, np.ones(B, bool))
    np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), mir.ag().astype(np.float32))
    saved = None
    ops = rng.choice(['mask', 'list', 'csr', 'fused', 'reset', 'partial', 'save', 'restore', 'multi'], size=140,
                     p=[0.33, 0.13, 0.08, 0.11, 0.04, 0.13, 0.06, 0.06, 0.06])
    for op in ops:
        ngeom = np.array([len(mir.v['models'][m]['id2obj']) for m in mir.model])
        if op == 'mask':
            m = legal_masks(rng, 1, B, mir.nobj(), density=int(rng.randint(1, 4)))[0]
            if rng.rand() < 0.3:                    # raw masks: any of the 21 pair bits
                m = (rng.randint(0, 1 << 21, size=B) & rng.randint(0, 1 << 21, size=B)).astype(np.uint32)
            check_step(env.step(None, contact_masks=m.view(np.int32)), mir.step_masks(m))
        elif op in ('list', 'csr'):
            maxcon = int(rng.choice([1, 4, 11, 12]))
            ncon = rng.randint(0, maxcon + 1, size=B).astype(np.int32)
            pairs = (rng.randint(0, 1 << 30, size=(B, maxcon, 2)) % ngeom[:, None, None]).astype(np.int16)
            exp = mir.step_contacts(pairs, ncon)
            if op == 'list':
                check_step(env.step(None, contacts=(pairs, ncon)), exp)
            else:                                   # the same lists packed back to back, row offsets
                flat = np.concatenate([pairs[b, :ncon[b]] for b in range(B)] + [np.zeros((0, 2), np.int16)])
                if len(flat) == 0:
                    flat = np.zeros((1, 2), np.int16)
                offs = np.concatenate([[0], np.cumsum(ncon)]).astype(np.int64)
                check_step(env.step(None, contacts=(flat, offs)), exp)
        elif op == 'fused':                         # T' steps in one launch, no reset, episode continues
            Tp = int(rng.randint(1, 8))
            masks = legal_masks(rng, Tp, B, mir.nobj(), density=2)
            if rng.rand() < 0.3:
                masks = (rng.randint(0, 1 << 21, size=(Tp, B)) & rng.randint(0, 1 << 21, size=(Tp, B))).astype(np.uint32)
            dm = torch.as_tensor(masks.view(np.int32), device=DEV)
            rew = torch.empty((Tp, B), dtype=torch.float32, device=DEV)
            flg = torch.empty((Tp, B), dtype=torch.uint8, device=DEV)
            G = torch.empty((B, Tp), dtype=torch.float64, device=DEV)
            ret = torch.full((B,), 9.0, dtype=torch.float32, device=DEV)
            with_G = rng.rand() < 0.6               # with d_G the return comes from the step-ok word, without from the counter
            a = bpg._lib.RolloutArgs(T=Tp, d_masks=dm.data_ptr(), d_reward_tb=rew.data_ptr(), d_flags_tb=flg.data_ptr(),
                                     d_G=G.data_ptr() if with_G else None, gamma=0.9, d_episode_return=ret.data_ptr())
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
            if not with_G:
                bpg._lib.check(L.bpg_discounted_return(C.c_void_p(rew.data_ptr()), B, Tp, 0.9, C.c_void_p(G.data_ptr()), 0, None))
            torch.cuda.synchronize()
            rr = rew.cpu().numpy()
            np.testing.assert_array_equal(bits(ret), O.episode_return(rr).view(np.uint32))
            for b in range(0, B, 17):
                np.testing.assert_array_equal(G[b].cpu().numpy().view(np.uint64),
                                              O.discounted_return(rr[:, b].astype(np.float64), 0.9).view(np.uint64))
            for t in range(Tp):
                exp = mir.step_masks(masks[t])
                np.testing.assert_array_equal(bits(rew[t]), exp[1])
                np.testing.assert_array_equal(flg[t].cpu().numpy() & 3, exp[3].astype(np.uint8) | (exp[2].astype(np.uint8) << 1))
            np.testing.assert_array_equal(env.achieved_goal_f64().cpu().numpy(), mir.ag())
        elif op == 'multi':                         # K x (reset + T' steps) in ONE launch (the evaluator's loop)
            K, Tp = int(rng.randint(2, 5)), int(rng.randint(1, 7))
            ngk = rng.randint(3, size=(K, B)).astype(np.int32)
            nob = 4 + ngk if mir.variation else np.broadcast_to(mir.nobj(), (K, B))
            masks = np.concatenate([legal_masks(rng, Tp, B, nob[k], density=2) for k in range(K)])
            if rng.rand() < 0.3:
                masks = (rng.randint(0, 1 << 21, size=(K * Tp, B)) & rng.randint(0, 1 << 21, size=(K * Tp, B))).astype(np.uint32)
            dm = torch.as_tensor(masks.view(np.int32), device=DEV)
            rew = torch.empty((K * Tp, B), dtype=torch.float32, device=DEV)
            flg = torch.empty((K * Tp, B), dtype=torch.uint8, device=DEV)
            cnt = torch.zeros(K, dtype=torch.int64, device=DEV)
            dng = torch.as_tensor(ngk, device=DEV)
            ret = torch.full((K, B), 9.0, dtype=torch.float32, device=DEV)
            a = bpg._lib.RolloutArgs(T=Tp, reset_first=1, rollouts=K, d_masks=dm.data_ptr(), d_reward_tb=rew.data_ptr(),
                                     d_flags_tb=flg.data_ptr(), d_num_grey=dng.data_ptr(), d_success_count=cnt.data_ptr(),
                                     d_episode_return=ret.data_ptr() if rng.rand() < 0.7 else None)
            bpg._lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
            torch.cuda.synchronize()
            if a.d_episode_return:
                rr = rew.cpu().numpy()
                for k in range(K):
                    np.testing.assert_array_equal(bits(ret[k]), O.episode_return(rr[k * Tp:(k + 1) * Tp]).view(np.uint32))
            for k in range(K):
                mir.reset(ngk[k], np.ones(B, bool))
                for t in range(Tp):
                    exp = mir.step_masks(masks[k * Tp + t])
                    np.testing.assert_array_equal(bits(rew[k * Tp + t]), exp[1])
                    np.testing.assert_array_equal(flg[k * Tp + t].cpu().numpy() & 3,
                                                  exp[3].astype(np.uint8) | (exp[2].astype(np.uint8) << 1))
                assert int(cnt[k].item()) == sum(int(e.e.has_succeeded) for e in mir.envs)
            np.testing.assert_array_equal(env.achieved_goal_f64().cpu().numpy(), mir.ag())
            if mir.variation:
                np.testing.assert_array_equal(env.num_objs_per_env().cpu().numpy(), mir.nobj())
                env._fill_static_observation()      # raw C-ABI call under a Python env: its observation rows follow the new draws
        elif op in ('reset', 'partial'):
            sel = np.ones(B, bool) if op == 'reset' else rng.rand(B) < 0.3
            ng = rng.randint(3, size=B).astype(np.int32)
            obs = env.reset(num_grey=ng, env_mask=None if op == 'reset' else sel.astype(np.uint8))
            mir.reset(ng, sel)
            np.testing.assert_array_equal(obs['achieved_goal'].cpu().numpy(), mir.ag().astype(np.float32))
            if mir.variation:
                np.testing.assert_array_equal(env.num_objs_per_env().cpu().numpy(), mir.nobj())
                np.testing.assert_array_equal(obs['observation'][:, 0].cpu().numpy(), (mir.nobj() - 2).astype(np.float32))
        elif op == 'save':
            saved = (env.state_dict(), [C.string_at(C.addressof(e.e), C.sizeof(e.e)) for e in mir.envs], mir.model.copy())
        elif op == 'restore' and saved is not None:
            env.load_state_dict(saved[0])
            for e, raw in zip(mir.envs, saved[1]):
                C.memmove(C.addressof(e.e), raw, len(raw))
            mir.model = saved[2].copy()
            np.testing.assert_array_equal(env.achieved_goal_f64().cpu().numpy(), mir.ag())
    exp_succ = sum(int(e.e.has_succeeded) for e in mir.envs)
    assert int(env.success_count().item()) == exp_succ
