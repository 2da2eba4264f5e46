"""Shared helpers for the test-suite (oracle side only)."""
import json
import os

import numpy as np

from oracle import bpg_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
VARIANTS = json.load(open(os.path.join(GOLDEN, 'variants.json')))
ENV_IDS = list(VARIANTS)


def short(env_id):
    return env_id.split('-')[0]


def valid_by_n():
    L = O.lib()
    return np.array([sum(1 << L.orc_pair_index(i, j) for i in range(n) for j in range(i, n))
                     for n in range(7)], np.uint32)


def legal_masks(rng, T, B, num_objs, density=3):
    """Random contact masks (T, B) that only involve objects present in each env."""
    m = rng.randint(0, 1 << 21, size=(T, B)).astype(np.uint32)
    for _ in range(density - 1):
        m &= rng.randint(0, 1 << 21, size=(T, B)).astype(np.uint32)
    num_objs = np.broadcast_to(np.asarray(num_objs), (B,))
    return m & valid_by_n()[num_objs][None, :]


def fixture_masks(env_id, contacts, ncon, model_idx):
    """(B, T, maxcon, 2) geom-id contacts -> (T, B) pair masks via the oracle."""
    v = VARIANTS[env_id]
    B, T = ncon.shape
    out = np.zeros((T, B), np.uint32)
    for b in range(B):
        id2obj = v['models'][int(model_idx[b])]['id2obj']
        for t in range(T):
            out[t, b] = O.contacts_to_mask(id2obj, contacts[b, t, :ncon[b, t]])
    return out


def pack_oracle_env(e):
    """The packed (now, ever) words of include/bpgym.h for the state of an O.OracleEnv: pairs whose matrix entry is 1 /
    is not -1, the elapsed-step counter (saturating at 1023), num_objs of the episode, the has_succeeded latch."""
    L = O.lib()
    env = e.e
    side = env.side
    now = ever = 0
    for i in range(side):
        for j in range(i, side):
            v = env.achieved_goal[i * side + j]
            assert v == env.achieved_goal[j * side + i]
            bit = 1 << L.orc_pair_index(i, j)
            if v == 1.0:
                now |= bit
            if v != -1.0:
                ever |= bit
    now |= min(int(env.elapsed_steps), 1023) << 21
    ever |= (int(env.num_objs) & 7) << 21
    if env.has_succeeded:
        ever |= 1 << 31
    return now, ever
