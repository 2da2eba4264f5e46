#!/usr/bin/env python3
"""How much does the headline launch depend on WHERE its streams were allocated?  N independent (masks, reward, flags)
sets in one process on one box, the launch timed on pairs of sets (alternating, so that nothing stays in the Infinity
Cache), several orders of allocation."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib  # noqa: E402

B, T, dev = 1 << 20, 50, torch.device('cuda', 0)
L = _lib.lib()
env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=dev, seed=0)
ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
mode = sys.argv[1] if len(sys.argv) > 1 else 'separate'
NSET = 8
LD = int(L.bpg_row_stride(B))
sets = []
if mode == 'grouped':               # all mask arrays first, then all reward arrays, then all flag arrays (bench.py's row-stride probe)
    ms = [torch.empty((T, LD), dtype=torch.int32, device=dev) for _ in range(NSET)]
    rs = [torch.empty((T, LD), dtype=torch.float32, device=dev) for _ in range(NSET)]
    fs = [torch.empty((T, LD), dtype=torch.uint8, device=dev) for _ in range(NSET)]
for k in range(NSET):
    env.seed(k)
    if mode == 'grouped':
        m, r, f = ms[k], rs[k], fs[k]
    elif mode == 'separate':        # three allocations per set (what bench.py does)
        m = torch.empty((T, LD), dtype=torch.int32, device=dev)
        r = torch.empty((T, LD), dtype=torch.float32, device=dev)
        f = torch.empty((T, LD), dtype=torch.uint8, device=dev)
    else:                           # one arena per set, streams back to back (2 MiB aligned pieces)
        al = 2 << 20
        nb = [T * LD * 4, T * LD * 4, T * LD]
        off = [0]
        for x in nb[:-1]:
            off.append((off[-1] + x + al - 1) // al * al)
        arena = torch.empty(off[-1] + nb[-1] + al, dtype=torch.uint8, device=dev)
        base = (-arena.data_ptr()) % al
        m = arena[base + off[0]: base + off[0] + nb[0]].view(torch.int32).view(T, LD)
        r = arena[base + off[1]: base + off[1] + nb[1]].view(torch.float32).view(T, LD)
        f = arena[base + off[2]: base + off[2] + nb[2]].view(T, LD)
    _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), LD, None))
    sets.append((m, r, f))
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(),
                         d_flags_tb=f.data_ptr(), ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr()) for m, r, f in sets]


def timed(idx, n=60):
    for i in range(20):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[idx[i % len(idx)]]), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[idx[i % len(idx)]]), None))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for i in range(300):
    _lib.check(L.bpg_rollout(env._h, C.byref(args[i % NSET]), None))
torch.cuda.synchronize()
res = {'mode': mode, 'all_sets_round_robin': [round(timed(list(range(NSET))), 2) for _ in range(3)]}
res['pairs'] = {'%d+%d' % (a, a + 1): [round(timed([a, a + 1]), 2) for _ in range(2)] for a in range(0, NSET, 2)}
res['ptrs_MiB'] = [[round(x.data_ptr() / 2**20, 1) for x in st] for st in sets[:3]]
print(json.dumps(res))
