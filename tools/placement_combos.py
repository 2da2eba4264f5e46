#!/usr/bin/env python3
"""The headline launch on every combination of N mask, N reward and N flag arrays allocated independently in one process:
which of the three streams' relative placement the launch duration depends on."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = _lib.lib()
env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=dev, seed=0)
ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
LD = int(L.bpg_row_stride(B))
ms = [torch.empty((T, LD), dtype=torch.int32, device=dev) for _ in range(N)]
rs = [torch.empty((T, LD), dtype=torch.float32, device=dev) for _ in range(N)]
fs = [torch.empty((T, LD), dtype=torch.uint8, device=dev) for _ in range(N)]
for k, m in enumerate(ms):
    env.seed(k)
    _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), LD, None))
cnt = torch.zeros(2, dtype=torch.int64, device=dev)


def args(i, j, k):
    return _lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=ms[i].data_ptr(), d_reward_tb=rs[j].data_ptr(),
                            d_flags_tb=fs[k].data_ptr(), ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr())


def timed(alist, n=40):
    for i in range(10):
        _lib.check(L.bpg_rollout(env._h, C.byref(alist[i % len(alist)]), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        _lib.check(L.bpg_rollout(env._h, C.byref(alist[i % len(alist)]), None))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for i in range(300):
    _lib.check(L.bpg_rollout(env._h, C.byref(args(i % N, i % N, i % N)), None))
torch.cuda.synchronize()
# does re-launching on ONE set find its data in the Infinity Cache?  one set repeated against two sets alternating
one = timed([args(0, 0, 0)])
two = timed([args(0, 0, 0), args(1, 1, 1)])
res = {'one_set_repeated': round(one, 2), 'two_sets_alternating': round(two, 2)}
tab = {}
for i in range(N):
    for j in range(N):
        for k in range(N):
            tab[(i, j, k)] = timed([args(i, j, k)], 30)
v = sorted(tab.values())
res['combos'] = len(v)
res['min_p10_median_p90_max'] = [round(v[0], 2), round(v[len(v) // 10], 2), round(v[len(v) // 2], 2), round(v[len(v) * 9 // 10], 2), round(v[-1], 2)]
# how much of the variance each index explains: mean over the other two
for name, ax in (('by_mask_array', 0), ('by_reward_array', 1), ('by_flag_array', 2)):
    res[name] = [round(sum(t for key, t in tab.items() if key[ax] == a) / (N * N), 2) for a in range(N)]
res['by_mask_reward_pair'] = [[round(sum(tab[(i, j, k)] for k in range(N)) / N, 1) for j in range(N)] for i in range(N)]
res['by_reward_flag_pair'] = [[round(sum(tab[(i, j, k)] for i in range(N)) / N, 1) for k in range(N)] for j in range(N)]
res['ptrs_MiB'] = {'masks': [round(x.data_ptr() / 2**20) for x in ms], 'reward': [round(x.data_ptr() / 2**20) for x in rs],
                   'flags': [round(x.data_ptr() / 2**20) for x in fs]}
print(json.dumps(res))
