#!/usr/bin/env python3
"""Row stride of the MASK stream x where the mask array was allocated: N independently allocated arenas, every arena read
with several row strides (the reward / flag streams fixed).  Is there a stride that is fast wherever the array landed?"""
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
PADS = [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '0,256,768,1280,2304,4352,8448,16640,33024'.split(','))]
L = _lib.lib()
env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=dev, seed=0)
ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
LDO = int(L.bpg_row_stride(B))
arenas = [torch.empty(T * (B + max(PADS)), dtype=torch.int32, device=dev) for _ in range(N)]
r = torch.empty((T, LDO), dtype=torch.float32, device=dev)
f = torch.empty((T, LDO), dtype=torch.uint8, device=dev)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)


def timed(a, n=30):
    for i in range(8):
        _lib.check(L.bpg_rollout(env._h, C.byref(a), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        _lib.check(L.bpg_rollout(env._h, C.byref(a), None))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tab = []
for k, ar in enumerate(arenas):
    row = []
    for pad in PADS:
        ld = B + pad
        m = ar[:T * ld].view(T, ld)
        env.seed(k)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), ld, None))
        a = _lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(),
                             d_flags_tb=f.data_ptr(), ld_masks=ld, ld_out=LDO, d_success_count=cnt.data_ptr())
        row.append(round(timed(a), 1))
    tab.append(row)
print(json.dumps({'pads_elements': PADS, 'us_per_launch[arena][pad]': tab,
                  'best_pad_per_arena': [PADS[row.index(min(row))] for row in tab],
                  'worst_over_arenas_per_pad': [max(row[i] for row in tab) for i in range(len(PADS))]}))
