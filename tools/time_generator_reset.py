#!/usr/bin/env python3
"""us per fused rollout launch with explicit num_grey draws against generator-drawn ones (the launch that draws advances the
generator's stream index itself since round 4; before, a one-lane launch behind it did)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib  # noqa: E402

dev, T, L = torch.device('cuda', 0), 50, _lib.lib()
for B in (65536, 1 << 20):
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=dev, seed=0)
    ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
    nset = int(2.5 * 256 * 2**20 // (B * T * 9)) + 1
    sets = []
    for k in range(nset):
        env.seed(k)
        m = bpg.time_major_empty(T, B, torch.int32, dev)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), int(m.stride(0)), None))
        sets.append((m, bpg.time_major_empty(T, B, torch.float32, dev), bpg.time_major_empty(T, B, torch.uint8, dev)))
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    for name, g in (('explicit num_grey', ng.data_ptr()), ('generator-drawn', None), ('explicit num_grey', ng.data_ptr()), ('generator-drawn', None)):
        args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=g, d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(), d_flags_tb=f.data_ptr(),
                                 ld_masks=int(m.stride(0)), ld_out=int(m.stride(0)), d_success_count=cnt.data_ptr()) for m, r, f in sets]
        for i in range(2 * nset):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i % nset]), None))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(60, 3 * nset)
        e0.record()
        for i in range(n):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i % nset]), None))
        e1.record()
        torch.cuda.synchronize()
        print('%8d envs  %-18s %7.2f us per launch' % (B, name, e0.elapsed_time(e1) * 1e3 / n))
