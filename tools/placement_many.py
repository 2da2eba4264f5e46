#!/usr/bin/env python3
"""Many mask arrays allocated one after the other (N x 210 MB), the headline launch timed on consecutive pairs of them (reward /
flag streams fixed): is there structure in WHICH allocations read fast?  Prints the device address of every array."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
MODE = sys.argv[2] if len(sys.argv) > 2 else 'torch'          # torch: torch.empty per array; hip: hipMalloc per array (no caching allocator)
L = _lib.lib()
env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=dev, seed=0)
ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
LD = int(L.bpg_row_stride(B))
WHERE = sys.argv[3] if len(sys.argv) > 3 else 'first'          # reward / flag arrays allocated before ('first'), after ('last') or in the
                                                               # middle of ('mid') the mask arrays


def out_arrays():
    return ([torch.empty((T, LD), dtype=torch.float32, device=dev) for _ in range(2)],
            [torch.empty((T, LD), dtype=torch.uint8, device=dev) for _ in range(2)])


BALLAST_MB = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # held allocation made before everything else
ballast = torch.empty(BALLAST_MB << 20, dtype=torch.uint8, device=dev) if BALLAST_MB else None
if WHERE == 'first':
    r, f = out_arrays()
src = torch.empty((T, LD), dtype=torch.int32, device=dev)
_lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(src.data_ptr()), LD, None))
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
hip = C.CDLL('libamdhip64.so')
ptrs = []
keep = []
for k in range(N):
    if MODE == 'hip':
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(T * LD * 4)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(src.data_ptr()), C.c_size_t(T * LD * 4), 3) == 0
        ptrs.append(p.value)
    else:
        m = torch.empty((T, LD), dtype=torch.int32, device=dev)
        m.copy_(src)
        keep.append(m)
        ptrs.append(m.data_ptr())
    if WHERE == 'mid' and k == N // 2:
        r, f = out_arrays()
if WHERE == 'last':
    r, f = out_arrays()
torch.cuda.synchronize()


def timed(pp, n=16):
    args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=p, d_reward_tb=r[k % 2].data_ptr(),
                             d_flags_tb=f[k % 2].data_ptr(), ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr()) for k, p in enumerate(pp)]
    for i in range(4):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[i % len(args)]), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[i % len(args)]), None))
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)


us = [timed([ptrs[k], ptrs[k + 1]]) for k in range(0, N - 1, 2)]
print(json.dumps({'mode': MODE, 'ballast_MB': BALLAST_MB, 'reward_flag_arrays': WHERE, 'reward_GiB': [round(x.data_ptr() / 2**30, 3) for x in r], 'us_per_launch_on_pairs': us, 'addresses_GiB': [round(p / 2**30, 3) for p in ptrs]}))
again = [timed([ptrs[k], ptrs[k + 1]]) for k in range(0, N - 1, 2)]
print(json.dumps({'again': again}))
