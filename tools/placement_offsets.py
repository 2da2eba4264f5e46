#!/usr/bin/env python3
"""Where inside ONE big allocation the mask array starts: the headline launch (1 M envs, T = 50) with its mask stream at byte
offsets of an 12 GiB arena -- small offsets (inside a 2 MiB page: the bits a virtual address shares with the physical one), 2 MiB
steps, 64 MiB and 1 GiB steps -- reward / flag streams fixed.  Prints the arrays' device addresses, so that runs can be compared."""
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
LD = int(L.bpg_row_stride(B))
NB = T * LD * 4                                   # bytes of one mask array
GiB, MiB, KiB = 1 << 30, 1 << 20, 1 << 10
arena = torch.empty(12 * GiB, dtype=torch.uint8, device=dev)
r = [torch.empty((T, LD), dtype=torch.float32, device=dev) for _ in range(2)]
f = [torch.empty((T, LD), dtype=torch.uint8, device=dev) for _ in range(2)]
src = torch.empty((T, LD), dtype=torch.int32, device=dev)
_lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(src.data_ptr()), LD, None))
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
print(json.dumps({'arena': hex(arena.data_ptr()), 'reward': [hex(x.data_ptr()) for x in r], 'flags': [hex(x.data_ptr()) for x in f],
                  'mask_array_bytes': NB}), flush=True)


def view(off):
    return arena[off:off + NB].view(torch.int32).view(T, LD)


def timed(offs, n=24):
    """the launch on mask arrays at `offs` (round-robin, so that none is read from the Infinity Cache), us per launch"""
    args = []
    for k, off in enumerate(offs):
        m = view(off)
        m.copy_(src)
        args.append(_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=r[k % 2].data_ptr(),
                                     d_flags_tb=f[k % 2].data_ptr(), ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr()))
    for i in range(6):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[i % len(args)]), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        _lib.check(L.bpg_rollout(env._h, C.byref(args[i % len(args)]), None))
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)


# (1) which GiB of the arena: three arrays 256 MiB apart inside each GiB-aligned region (same low bits)
print('per GiB region (3 arrays 256 MiB apart):', [timed([g * GiB + k * 256 * MiB for k in range(3)]) for g in range(0, 11)], flush=True)
# (2) offsets inside a 2 MiB page: arrays at 0 / 4 / 8 GiB + small offset
small = [0, 256, 1 * KiB, 4 * KiB, 16 * KiB, 64 * KiB, 128 * KiB, 256 * KiB, 512 * KiB, 1 * MiB, 1 * MiB + 512 * KiB]
print('offset inside a 2 MiB page', small, ':', [timed([g * GiB + o for g in (0, 4, 8)]) for o in small], flush=True)
# (3) 2 MiB steps, (4) 64 MiB steps
print('2 MiB steps 0..15:', [timed([g * GiB + k * 2 * MiB for g in (0, 4, 8)]) for k in range(16)], flush=True)
print('64 MiB steps 0..15:', [timed([g * GiB + k * 64 * MiB for g in (0, 4, 8)]) for k in range(16)], flush=True)
# (5) single arrays one by one every 512 MiB (each alone but alternating with a far one at 11 GiB so the cache is flushed)
far = 11 * GiB
print('single array at k x 512 MiB, alternating with one at 11 GiB:', [timed([k * 512 * MiB, far]) for k in range(0, 21)], flush=True)
