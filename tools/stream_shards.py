#!/usr/bin/env python3
"""1 M envs as ONE handle on one stream against S shards of 1 M / S envs, each on its own stream (launches of different
shards overlap: while one shard's launch ramps up or drains, the others stream).  us per 1 M-env rollout, masks cold."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib, distributed as D  # noqa: E402

dev, T, L = torch.device('cuda', 0), 50, _lib.lib()
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
SPLITS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4]
# 'own' (default): every shard has (T, n_shard) arrays of its own; 'shared': the shards' arrays are column ranges of ONE set
# of (T, total) arrays (what a caller with one batch-wide array would hand out)
LAYOUTS = sys.argv[3].split(',') if len(sys.argv) > 3 else ['own']
NB = max(4, int(2.5 * 256 * 2**20 // (TOTAL * T * 9)) + 1)


def setup(S, layout='own'):
    shards = []
    whole = None
    if layout == 'shared':
        whole = [(bpg.time_major_empty(T, TOTAL, torch.int32, dev), bpg.time_major_empty(T, TOTAL, torch.float32, dev),
                  bpg.time_major_empty(T, TOTAL, torch.uint8, dev)) for _ in range(NB)]
    for r in range(S):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            env = D.make_sharded('BlocksTouchVariation-v0', TOTAL, r, S, device=dev, seed=0)
            B = env.num_envs
            ng = ((torch.arange(B, device=dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
            sets = []
            for k in range(NB):
                env.seed(k)
                if whole is None:
                    m = bpg.time_major_empty(T, B, torch.int32, dev)
                    rw, fl = bpg.time_major_empty(T, B, torch.float32, dev), bpg.time_major_empty(T, B, torch.uint8, dev)
                else:
                    lo = env.env_id_base
                    m, rw, fl = (x[:, lo:lo + B] for x in whole[k])
                _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), int(m.stride(0)), C.c_void_p(st.cuda_stream)))
                sets.append((m, rw, fl))
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=rr.data_ptr(),
                                     d_flags_tb=f.data_ptr(), ld_masks=int(m.stride(0)), ld_out=int(m.stride(0)),
                                     d_success_count=cnt.data_ptr()) for m, rr, f in sets]
        shards.append((env, C.c_void_p(st.cuda_stream), args, (ng, sets, cnt, st)))
    torch.cuda.synchronize()
    return shards


def run(shards, n):
    for i in range(n):
        for env, st, args, _ in shards:
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i % NB]), st))


for S, layout in [(S, lay) for _ in range(2) for S in SPLITS for lay in LAYOUTS]:
    shards = setup(S, layout)
    run(shards, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    run(shards, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-6s %2d shard(s) x %7d envs on %2d stream(s): %7.2f us per %d-env rollout = %.3f of 8 TB/s (wall, %d rollouts)' % (layout, S, TOTAL // S, S, dt / n * 1e6, TOTAL, TOTAL * (T * 9 + 12) / (dt / n) / 8e12, n))
    del shards
    torch.cuda.empty_cache()
