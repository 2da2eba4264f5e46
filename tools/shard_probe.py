#!/usr/bin/env python3
"""The batch as ONE launch against S views of it (bpg_create_view) on S streams -- on column ranges of the same arrays
(`shared`) and on arrays of their own (`own`), launched back to back and forked / joined once -- against K rollouts per launch
and against RolloutStudent.evaluate_rollouts_fused with and without shards.  us per rollout of the whole batch, masks cold.

    python tools/shard_probe.py            (GPU box; log: profiles/r05_shard_probe.log)
"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import blockpuzzle_gym_amd as bpg
from blockpuzzle_gym_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
T = 50
ENV = 'BlocksTouchVariation-v0'
torch.cuda.set_stream(torch.cuda.Stream(dev))
cur = torch.cuda.current_stream(dev)
cst = C.c_void_p(cur.cuda_stream)

def ev():
    return torch.cuda.Event(enable_timing=True)

def frac(B, ms):
    return B * (T * 9 + 12) / (ms * 1e-3) / 8e12

for B in (1 << 20,):
    NBUF = max(4, int(2.5 * 256 * 2**20 // (B * T * 9)) + 1)
    env = bpg.make(ENV, num_envs=B, device=dev, seed=0)
    LD = int(L.bpg_row_stride(B))
    ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
    def tm(dt, ld=LD, n=B, rows=T): return torch.empty((rows, ld), dtype=dt, device=dev)[:, :n]
    sets = []
    for k in range(NBUF):
        env.seed(k)
        m = tm(torch.int32)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), LD, cst))
        sets.append((m, tm(torch.float32), tm(torch.uint8)))
    env.seed(0)
    cnt = torch.zeros(64, dtype=torch.int64, device=dev)
    one = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(), d_flags_tb=f.data_ptr(),
                            ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr()) for m, r, f in sets]
    def run_one(n):
        for i in range(n):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(one[i % NBUF]), cst))
    def timed(fn, n, reps=3):
        fn(n); torch.cuda.synchronize()
        best = []
        for _ in range(reps):
            e0, e1 = ev(), ev()
            e0.record(cur); fn(n); e1.record(cur); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / n)
        return min(best), best
    ms, all_ = timed(run_one, 40)
    print('B %d one handle, 1 rollout/launch: %.2f us  frac %.3f  %s' % (B, ms * 1e3, frac(B, ms), ['%.1f' % (x * 1e3) for x in all_]), flush=True)
    for S in (2, 4, 8):
        vs = bpg.rollout._ViewShards(env, S)
        for layout in ('shared', 'own'):
            vargs = []
            keep = []
            for hv, st, lo, n in vs.views:
                if layout == 'shared':
                    vargs.append([_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr() + 4 * lo, d_masks=m.data_ptr() + 4 * lo,
                                                   d_reward_tb=r.data_ptr() + 4 * lo, d_flags_tb=f.data_ptr() + lo, ld_masks=LD, ld_out=LD,
                                                   d_success_count=cnt.data_ptr()) for m, r, f in sets])
                else:
                    ldv = int(L.bpg_row_stride(n))
                    mine = []
                    for m, r, f in sets:
                        mm = tm(torch.int32, ldv, n); mm.copy_(m[:, lo:lo + n])
                        mine.append((mm, tm(torch.float32, ldv, n), tm(torch.uint8, ldv, n)))
                    keep.append(mine)
                    vargs.append([_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr() + 4 * lo, d_masks=mm.data_ptr(),
                                                   d_reward_tb=rr.data_ptr(), d_flags_tb=ff.data_ptr(), ld_masks=ldv, ld_out=ldv,
                                                   d_success_count=cnt.data_ptr()) for mm, rr, ff in mine])
            def run_views(nn):
                vs.fork()
                for i in range(nn):
                    for (hv, st, lo, n), va in zip(vs.views, vargs):
                        _lib.check(L.bpg_rollout_v2(hv, C.byref(va[i % NBUF]), C.c_void_p(st.cuda_stream)))
                vs.join()
            ms, all_ = timed(run_views, 40)
            print('B %d %d views %-6s arrays, back to back: %.2f us  frac %.3f  %s' % (B, S, layout, ms * 1e3, frac(B, ms), ['%.1f' % (x * 1e3) for x in all_]), flush=True)
            del keep, vargs
        vs.close()
    # MULTI direct: K rollouts per launch on one handle
    for K in (5, 20):
        masks = tm(torch.int32, rows=K * T); rw = tm(torch.float32, rows=K * T); fl = tm(torch.uint8, rows=K * T)
        for k in range(K):
            masks[k * T:(k + 1) * T].copy_(sets[k % NBUF][0])
        ngk = ng[None].expand(K, B).contiguous()
        a = _lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_num_grey=ngk.data_ptr(), d_masks=masks.data_ptr(), d_reward_tb=rw.data_ptr(),
                             d_flags_tb=fl.data_ptr(), ld_masks=LD, ld_out=LD, d_success_count=cnt.data_ptr())
        def run_multi(nn):
            for i in range(nn):
                _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), cst))
        ms, all_ = timed(run_multi, 3)
        print('B %d one handle, %d rollouts/launch: %.2f us per rollout  frac %.3f' % (B, K, ms / K * 1e3, frac(B, ms / K)), flush=True)
        dims = dict(o=env.dimo, u=4, g=env.dimg, info_is_success=1)
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV, num_envs=num_envs, device=dev, seed=0), None, dims, None, T, rollout_batch_size=B, output='torch')
        for sh in (1, 'auto'):
            for outs in (True, False):
                kw = dict(reward_tb=rw, flags_tb=fl) if outs else {}
                def run_w(nn):
                    for i in range(nn):
                        w.evaluate_rollouts_fused(masks, num_grey=ngk, shards=sh, **kw)
                ms, all_ = timed(run_w, 3)
                nb = B * (T * (9 if outs else 4) + 12)
                print('B %d worker.evaluate K=%d shards=%s outputs=%s: %.2f us per rollout  frac(of its bytes) %.3f' % (B, K, sh, outs, ms / K * 1e3, nb / (ms / K * 1e-3) / 8e12), flush=True)
        del w, masks, rw, fl
        torch.cuda.empty_cache()
