#!/usr/bin/env python3
"""Runs a few launches of selected kernels of the path at the bench configuration, for
rocprofv3 (kernel trace or one --pmc counter set per invocation):

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_kernels.py return rollout synth
    rocprofv3 --pmc FETCH_SIZE -d out -- python3 tools/prof_kernels.py return

kernels: rollout (fused, masks, 1 M envs), rollout64k / rollout128k / rollout256k / rollout512k (the other launch shapes),
rolloutret / rolloutret512k (the same with the per-env episode return, bpg_rollout_args.d_episode_return),
rollout4m (4 M envs, rows a power of two apart), synth (fused, in-kernel contacts), pg (rollout + the policy-gradient
return from the step-ok words, one call), step (13 batches round-robin), record, return, scan, episode, reward,
assemble (+ set_action, placement), nan, reset
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib  # noqa: E402

ENV_ID, T, DEV = 'BlocksTouchVariation-v0', 50, torch.device('cuda', 0)
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
TM = lambda B, dtype: bpg.time_major_empty(T, B, dtype, DEV)   # noqa: E731  (T, B) stream, rows bpg_row_stride(B) apart


def rollout(B, reps, synth=False, contiguous=False, pg=False, ret=False):
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
    ng = (torch.arange(B, device=DEV, dtype=torch.int32) * 7) % 3
    env.reset(num_grey=ng)
    nset = min(int(2.5 * 256 * 2**20 // (B * T * 9)) + 1, 24)
    tm = (lambda dtype: torch.empty((T, B), dtype=dtype, device=DEV)) if contiguous else (lambda dtype: TM(B, dtype))
    sets = []
    for k in range(max(nset, 3 if B > (1 << 21) else 4)):
        env.seed(k)
        m = tm(torch.int32)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), m.stride(0), None))
        sets.append((m, None if pg else tm(torch.float32), tm(torch.uint8),
                     torch.empty((B, T), dtype=torch.float64, device=DEV) if pg else None))
    cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
    er = torch.empty(B, dtype=torch.float32, device=DEV) if ret else None      # bpg_rollout_args.d_episode_return
    for i in range(reps):
        m, r, f, G = sets[i % len(sets)]
        a = _lib.RolloutArgs(T=T, reset_first=1, d_num_grey=None if synth else ng.data_ptr(), ld_masks=m.stride(0),
                             ld_out=f.stride(0),
                             d_masks=None if synth else m.data_ptr(), d_reward_tb=None if r is None else r.data_ptr(),
                             d_flags_tb=f.data_ptr(), d_success_count=None if synth else cnt.data_ptr(),
                             d_G=None if G is None else G.data_ptr(), gamma=0.98,
                             d_episode_return=None if er is None else er.data_ptr())
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    torch.cuda.synchronize()


def multi(B, K, reps):
    """K consecutive rollouts per launch (bpg_rollout_args.rollouts), explicit draws, streams of K * T rows."""
    env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
    ng = ((torch.arange(K * B, device=DEV, dtype=torch.int32) * 7) % 3).reshape(K, B)
    env.reset(num_grey=ng[0])
    nset = max(2, min(int(2.5 * 256 * 2**20 // (K * B * T * 9)) + 1, 8))
    sets = []
    for k in range(nset):
        env.seed(k)
        m = bpg.time_major_empty(K * T, B, torch.int32, DEV)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, K * T, P(m), m.stride(0), None))
        sets.append((m, bpg.time_major_empty(K * T, B, torch.float32, DEV), bpg.time_major_empty(K * T, B, torch.uint8, DEV)))
    cnt = torch.zeros(K, dtype=torch.int64, device=DEV)
    for i in range(reps):
        m, r, f = sets[i % nset]
        a = _lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_num_grey=ng.data_ptr(), ld_masks=m.stride(0), ld_out=f.stride(0),
                             d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(), d_flags_tb=f.data_ptr(), d_success_count=cnt.data_ptr())
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
    torch.cuda.synchronize()


def trim(B, reps):
    o = torch.randn((3, B, 87), device=DEV)
    o[:, :, 11 + 15:11 + 19] = torch.tensor([0., 0., 1., 0.], device=DEV)              # slot 0 green
    o[:, :, 11 + 19 + 15:11 + 19 + 19] = torch.tensor([0., 0., 0., 1.], device=DEV)    # slot 1 blue
    for j in (2, 3):
        o[:, :, 11 + 19 * j + 15:11 + 19 * j + 19] = torch.tensor([1., 0., 0., 0.], device=DEV)
    g = torch.randint(-1, 2, (3, B, 36), device=DEV).to(torch.float32)
    ag = torch.randint(-1, 2, (3, B, 36), device=DEV).to(torch.float32)
    out = [(torch.empty((B, 40), dtype=torch.float64, device=DEV), torch.empty((B, 16), dtype=torch.float32, device=DEV),
            torch.empty((B, 16), dtype=torch.float32, device=DEV)) for _ in range(3)]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for i in range(reps):
        k = i % 3
        _lib.check(L.bpg_trim_expert_inputs(P(o[k]), P(g[k]), P(ag[k]), B, 87, 36, 40, 16, 4, 1, P(out[k][0]), P(out[k][1]),
                                            P(out[k][2]), P(flag), 0, None))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0


def main():
    which = sys.argv[1:] or ['rollout']
    B = 1 << 20
    for w in which:
        if w == 'rollout':
            rollout(B, 20)
        elif w == 'rollout64k':
            rollout(65536, 60)
        elif w == 'rollout128k':
            rollout(131072, 40)
        elif w == 'rollout256k':
            rollout(262144, 30)
        elif w == 'rollout512k':
            rollout(524288, 20)
        elif w == 'rollout4m':
            rollout(4194304, 8, contiguous=True)
        elif w == 'rolloutret':
            rollout(B, 20, ret=True)
        elif w == 'rolloutret512k':
            rollout(524288, 20, ret=True)
        elif w == 'multi64k':
            multi(65536, 10, 30)
        elif w == 'multi1m':
            multi(B, 10, 6)
        elif w == 'trim':
            trim(B, 15)
        elif w == 'pg':
            rollout(B, 20, pg=True)
        elif w == 'synth':
            rollout(B, 20, synth=True)
        elif w == 'step':
            sets = []
            for k in range(13):
                e = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=k)
                e.reset()
                m = torch.empty(B, dtype=torch.int32, device=DEV)
                _lib.check(L.bpg_synth_masks(e._h, 0, 1, P(m), None))
                sets.append((e, m, torch.empty(B, dtype=torch.float32, device=DEV), torch.empty(B, dtype=torch.uint8, device=DEV)))
            for i in range(13 * 10):
                e, m, r, f = sets[i % 13]
                _lib.check(L.bpg_step_masks(e._h, P(m), P(r), P(f), None))
            torch.cuda.synchronize()
        elif w in ('record', 'return', 'scan', 'episode', 'reward'):
            env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
            env.reset()
            m, r, f = TM(B, torch.int32), TM(B, torch.float32), TM(B, torch.uint8)
            LD = m.stride(0)
            _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), LD, None))
            a = _lib.RolloutArgs(T=T, reset_first=1, d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(), d_flags_tb=f.data_ptr(),
                                 ld_masks=LD, ld_out=LD)
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
            if w == 'record':
                ag = torch.empty((B, 36), dtype=torch.float32, device=DEV)
                sf = torch.empty(B, dtype=torch.float32, device=DEV)
                for i in range(20):
                    sa = _lib.StepArgs(d_masks=m[i].data_ptr(), d_reward=r[i].data_ptr(), d_flags=f[i].data_ptr(),
                                       d_success_f32=sf.data_ptr(), d_ag=ag.data_ptr(), ld_ag=36)
                    _lib.check(L.bpg_step_record_v2(env._h, C.byref(sa), None))
            elif w in ('return', 'scan'):
                # three (reward, G) pairs round-robin: nothing of the previous launch is left in the 256 MiB cache
                rs = [r] + [TM(B, torch.float32) for _ in range(2)]
                for r_ in rs[1:]:
                    r_.copy_(r)
                Gs = [torch.empty((B, T), dtype=torch.float64, device=DEV) for _ in range(3)]
                fn = L.bpg_discounted_return_ld if w == 'return' else L.bpg_discounted_return_scan
                for i in range(15):
                    _lib.check(fn(P(rs[i % 3]), B, LD, T, 0.98, P(Gs[i % 3]), 0, None))
            elif w == 'episode':
                ag = torch.empty((B, T + 1, 36), dtype=torch.float32, device=DEV)
                gg = torch.empty((B, T, 36), dtype=torch.float32, device=DEV)
                sc = torch.empty((B, T, 1), dtype=torch.float32, device=DEV)
                a3 = _lib.RolloutArgs(T=T, reset_first=1, d_masks=m.data_ptr(), ld_masks=LD, d_ep_ag=ag.data_ptr(), d_ep_g=gg.data_ptr(),
                                      d_ep_success=sc.data_ptr())
                for i in range(6):
                    _lib.check(L.bpg_rollout_v2(env._h, C.byref(a3), None))
            else:
                n = 4 * B
                cag = torch.randint(-1, 2, (n, 36), device=DEV).to(torch.float32)
                cg = env._g[:1].expand(n, 36).contiguous()
                co = torch.empty(n, dtype=torch.float32, device=DEV)
                for i in range(10):
                    _lib.check(L.bpg_compute_reward_f32(P(cag), P(cg), n, 36, P(co), 0, None))
                for i in range(10):
                    _lib.check(L.bpg_compute_reward_f32_goal(P(cag), P(cg), n, 36, P(co), 0, None))
                del cag, cg
                n = 2 * B
                dag = torch.randint(-1, 2, (n, 36), device=DEV).to(torch.float64)
                dg = env._g[:1].to(torch.float64).expand(n, 36).contiguous()
                for i in range(10):
                    _lib.check(L.bpg_compute_reward_f64(P(dag), P(dg), n, 36, P(co), 0, None))
                for i in range(10):
                    _lib.check(L.bpg_compute_reward_f64_goal(P(dag), P(dg), n, 36, P(co), 0, None))
            torch.cuda.synchronize()
        elif w == 'assemble':
            env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
            env.reset()
            g = torch.Generator(device=DEV).manual_seed(0)
            phys = dict(grip_pos=torch.randn(B, 3, dtype=torch.float64, device=DEV, generator=g),
                        grip_velp=torch.randn(B, 3, dtype=torch.float64, device=DEV, generator=g),
                        gripper_state=torch.randn(B, 2, dtype=torch.float64, device=DEV, generator=g),
                        gripper_vel=torch.randn(B, 2, dtype=torch.float64, device=DEV, generator=g))
            for k in ('obj_pos', 'obj_rot', 'obj_velp', 'obj_velr'):
                phys[k] = torch.randn(B, 4, 3, dtype=torch.float64, device=DEV, generator=g)
            u = torch.rand(B, 4, device=DEV) * 3 - 1.5
            for i in range(10):
                env.assemble_observation(phys)
                env.set_action_vector(u)
                env.randomize_objects()
            torch.cuda.synchronize()
        elif w == 'reset':
            env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
            ng = (torch.arange(B, device=DEV, dtype=torch.int32) * 7) % 3
            obs = [torch.empty(B, 87, dtype=torch.float32, device=DEV) for _ in range(3)]
            for i in range(13):
                env.reset(num_grey=ng, observation_out=obs[i % 3])
            torch.cuda.synchronize()
        elif w == 'contacts':
            # geom-id pair lists (maxcon 16): one step at 1 M envs over 8 list sets (HBM-cold), and the lists of a recorded
            # rollout (T = 50, 262 144 envs, 0.94 GB per set, 2 sets) into the fused rollout's mask stream
            maxcon = 16
            names = bpg.default_geom_names(ENV_ID)[2]
            objs = [i for i, nm in enumerate(names) if nm and (nm == 'table' or nm.startswith('object') or 'finger' in nm)]
            gsel = torch.tensor(objs, dtype=torch.int16, device=DEV)
            env = bpg.make(ENV_ID, num_envs=B, device=DEV, seed=1)
            env.reset()
            sets = [(gsel[torch.randint(0, len(objs), (B, maxcon, 2), device=DEV)].contiguous(),
                     torch.randint(0, 9, (B,), dtype=torch.int32, device=DEV)) for _ in range(8)]
            r, f = torch.empty(B, dtype=torch.float32, device=DEV), torch.empty(B, dtype=torch.uint8, device=DEV)
            for i in range(24):
                p_, c_ = sets[i % 8]
                _lib.check(L.bpg_step_contacts(env._h, P(p_), P(c_), maxcon, P(r), P(f), None))
            torch.cuda.synchronize()
            del sets
            n = 262144
            e2 = bpg.make(ENV_ID, num_envs=n, device=DEV, seed=1)
            ng = (torch.arange(n, device=DEV, dtype=torch.int32) * 7) % 3
            tb = [(gsel[torch.randint(0, len(objs), (T, n, maxcon, 2), device=DEV)].contiguous(),
                   torch.randint(0, 9, (T, n), dtype=torch.int32, device=DEV), TM(n, torch.int32)) for _ in range(2)]
            for i in range(8):
                p_, c_, m_ = tb[i % 2]
                _lib.check(L.bpg_contacts_to_masks_tb(e2._h, P(p_), P(c_), T, maxcon, P(ng), P(m_), int(m_.stride(0)), None))
            torch.cuda.synchronize()
        elif w == 'nan':
            x = torch.zeros(4, B, 87, dtype=torch.float32, device=DEV)
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            for i in range(12):
                _lib.check(L.bpg_nan_flag_f32(P(x[i % 4]), B, 87, 87, P(flag), 0, None))
            torch.cuda.synchronize()
        else:
            raise SystemExit('unknown kernel %r' % w)
    print('done', which)


if __name__ == '__main__':
    main()
