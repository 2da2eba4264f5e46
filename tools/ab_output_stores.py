#!/usr/bin/env python3
"""A/B of the output stores of the one-shot kernels: the library as built (nontemporal, BPG_OUT_NT = 1) against the same
source compiled with -DBPG_OUT_NT=0 (plain stores), same box, same process order, alternating.

    python -c "import importlib.util as u; s = u.spec_from_file_location('b', 'blockpuzzle-gym_amd/build.py'); b = u.module_from_spec(s); \
               s.loader.exec_module(b); b.build(force=True, extra_flags=['-DBPG_OUT_NT=0'], out='tools/libbpgym_plainout.so')"     (build container)
    python tools/ab_output_stores.py                      (GPU box; runs itself once per library, twice, alternating)

Per kernel and batch size: HIP events around a train of back-to-back launches on rotating output buffers; us per launch.
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = {'nontemporal': os.path.join(ROOT, 'blockpuzzle-gym_amd', 'lib', 'libbpgym.so'),
        'plain': os.path.join(ROOT, 'tools', 'libbpgym_plainout.so')}
ENV_ID, T = 'BlocksTouchVariation-v0', 50


def child(which):
    from blockpuzzle_gym_amd import _lib
    _lib.LIB_PATH = LIBS[which]
    import torch
    import blockpuzzle_gym_amd as bpg
    dev = torch.device('cuda', 0)
    L = _lib.lib()
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    res = {}

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = []
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / reps * 1e3)
        return min(best)

    for B in (65536, 1048576):
        env = bpg.make(ENV_ID, num_envs=B, device=dev, seed=1)
        env.reset()
        nrot = 3 if B > 100000 else 8
        it = [0]

        def rot(bufs):
            it[0] += 1
            return bufs[it[0] % len(bufs)]
        g = torch.Generator(device=dev).manual_seed(0)
        ph = _lib.PhysState(dt=0.04)
        keep = []
        for name, shape in (('d_grip_pos', (B, 3)), ('d_grip_velp', (B, 3)), ('d_gripper_state', (B, 2)), ('d_gripper_vel', (B, 2)),
                            ('d_obj_pos', (B, 4, 3)), ('d_obj_rot', (B, 4, 3)), ('d_obj_velp', (B, 4, 3)), ('d_obj_velr', (B, 4, 3))):
            t = torch.randn(shape, dtype=torch.float64, device=dev, generator=g)
            keep.append(t)
            setattr(ph, name, t.data_ptr())
        obs = [torch.empty((B, 87), dtype=torch.float32, device=dev) for _ in range(nrot)]
        res['assemble_obs_f32 %d' % B] = timed(lambda: _lib.check(L.bpg_assemble_obs_f32(env._h, C.byref(ph), P(rot(obs)), 87, None)), 30)
        ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
        res['reset + obs rows %d' % B] = timed(lambda: _lib.check(L.bpg_reset_obs(env._h, P(ng), None, P(rot(obs)), 87, None)), 30)
        u = torch.rand(B, 4, device=dev) * 3 - 1.5
        out9 = [torch.empty((B, 9), dtype=torch.float64, device=dev) for _ in range(nrot)]
        res['set_action_f64 %d' % B] = timed(lambda: _lib.check(L.bpg_set_action_f64(env._h, P(u), P(rot(out9)), None)), 30)
        ag = [torch.empty((B, 36), dtype=torch.float32, device=dev) for _ in range(nrot)]
        res['get_achieved_goal_f32 %d' % B] = timed(lambda: _lib.check(L.bpg_get_achieved_goal_f32(env._h, P(rot(ag)), 36, None)), 30)
        res['get_desired_goal_f32 %d' % B] = timed(lambda: _lib.check(L.bpg_get_desired_goal_f32(env._h, P(rot(ag)), 36, None)), 30)
        r = bpg.time_major_empty(T, B, torch.float32, dev)
        m = bpg.time_major_empty(T, B, torch.int32, dev)
        ld = int(m.stride(0))
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), ld, None))
        a = _lib.RolloutArgs(T=T, reset_first=1, d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(), ld_masks=ld, ld_out=ld)
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a), None))
        G = [torch.empty((B, T), dtype=torch.float64, device=dev) for _ in range(nrot)]
        res['discounted_return %d' % B] = timed(lambda: _lib.check(L.bpg_discounted_return_ld(P(r), B, ld, T, 0.98, P(rot(G)), 0, None)), 12)
        res['discounted_return_scan %d' % B] = timed(lambda: _lib.check(L.bpg_discounted_return_scan(P(r), B, ld, T, 0.98, P(rot(G)), 0, None)), 12)

        def pg():
            a2 = _lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), ld_masks=ld, d_G=rot(G).data_ptr(), gamma=0.98)
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(a2), None))
        res['rollout + G from step-ok words %d' % B] = timed(pg, 12)
        sf = torch.empty(B, dtype=torch.float32, device=dev)
        def record():
            sa = _lib.StepArgs(d_masks=m[3].data_ptr(), d_reward=r[3].data_ptr(), d_success_f32=sf.data_ptr(),
                               d_ag=rot(ag).data_ptr(), ld_ag=36)
            _lib.check(L.bpg_step_record_v2(env._h, C.byref(sa), None))
        res['step_record (nontemporal in both) %d' % B] = timed(record, 30)
        fl = torch.empty(B, dtype=torch.uint8, device=dev)
        res['step_masks, same batch (state cache-resident) %d' % B] = timed(
            lambda: _lib.check(L.bpg_step_masks(env._h, P(m[5]), P(r[5]), P(fl), None)), 100)
        sets = []
        for k in range(13 if B > 100000 else 64):
            e2 = bpg.make(ENV_ID, num_envs=B, device=dev, seed=100 + k)
            e2.reset()
            sets.append((e2, m[k % T].clone(), torch.empty(B, dtype=torch.float32, device=dev)))

        def cold():
            e2, mm, rr = rot(sets)
            _lib.check(L.bpg_step_masks(e2._h, P(mm), P(rr), P(fl), None))
        res['step_masks, batches round-robin (HBM-cold) %d' % B] = timed(cold, 130)
        del env, obs, out9, ag, G, r, m, keep, sets
        torch.cuda.empty_cache()
    print(json.dumps(res))


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    runs = {k: [] for k in LIBS}
    for rep in range(2):
        for which in ('plain', 'nontemporal'):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), which], capture_output=True, text=True, check=True).stdout
            runs[which].append(json.loads(out.strip().splitlines()[-1]))
    print('%-46s %22s %22s' % ('us per launch (two processes each, alternating)', 'plain stores', 'nontemporal stores'))
    for k in runs['plain'][0]:
        p = [r[k] for r in runs['plain']]
        n = [r[k] for r in runs['nontemporal']]
        print('%-46s %10.2f %10.2f  %10.2f %10.2f   %+5.1f %%' % (k, p[0], p[1], n[0], n[1], (min(n) / min(p) - 1) * 100))


if __name__ == '__main__':
    main()
