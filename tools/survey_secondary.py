#!/usr/bin/env python3
"""Every kernel of the library that has no row in profiles/README.md's table, at 1 M envs: us per launch and fraction of 8 TB/s
over its algorithmic bytes, inputs / outputs rotated over enough buffer sets that nothing is read from the Infinity Cache."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib  # noqa: E402

dev, T, L = torch.device('cuda', 0), 50, _lib.lib()
ENV_ID = 'BlocksTouchVariation-v0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731


def timed(name, nbytes, fns, reps=None):
    """fns: one closure per buffer set, called round-robin."""
    n = len(fns)
    reps = reps or max(3 * n, 12)
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fns[i % n]()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    us = min(best)
    print('%-58s %9.2f us  %8.1f MB  %.3f of 8 TB/s  (%d sets)' % (name, us, nbytes / 1e6, nbytes / (us * 1e-6) / 8e12, n), flush=True)


def nsets(nbytes):
    return max(2, int(2.5 * 256 * 2**20 // nbytes) + 1)


env = bpg.make(ENV_ID, num_envs=B, device=dev, seed=1)
ng = (torch.arange(B, device=dev, dtype=torch.int32) * 7) % 3
env.reset(num_grey=ng)
dimg, dimo = env.dimg, env.dimo
envs = [env] + [bpg.make(ENV_ID, num_envs=B, device=dev, seed=2 + k) for k in range(nsets(8 * B) - 1)]
for e in envs[1:]:
    e.reset(num_grey=ng)

# reset with the observation rows (k_reset<true>): 4 num_grey in, 8 state + 4 dimo out
obs = [torch.empty(B, dimo, dtype=torch.float32, device=dev) for _ in range(nsets(B * 4 * dimo))]
timed('reset + observation rows (explicit num_grey)', B * (12 + 4 * dimo),
      [lambda o=o: _lib.check(L.bpg_reset_obs(env._h, P(ng), None, P(o), dimo, None)) for o in obs])
timed('reset, state only (explicit num_grey)', B * 12, [lambda e=e: _lib.check(L.bpg_reset(e._h, P(ng), None, None)) for e in envs])
timed('reset, state only (generator-drawn, + counter bump)', B * 8, [lambda e=e: _lib.check(L.bpg_reset(e._h, None, None, None)) for e in envs])
for e in envs:
    e.reset(num_grey=ng)
del obs
# desired goal rows / achieved goal rows f64
g32 = [torch.empty(B, dimg, dtype=torch.float32, device=dev) for _ in range(nsets(B * 4 * dimg))]
timed('get_desired_goal_f32 (k_fill_goal)', B * 4 * dimg, [lambda o=o: _lib.check(L.bpg_get_desired_goal_f32(env._h, P(o), dimg, None)) for o in g32])
timed('get_achieved_goal_f32 (k_get_ag)', B * (8 + 4 * dimg), [lambda o=o: _lib.check(L.bpg_get_achieved_goal_f32(env._h, P(o), dimg, None)) for o in g32])
del g32
g64 = [torch.empty(B, dimg, dtype=torch.float64, device=dev) for _ in range(nsets(B * 8 * dimg))]
timed('get_achieved_goal_f64 (k_get_ag)', B * (8 + 8 * dimg), [lambda o=o: _lib.check(L.bpg_get_achieved_goal_f64(env._h, P(o), dimg, None)) for o in g64])
del g64
# success count, packed reward
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
timed('success_count', B * 4, [lambda e=e: _lib.check(L.bpg_success_count(e._h, P(cnt), None)) for e in envs])
ro = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(len(envs))]
timed('compute_reward_packed', B * 12,
      [lambda e=e, o=o: _lib.check(L.bpg_compute_reward_packed(P(e.state_now), P(e.state_ever), 1 << 9, 0, B, P(o), 0, None)) for e, o in zip(envs, ro)])
# synthetic masks (T, LD)
ms = [bpg.time_major_empty(T, B, torch.int32, dev) for _ in range(nsets(B * 4 * T))]
timed('synth_masks, T = 50 rows', B * (4 * T + 4), [lambda m=m: _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, P(m), int(m.stride(0)), None)) for m in ms])
del ms
# placement sampler (generator-drawn): 4 in, nb x 2 f64 + status out
nb = env.num_objs - 2
xy = [torch.empty(B, nb, 2, dtype=torch.float64, device=dev) for _ in range(nsets(B * 16 * nb))]
st = torch.empty(B, dtype=torch.int32, device=dev)
timed('randomize_objects (generator-drawn placements)', B * (4 + 16 * nb + 4), [lambda o=o: env.randomize_objects(out=o, status=st) for o in xy])
del xy
# compute_reward f64 rows / broadcast goal
n4 = 2 * B
ag = [torch.randint(-1, 2, (n4, dimg), device=dev).to(torch.float64) for _ in range(nsets(n4 * 8 * dimg))]
gg = env._g[:1].to(torch.float64).expand(n4, dimg).contiguous()
co = torch.empty(n4, dtype=torch.float32, device=dev)
timed('compute_reward_f64, 2 M rows of 36 (goal rows)', n4 * (16 * dimg + 4), [lambda a=a: _lib.check(L.bpg_compute_reward_f64(P(a), P(gg), n4, dimg, P(co), 0, None)) for a in ag])
timed('compute_reward_f64_goal, 2 M rows of 36 (one goal row)', n4 * (8 * dimg + 4), [lambda a=a: _lib.check(L.bpg_compute_reward_f64_goal(P(a), P(gg), n4, dimg, P(co), 0, None)) for a in ag])
del ag, gg
ag = [torch.randint(-1, 2, (2 * n4, dimg), device=dev).to(torch.float32) for _ in range(nsets(2 * n4 * 4 * dimg))]
co = torch.empty(2 * n4, dtype=torch.float32, device=dev)
g1 = env._g[:1].contiguous()
timed('compute_reward_f32_goal, 4 M rows of 36 (one goal row)', 2 * n4 * (4 * dimg + 4), [lambda a=a: _lib.check(L.bpg_compute_reward_f32_goal(P(a), P(g1), 2 * n4, dimg, P(co), 0, None)) for a in ag])
del ag
# mat2euler
mats = [torch.randn(B * 4, 3, 3, dtype=torch.float64, device=dev) for _ in range(nsets(B * 4 * 96))]
eu = torch.empty(B * 4, 3, dtype=torch.float64, device=dev)
timed('mat2euler_f64, 4 M matrices', B * 4 * 96, [lambda m=m: _lib.check(L.bpg_mat2euler_f64(P(m), B * 4, P(eu), 0, None)) for m in mats])
del mats
# ragged (CSR) contact lists -> masks
names = bpg.default_geom_names(ENV_ID)[2]
objs = [i for i, nm in enumerate(names) if nm and (nm == 'table' or nm.startswith('object') or 'finger' in nm)]
gsel = torch.tensor(objs, dtype=torch.int16, device=dev)
csr = []
for k in range(nsets(B * (4 * 4 + 8))):
    ncon = torch.randint(0, 9, (B,), device=dev)
    offs = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(ncon, 0)
    tot = int(offs[-1])
    csr.append((gsel[torch.randint(0, len(objs), (tot, 2), device=dev)].contiguous(), offs, tot, torch.empty(B, dtype=torch.int32, device=dev)))
avg = sum(c[2] for c in csr) / len(csr)
timed('contacts_csr_to_masks (0..8 contacts per env)', int(4 * avg) + B * (8 + 4 + 4),
      [lambda c=c: _lib.check(L.bpg_contacts_csr_to_masks(env._h, P(c[0]), P(c[1]), c[2], P(c[3]), None)) for c in csr])
del csr
# returns at a horizon without a register-resident instantiation (T = 40): the LDS-tile kernels
T2 = 40
rw = [bpg.time_major_empty(T2, B, torch.float32, dev) for _ in range(2)]
for r in rw:
    r.copy_(-(torch.rand(T2, B, device=dev) < 0.5).float())
Gs = [torch.empty(B, T2, dtype=torch.float64, device=dev) for _ in range(2)]
timed('discounted_return, T = 40 (64-accumulator register form)', B * 12 * T2,
      [lambda r=r, g=g: _lib.check(L.bpg_discounted_return_ld(P(r), B, int(r.stride(0)), T2, 0.98, P(g), 0, None)) for r, g in zip(rw, Gs)])
timed('discounted_return_scan, T = 40 (LDS-tile kernel)', B * 12 * T2,
      [lambda r=r, g=g: _lib.check(L.bpg_discounted_return_scan(P(r), B, int(r.stride(0)), T2, 0.98, P(g), 0, None)) for r, g in zip(rw, Gs)])
m40 = bpg.time_major_empty(T2, B, torch.int32, dev)
_lib.check(L.bpg_synth_masks_ld(env._h, 0, T2, P(m40), int(m40.stride(0)), None))
env2 = bpg.make(ENV_ID, num_envs=B, device=dev, seed=1, max_episode_steps=T2)


def pg40(g):
    a = _lib.RolloutArgs(T=T2, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m40.data_ptr(), ld_masks=int(m40.stride(0)), d_G=g.data_ptr(), gamma=0.98)
    _lib.check(L.bpg_rollout_v2(env2._h, C.byref(a), None))
timed('rollout + G from step-ok words, T = 40 (two launches)', B * (4 * T2 + 12 + 16 + 8 * T2), [lambda g=g: pg40(g) for g in Gs])
