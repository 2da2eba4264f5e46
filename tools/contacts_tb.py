#!/usr/bin/env python3
"""bpg_contacts_to_masks_tb and bpg_step_contacts.  The first: the contact lists of a recorded rollout, (T, n, maxcon, 2) int16 + (T, n) int32, -> the fused
rollout's (T, ld) mask stream.  us per launch and fraction of 8 TB/s over the algorithmic bytes (the padded lists as stored:
T n (4 maxcon + 4) in, 4 T n out), inputs rotated so that none is read from the Infinity Cache."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import _lib  # noqa: E402

dev, T, L = torch.device('cuda', 0), 50, _lib.lib()
ENV_ID = 'BlocksTouchVariation-v0'
P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
cases = [(65536, 16), (262144, 16), (1048576, 16), (262144, 8), (262144, 32), (262144, 12)] if os.environ.get("BPG_TOOL_SKIP_TB") != "1" else []
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in c.split(':')) for c in sys.argv[1:]]
for n, maxcon in cases:
    env = bpg.make(ENV_ID, num_envs=n, device=dev, seed=0)
    names = bpg.default_geom_names(ENV_ID)[2]
    objs = [i for i, nm in enumerate(names) if nm and (nm == 'table' or nm.startswith('object') or 'finger' in nm)]
    gsel = torch.tensor(objs, dtype=torch.int16, device=dev)
    nbytes = T * n * (4 * maxcon + 8)
    nset = max(2, int(2.5 * 256 * 2**20 // nbytes) + 1)
    sets = []
    for k in range(nset):
        idx = torch.randint(0, len(objs), (T, n, maxcon, 2), device=dev)
        pairs = gsel[idx].contiguous()
        del idx
        ncon = torch.randint(0, min(maxcon, 8) + 1, (T, n), dtype=torch.int32, device=dev)
        sets.append((pairs, ncon, bpg.time_major_empty(T, n, torch.int32, dev)))
    ng = (torch.arange(n, device=dev, dtype=torch.int32) * 7) % 3
    it = [0]

    def go():
        p, c, m = sets[it[0] % nset]
        it[0] += 1
        _lib.check(L.bpg_contacts_to_masks_tb(env._h, P(p), P(c), T, maxcon, P(ng), P(m), int(m.stride(0)), None))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(4, min(40, int(3e9 // nbytes)))
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    us = min(best)
    print('T=%d n=%8d maxcon=%2d  %9.1f us per launch  %6.1f MB  %.3f of 8 TB/s   (%d input sets, runs %s)'
          % (T, n, maxcon, us, nbytes / 1e6, nbytes / (us * 1e-6) / 8e12, nset, ['%.1f' % b for b in best]), flush=True)
    del sets, env
    torch.cuda.empty_cache()

# bpg_step_contacts (one env.step with the lists as input), lists rotated so that they come from HBM
print('bpg_step_contacts')
for n, maxcon in [(65536, 16), (1048576, 16), (1048576, 8), (1048576, 32)]:
    env = bpg.make(ENV_ID, num_envs=n, device=dev, seed=0)
    env.reset()
    names = bpg.default_geom_names(ENV_ID)[2]
    objs = [i for i, nm in enumerate(names) if nm and (nm == 'table' or nm.startswith('object') or 'finger' in nm)]
    gsel = torch.tensor(objs, dtype=torch.int16, device=dev)
    nbytes = n * (4 * maxcon + 4 + 8 + 13)
    nset = max(3, int(2.5 * 256 * 2**20 // nbytes) + 1)
    sets = []
    for k in range(nset):
        pairs = gsel[torch.randint(0, len(objs), (n, maxcon, 2), device=dev)].contiguous()
        sets.append((pairs, torch.randint(0, min(maxcon, 8) + 1, (n,), dtype=torch.int32, device=dev),
                     torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)))
    it = [0]

    def go():
        p, c, r, f = sets[it[0] % nset]
        it[0] += 1
        _lib.check(L.bpg_step_contacts(env._h, P(p), P(c), maxcon, P(r), P(f), None))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * nset
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    us = min(best)
    print('n=%8d maxcon=%2d  %8.2f us per launch  %6.1f MB  %.3f of 8 TB/s   (%d input sets)' % (n, maxcon, us, nbytes / 1e6, nbytes / (us * 1e-6) / 8e12, nset), flush=True)
    del sets, env
    torch.cuda.empty_cache()
