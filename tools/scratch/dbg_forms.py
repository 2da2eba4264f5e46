import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
import blockpuzzle_gym_amd as bpg
from blockpuzzle_gym_amd import _lib
L = _lib.lib()
DEV = 'cuda:0'
for B in (1000, 300000, 600000):
    env = bpg.make('BlocksTouchVariation-v0', num_envs=B, device=DEV)
    T = 5
    masks = torch.zeros((3 * T, B), dtype=torch.int32, device=DEV)
    rew = torch.empty((3 * T, B), dtype=torch.float32, device=DEV)
    cnt = torch.zeros(4, dtype=torch.int64, device=DEV)
    for name, kw in (('cont', dict()), ('reset', dict(reset_first=1)), ('multi', dict(reset_first=1, rollouts=3)),
                     ('synth', dict(reset_first=1, d_masks=None))):
        a = _lib.RolloutArgs(T=T, d_masks=masks.data_ptr(), d_reward_tb=rew.data_ptr(), d_success_count=cnt.data_ptr())
        for k, v in kw.items():
            setattr(a, k, v)
        print(B, name, 'launch', flush=True)
        rc = L.bpg_rollout_v2(env._h, C.byref(a), None)
        print(B, name, 'rc', rc, L.bpg_last_error(), flush=True)
        torch.cuda.synchronize()
        print(B, name, 'done', cnt.tolist(), flush=True)
