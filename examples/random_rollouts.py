"""Minimal use of the batched env and the rollout workers (needs one MI355X).

    python examples/random_rollouts.py [num_envs]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blockpuzzle_gym_amd as bpg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536

# 1. gym-style: drop-in single env (numpy, reference dtypes) --------------------------------
env = bpg.make('BlocksTouch-v0')                       # gym.make('BlocksTouch-v0')
obs = env.reset()
finger, cube0, cube1 = 17, 23, 24                      # geom ids of the 2-block scene (env.id2obj[0])
obs, reward, done, info = env.step(env.action_space.sample(),
                                   contacts=(np.array([[[cube0, cube1], [finger, cube0]]], np.int16),
                                             np.array([2], np.int32)))
print('single env: reward %r  done %r  info %r' % (reward, done, info))
print('achieved_goal matrix:\n', obs['achieved_goal'].reshape(4, 4))

# 2. batched: B envs per launch, torch tensors on the GPU --------------------------------------
benv = bpg.make('BlocksTouchVariation-v0', num_envs=B, seed=0)      # synthetic contact source
benv.reset()
for t in range(50):
    obs, reward, done, info = benv.step(torch.rand(B, 4, device=benv.device) * 2 - 1)
print('batched: %d envs, success rate after 50 steps %.4f, all done: %s'
      % (B, info['is_success'].float().mean().item(), bool(done.all())))

# 3. the reference's rollout worker signature, device-resident policy ----------------------------
class RandomPolicy:
    device_tensors = True

    def get_actions(self, o, ag, g, **kw):
        return torch.rand(o.shape[0], 4, device=o.device) * 2 - 1


dims = bpg.configure_dims(bpg.make('BlocksTouchVariation-v0'))
worker = bpg.RolloutStudent(lambda num_envs: bpg.make('BlocksTouchVariation-v0', num_envs=num_envs),
                            RandomPolicy(), dims, logger=None, T=50, rollout_batch_size=B)
episode = worker.generate_rollouts()                   # policy in the loop, one launch per step
print('episode:', {k: tuple(v.shape) for k, v in episode.items()}, worker.logs('train'))
episode = worker.generate_rollouts_fused()             # whole rollout in one launch
print('fused success rate %.4f' % worker.current_success_rate())

# 4. the evaluator's test loop (train.py:62-64) as ONE launch: 10 rollouts on recorded contacts ----------------
K, T = 10, 50
masks = bpg.time_major_empty(K * T, B, torch.int32, worker.device)           # (K*T, B) contact words, rows padded
bpg._lib.check(bpg._lib.lib().bpg_synth_masks_ld(worker.env._h, 0, K * T, masks.data_ptr(), int(masks.stride(0)), None))
worker.clear_history()
rates = worker.evaluate_rollouts_fused(masks, n_rollouts=K)
print('evaluator: %d rollouts in one launch, success rates %s -> %s' % (K, ['%.4f' % r for r in rates[:3]], worker.logs('test')))
