"""Times the drop-in RolloutStudent.generate_rollouts (policy in the loop, one step launch per
time step, batch-major episode buffers) with a device-resident random policy."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blockpuzzle_gym_amd as bpg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env_id = sys.argv[2] if len(sys.argv) > 2 else 'BlocksTouchVariation-v0'
T = 50


class RandomPolicy:
    device_tensors = True

    def get_actions(self, o, ag, g, **kw):
        return torch.rand(o.shape[0], 4, device=o.device) * 2 - 1


env = bpg.make(env_id, num_envs=B, device='cuda:0')
dims = dict(o=env.dimo, u=4, g=env.dimg, info_is_success=1)
w = bpg.RolloutStudent(lambda num_envs: env, RandomPolicy(), dims, None, T, rollout_batch_size=B)
for name, fn in (('generate_rollouts (per-step launches, policy in loop)', w.generate_rollouts),
                 ('generate_rollouts_graphed (same loop, hipGraph replay)', w.generate_rollouts_graphed),
                 ('generate_rollouts_fused (one launch)', w.generate_rollouts_fused)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        ep = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nbytes = sum(v.numel() * v.element_size() for v in ep.values())
    print('%-55s %8.2f ms/rollout  %7.2f G env-steps/s  episode %.1f GB' % (name, dt * 1e3, B * T / dt / 1e9, nbytes / 1e9))
    del ep
