#!/usr/bin/env python3
"""One process per GPU: every rank rolls out its contiguous shard of the env batch, the rollout launch writes the shard's
per-env episode fail counts (one byte per env: the episode return is its negative), ONE all-gather assembles them in global
env order and they are expanded to the float32 returns on the device (BASELINE configs[3]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_returns.py
    python examples/sharded_returns.py            # one process, no process group: the same code path, nothing gathered

On a one-GPU box the ranks can share the card for a rehearsal: BPG_SINGLE_DEVICE=1 ... --nproc-per-node 2 with the gloo
backend (RCCL refuses two ranks on one device).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import blockpuzzle_gym_amd as bpg  # noqa: E402
from blockpuzzle_gym_amd import distributed as D  # noqa: E402

TOTAL, T, ENV_ID = 4194304, 50, 'BlocksTouchVariation-v0'


def main():
    single = os.environ.get('BPG_SINGLE_DEVICE') == '1'
    if single:
        os.environ['LOCAL_RANK'] = '0'
    rank, world, local = D.init_from_env(backend='gloo' if single else None)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    lo, hi = D.shard_range(TOTAL, rank, world)
    dims = dict(o=87, u=4, g=36, info_is_success=1)
    worker = bpg.RolloutStudent(lambda num_envs: D.make_sharded(ENV_ID, TOTAL, rank, world, device=dev, seed=0), None, dims,
                                None, T, rollout_batch_size=hi - lo, output='torch', async_stats=True)
    for it in range(3):
        # contacts from the built-in synthetic source here; a physics backend passes masks=(T, B) or a contact_source
        ep = worker.generate_rollouts_fused(episode_fail_count=True)
        counts = D.all_gather_per_env(ep['episode_fail_count'], total_envs=TOTAL)     # (TOTAL,) uint8 on every rank: 1 B/env on the wire
        returns = D.episode_return_from_fail_count(counts)                             # (TOTAL,) float32, bit for bit d_episode_return
        rate = D.global_success_rate(worker.env.success_count()[0], hi - lo, as_tensor=True)     # one 16-byte all-reduce
        if rank == 0:
            print('rollout %d: %d envs over %d rank(s), mean episode return %.3f, global success rate %.4f'
                  % (it, returns.numel(), world, float(returns.mean()), float(rate)))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
