"""Two ranks, one process each, both on cuda:0 (the GPU box has a single card), gloo for the
collectives: the sharded path end to end on the real kernels -- contiguous env shards keyed by
global env index, no data-path collective, one all-gather / all-reduce per rollout."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV_ID, TOTAL, T, SEED = 'BlocksTouchVariation-v0', 40001, 50, 77


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rollout(env):
    import ctypes as C
    import blockpuzzle_gym_amd as bpg
    rew = torch.empty((T, env.num_envs), dtype=torch.float32, device=env.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=env.device)
    args = bpg._lib.RolloutArgs(T=T, reset_first=1, d_reward_tb=rew.data_ptr(), d_success_count=cnt.data_ptr())
    bpg._lib.check(bpg._lib.lib().bpg_rollout(env._h, C.byref(args), None))
    ret = rew.double().sum(0).float()                  # per-env undiscounted episode return
    succ = (env.state_ever < 0).float()
    return ret, succ, cnt


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from blockpuzzle_gym_amd import distributed as D
    r, w, _ = D.init_from_env(backend='gloo')
    env = D.make_sharded(ENV_ID, TOTAL, r, w, device='cuda:0', seed=SEED)
    ret, succ, cnt = _rollout(env)
    all_ret = D.all_gather_per_env(ret.cpu())
    all_succ = D.all_gather_per_env(succ.cpu())
    rate = D.global_success_rate(cnt.cpu()[0], env.num_envs)
    q.put((r, env.num_envs, all_ret.numpy(), all_succ.numpy(), rate))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_sharded_rollout_equals_single_batch():
    sys.path.insert(0, ROOT)
    import blockpuzzle_gym_amd as bpg
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = bpg.make(ENV_ID, num_envs=TOTAL, device='cuda:0', seed=SEED)
    ret, succ, cnt = _rollout(full)
    assert sorted(r[1] for r in results) == [TOTAL // 2, TOTAL - TOTAL // 2]
    for _, _, all_ret, all_succ, rate in results:
        np.testing.assert_array_equal(all_ret, ret.cpu().numpy())
        np.testing.assert_array_equal(all_succ, succ.cpu().numpy())
        assert rate == pytest.approx(float(cnt.item()) / TOTAL, abs=1e-12)


def test_bench_starts_its_own_ranks_and_prints_one_parseable_line():
    """`python bench.py --gpus 2` with no WORLD_SIZE set: the parent (which never touches the GPU) starts
    two fresh rank processes -- the reference's mpi_fork, train.py:113-119 -- here both on cuda:0 with gloo
    (one card on the box), relays rank 0's JSON line and exits 0.  --total-envs = BASELINE configs[3]'s
    per-GPU shard size x 1 (524 288 envs over the two ranks), all-gather of the per-env final flag bytes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--single-device', '--backend', 'gloo',
           '--total-envs', '524288', '--steps', '4', '--warmup', '2', '--prewarm', '4', '--collective', 'allgather',
           '--reduce-every', '2', '--no-extras', '--no-cpu-baseline', '--launch-timeout', '500']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert 'error' not in line
    assert line['n_gpus'] == 2 and line['rccl_world_size'] == 2 and line['backend'] == 'gloo'
    assert line['scaling'] == 'strong' and line['config']['total_envs'] == 524288
    assert line['config']['envs_per_gpu'] == 262144
    assert line['collective'] == 'allgather' and line['gathered_bytes_per_rollout'] == 524288
    km = line['kernel_ms_per_rank']
    assert len(km['all']) == 2 and 0 < km['min'] <= km['max']
    assert line['launcher']['rank_exit_codes'] == [0, 0]
    assert line['value'] > 0 and 0 <= line['config']['success_rate'] <= 1


def test_bench_under_torchrun_two_ranks_on_one_card():
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (WORLD_SIZE set: bench.py is a rank, it must not start ranks of its own).
    Two ranks on cuda:0 with gloo here; rank 0 prints the one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--single-device',
           '--backend', 'gloo', '--envs', '65536', '--steps', '4', '--warmup', '2', '--prewarm', '4', '--reduce-every', '2',
           '--no-extras', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = []
    for ln in out.stdout.splitlines():
        try:
            cand = json.loads(ln)
        except ValueError:
            continue
        if isinstance(cand, dict) and 'metric' in cand:
            lines.append(cand)
    assert len(lines) == 1, out.stdout[-2000:]
    assert 'Gloo' not in out.stdout and 'RCCL' not in out.stdout     # library chatter goes to stderr, not beside the line
    line = lines[0]
    assert 'error' not in line and 'launcher' not in line          # no self-launch under torchrun
    assert line['n_gpus'] == 2 and line['rccl_world_size'] == 2 and line['scaling'] == 'weak'
    assert line['config']['total_envs'] == 131072 and line['collective'] == 'allreduce'
    assert line['collective_bytes_per_rollout'] == 16 and len(line['kernel_ms_per_rank']['all']) == 2
