"""world_size-2 gloo test of the multi-GPU aggregation path (CPU tensors)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from blockpuzzle_gym_amd import distributed as D
    r, w, _ = D.init_from_env(backend='gloo')
    start, stop = D.shard_range(total, r, w)
    # per-env statistic = a function of the GLOBAL env index, so the gather order is checkable
    local = torch.arange(start, stop, dtype=torch.float32) * 0.5
    succ = (torch.arange(start, stop) % 3 == 0)
    gathered = D.all_gather_per_env(local)
    rate = D.global_success_rate(succ.sum(), stop - start)
    # train.py:21-26 / :67-73: every logged statistic is averaged over the ranks (weighted by the number of values)
    avg = [D.mpi_average(0.25 + r), D.mpi_average([1.0, 2.0, 3.0] if r == 0 else [10.0]), D.mpi_average([] if r == 0 else [4.0])]
    q.put((r, gathered.tolist(), rate, avg))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('total', [10, 11])
def test_two_rank_gather_and_success_rate(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [0.5 * i for i in range(total)]
    expect_rate = sum(1 for i in range(total) if i % 3 == 0) / total
    for _, gathered, rate, avg in results:
        assert gathered == expect
        assert abs(rate - expect_rate) < 1e-12
        assert avg == [0.75, 4.0, 2.0]          # (0.25 + 1.25) / 2; (1 + 2 + 3 + 10) / 4; ([0.] + [4.]) / 2


def test_mpi_average_without_a_process_group():
    sys.path.insert(0, ROOT)
    from blockpuzzle_gym_amd import distributed as D
    assert D.mpi_average(3.0) == 3.0 and D.mpi_average([1.0, 2.0]) == 1.5 and D.mpi_average([]) == 0.0
