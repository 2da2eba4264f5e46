"""gloo tests of the multi-GPU aggregation path on CPU tensors: world_size 2 (equal and ragged shards) and the
world_size 8 of BASELINE configs[3]; every per-rollout statistic must cost exactly ONE collective."""
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


class CollectiveCounter:
    """Wraps the torch.distributed entry points the package may use and counts the calls."""
    NAMES = ('all_reduce', 'all_gather', 'all_gather_into_tensor', 'broadcast', 'reduce', 'all_to_all', 'gather', 'scatter',
             'all_gather_object', 'reduce_scatter_tensor')

    def __init__(self, dist):
        self.dist, self.calls, self.orig = dist, [], {}
        for n in self.NAMES:
            if hasattr(dist, n):
                self.orig[n] = getattr(dist, n)
                setattr(dist, n, self._wrap(n))

    def _wrap(self, n):
        def f(*a, **kw):
            self.calls.append(n)
            return self.orig[n](*a, **kw)
        return f

    def take(self):
        c, self.calls = self.calls, []
        return c

    def restore(self):
        for n, f in self.orig.items():
            setattr(self.dist, n, f)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from blockpuzzle_gym_amd import distributed as D
    r, w, _ = D.init_from_env(backend='gloo')
    start, stop = D.shard_range(total, r, w)
    assert D.shard_sizes_of(total, w)[r] == stop - start and sum(D.shard_sizes_of(total, w)) == total
    # per-env statistic = a function of the GLOBAL env index, so the gather order is checkable
    local = torch.arange(start, stop, dtype=torch.float32) * 0.5
    succ = (torch.arange(start, stop) % 3 == 0)
    cc = CollectiveCounter(dist)
    gathered = D.all_gather_per_env(local, total_envs=total)
    n_gather = cc.take()
    out = torch.full((total,), -1.0)
    res = D.all_gather_per_env(local, total_envs=total, out=out)
    into_out = bool(res is out) and cc.take() == ['all_gather_into_tensor']
    async_ok = None
    if total % w == 0:                       # equal shards: the collective writes straight into `out`, optionally async
        o2, work = D.all_gather_per_env(local, total_envs=total, out=torch.empty(total), async_op=True)
        work.wait()
        async_ok = o2.tolist() == gathered.tolist() and cc.take() == ['all_gather_into_tensor']
    two_d = D.all_gather_per_env(torch.stack([local, -local], 1), total_envs=total)       # trailing dimensions travel along
    cc.take()
    rate_t = D.global_success_rate(succ.sum(), stop - start, as_tensor=True)
    n_rate = cc.take()
    rate = D.global_success_rate(succ.sum(), stop - start)
    cc.take()
    # train.py:21-26 / :67-73: every logged statistic is averaged over the ranks (weighted by the number of values)
    avg = [D.mpi_average(0.25 + r), D.mpi_average([1.0, 2.0, 3.0] if r == 0 else [10.0]), D.mpi_average([] if r == 0 else [4.0])]
    n_avg = cc.take()
    cc.restore()
    try:
        D.all_gather_per_env(local)          # sizes are derived, never exchanged: total_envs is required
        needs_total = False
    except ValueError:
        needs_total = True
    q.put((r, gathered.tolist(), out.tolist(), into_out, async_ok, two_d[:, 1].tolist(), n_gather, n_rate, n_avg,
           torch.is_tensor(rate_t) and rate_t.dim() == 0, float(rate_t), rate, avg, needs_total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,total', [(2, 10), (2, 11), (8, 4096), (8, 4099)])
def test_gather_and_success_rate_cost_one_collective_each(world, total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [0.5 * i for i in range(total)]
    expect_rate = sum(1 for i in range(total) if i % 3 == 0) / total
    for (r, gathered, out, into_out, async_ok, neg, n_gather, n_rate, n_avg, is_t, rate_t, rate, avg, needs_total) in results:
        assert gathered == expect and out == expect and into_out
        assert neg == [-x for x in expect]
        assert n_gather == ['all_gather_into_tensor']            # ONE collective, also for ragged shards
        assert async_ok is (True if total % world == 0 else None)
        assert n_rate == ['all_reduce'] and is_t and abs(rate_t - expect_rate) < 1e-12 and abs(rate - expect_rate) < 1e-12
        assert n_avg == ['all_reduce'] * 3
        assert needs_total
        if world == 2:
            assert avg == [0.75, 4.0, 2.0]          # (0.25 + 1.25) / 2; (1 + 2 + 3 + 10) / 4; ([0.] + [4.]) / 2


def test_mpi_average_without_a_process_group():
    sys.path.insert(0, ROOT)
    from blockpuzzle_gym_amd import distributed as D
    assert D.mpi_average(3.0) == 3.0 and D.mpi_average([1.0, 2.0]) == 1.5 and D.mpi_average([]) == 0.0
    t = torch.arange(5.)
    assert D.all_gather_per_env(t) is t and D.all_gather_per_env(t, async_op=True) == (t, None)
    assert D.global_success_rate(torch.tensor(3), 4) == 0.75
