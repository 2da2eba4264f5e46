"""Multi-GPU sharding of the env batch: one process per GPU, no data-path
collective.  Envs are independent (reference: separate objects rollout.py:33,
separate MPI ranks train.py:113-119), so the global env index range is split
into contiguous shards and each rank keeps its shard resident in its own HBM.
The only exchange is one collective per rollout for the aggregated statistics,
mirroring `success_rate = mean(successes[-1])` (rollout.py:163-167) and
`mpi_average` (train.py:21-26): RCCL (`nccl` backend) over xGMI on GPUs, `gloo`
on CPU for the tests.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib


def shard_range(total_envs, rank, world_size):
    """Contiguous [start, stop) of global env indices owned by `rank`; the first
    `total % world` ranks get one extra env."""
    if not (0 <= rank < world_size):
        raise ValueError('rank out of range')
    base, extra = divmod(int(total_envs), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (the
    torchrun contract).  Returns (rank, world_size, local_rank); single-process
    runs return (0, 1, 0) without creating a process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def rank_seed(seed, rank):
    """train.py:132-133: `rank_seed = seed + 1000000 * rank`."""
    return int(seed) + 1000000 * int(rank)


def make_sharded(env_id, total_envs, rank, world_size, device=None, seed=0, reference_launch=False, **kw):
    """This rank's shard of a `total_envs`-env job.  Env i of the job is keyed by seed + 1000 * i with i
    the GLOBAL env index (env_id_base = first index of the shard; the reference's
    `env.seed(seed + 1000 * idx)`, rollout.py:209-210), so results do not depend on how the batch is
    sharded.
    reference_launch=True mirrors the reference's own launch instead (train.py:132-133 +
    rollout.py:206-210): every rank seeds its worker with seed + 1000000 * rank and numbers its envs from 0,
    i.e. env j of rank r is keyed by seed + 1000000 * r + 1000 * j -- as in the reference, that depends on the
    number of ranks, and from 1000 envs per rank on the streams of neighbouring ranks overlap."""
    from .registration import make
    start, stop = shard_range(total_envs, rank, world_size)
    if reference_launch:
        return make(env_id, num_envs=stop - start, device=device, seed=rank_seed(seed, rank), env_id_base=0, **kw)
    return make(env_id, num_envs=stop - start, device=device, seed=seed, env_id_base=start, **kw)


class StreamShards:
    """Several env shards on ONE GPU, each with a stream of its own -- the same partition of the env index range that
    spreads a job over GPUs (contiguous shards keyed by global env index, no data-path exchange), applied once more
    inside a GPU.  The shards' launches are independent, so the hardware runs them side by side: while one shard's
    rollout launch ramps up or drains, the others stream, and the two ends of a launch that a single stream leaves
    exposed are covered by other shards' work.  Measured (tools/stream_shards.py,
    profiles/r04_stream_shards.log, launches back to back, nothing joined between rollouts): 524 288 envs as 2 x 262 144:
    38.6-39.6 us per rollout against 45.2-45.9 us for one handle (0.76-0.79 against 0.66-0.67 of the HBM peak); 1 048 576
    envs as 4 x 262 144: 79.9-80.8 us against 90.7-91.1 us (0.75-0.76 against 0.67); 4 194 304 envs as 16 x 262 144: 315-320 us
    against 355-363 us.  Shards of about 262 144 envs are the ones that share CUs well (256-lane blocks; the 1024-lane
    shape of larger shards fills a CU on its own).  The gain needs arrays of the shard's own and no per-rollout join: an
    event wait ahead of every launch costs more than the overlap gains (profiles/r04_substreams_in_library.log).

    shards[i] = (env, stream); `first[i]` = first global env index of shard i.  Per-env outputs of the shards are slices
    of ONE array when the caller hands each shard its slice (`slices(x)`), so a gather over GPUs still sees one
    contiguous local array.  Nothing synchronises the shards with each other between rollouts: join() makes the current
    stream wait for all of them (a stream-level wait), fork() makes all of them wait for the current stream."""

    def __init__(self, env_id, num_envs, shards, device=None, seed=0, env_id_base=0, **kw):
        from .registration import make
        dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.device, self.num_envs, self.shards, self.first = dev, int(num_envs), [], []
        for r in range(int(shards)):
            lo, hi = shard_range(num_envs, r, shards)
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                env = make(env_id, num_envs=hi - lo, device=dev, seed=seed, env_id_base=int(env_id_base) + lo, **kw)
            self.shards.append((env, st))
            self.first.append(lo)

    def __len__(self):
        return len(self.shards)

    def slices(self, x, dim=0):
        """x (num_envs along `dim`) -> the shards' views of it."""
        return [x.narrow(dim, lo, env.num_envs) for lo, (env, _) in zip(self.first, self.shards)]

    def fork(self):
        cur = torch.cuda.current_stream(self.device)
        for _, st in self.shards:
            st.wait_stream(cur)

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for _, st in self.shards:
            cur.wait_stream(st)

    def for_each(self, fn):
        """fn(i, env, stream) with shard i's stream current; returns the list of results."""
        out = []
        for i, (env, st) in enumerate(self.shards):
            with torch.cuda.stream(st):
                out.append(fn(i, env, st))
        return out


def global_success_counts(local_success_count, local_num_envs, device=None):
    """One all-reduce of (successes, envs) per rollout; the summed pair comes back as a 2-element float64 tensor ON THE
    DEVICE of `local_success_count` -- no host sync: the caller divides (or .item()s) where it synchronises anyway, once
    per epoch in the reference (train.py:67-76)."""
    c = torch.as_tensor(local_success_count)
    if device is not None:
        c = c.to(device)
    t = torch.stack([c.reshape(()).to(torch.float64),
                     torch.as_tensor(float(local_num_envs), dtype=torch.float64, device=c.device)])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def global_success_rate(local_success_count, local_num_envs, as_tensor=False):
    """`success_rate = np.mean(successful)` (rollout.py:163-167) over the envs of ALL ranks: one all-reduce of
    (successes, envs).  as_tensor=True returns the rate as a 0-dim device tensor without synchronising the host (use it
    per rollout); the default reads it back as a Python float (one host sync: per epoch, as train.py:67-76 logs it)."""
    t = global_success_counts(local_success_count, local_num_envs)
    rate = t[0] / t[1]
    return rate if as_tensor else float(rate.item())


def mpi_average(value, device=None):
    """train.py:21-26 (`mpi_average`, used for every logged statistic, :67-73): the mean over ALL ranks' values --
    a scalar or a list per rank (an empty list counts as [0.]), weighted by how many values a rank holds, as
    baselines' mpi_moments computes it -- with one all-reduce of (sum, count) over RCCL / gloo instead of MPI.
    `device`: where the two-element tensor lives (the rank's GPU for the nccl backend; default: the current CUDA
    device when the backend is nccl, else the CPU)."""
    import numpy as np
    if isinstance(value, (list, tuple)) and len(value) == 0:
        value = [0.]
    if not isinstance(value, (list, tuple)):
        value = [value]
    x = np.asarray([float(v) for v in value], np.float64)
    if device is None:
        device = (torch.device('cuda', torch.cuda.current_device())
                  if dist.is_initialized() and dist.get_backend() == 'nccl' else torch.device('cpu'))
    t = torch.tensor([x.sum(), float(x.size)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0].item() / t[1].item())


def shard_sizes_of(total_envs, world_size):
    """Sizes of the contiguous shards of shard_range(), rank order: known to every rank without communication."""
    return [b - a for a, b in (shard_range(total_envs, r, world_size) for r in range(world_size))]


def all_gather_per_env(local_values, total_envs=None, shard_sizes=None, out=None, async_op=False):
    """All-gather of a per-env statistic (the episode return bpg_rollout_args.d_episode_return writes, 4 B/env: 2 MB
    per GPU at 4 194 304 envs over 8 GPUs; or the final is_success) into global env order -- the ONE collective of a
    rollout (`mpi_average` of train.py:21-26 is its reduced form).

    The shard sizes follow from shard_range(total_envs, r, world) on every rank, so nothing is exchanged to learn them:
    pass `total_envs` (or the explicit `shard_sizes` list).  Equal shards (total % world == 0, BASELINE configs[3]):
    exactly one all_gather_into_tensor straight into `out` (a (total, ...) tensor, allocated if None) -- no padding, no
    concatenation, nothing but the collective touches the data.  Ragged shards: the shards are padded to the longest
    (at most one extra env), still one collective, and the result is compacted on the device.
    async_op=True (equal shards only) returns (out, work) without waiting.
    Without a process group the input is returned as is (copied into `out` when one is given, so the result is where
    the caller expects it on every path); a one-rank group still runs the collective (the rehearsal of the RCCL path on
    a single GPU)."""
    if not dist.is_initialized():
        if out is not None:              # the same contract as under a group: the result is in `out`
            out.copy_(local_values)
            local_values = out
        return (local_values, None) if async_op else local_values
    world = dist.get_world_size()
    n_local = local_values.shape[0]
    if shard_sizes is None:
        if total_envs is None:
            raise ValueError('all_gather_per_env needs total_envs (or shard_sizes): shard sizes are derived, not exchanged')
        shard_sizes = shard_sizes_of(total_envs, world)
    shard_sizes = [int(x) for x in shard_sizes]
    if len(shard_sizes) != world or shard_sizes[dist.get_rank()] != n_local:
        raise ValueError('shard_sizes %s do not match this rank\'s %d values' % (shard_sizes, n_local))
    tail = tuple(local_values.shape[1:])
    m = max(shard_sizes)
    if min(shard_sizes) == m:                                   # equal shards: the collective writes the result
        if out is None:
            out = torch.empty((world * m,) + tail, dtype=local_values.dtype, device=local_values.device)
        work = dist.all_gather_into_tensor(out, local_values.contiguous(), async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError('async_op needs equal shards')
    pad = torch.zeros((m,) + tail, dtype=local_values.dtype, device=local_values.device)
    pad[:n_local] = local_values
    buf = torch.empty((world * m,) + tail, dtype=local_values.dtype, device=local_values.device)
    dist.all_gather_into_tensor(buf, pad)
    res = torch.cat([buf[r * m:r * m + shard_sizes[r]] for r in range(world)])
    if out is not None:
        out.copy_(res)
        return out
    return res


def episode_return_from_fail_count(counts, out=None):
    """(n,) uint8 episode fail counts (bpg_rollout_args.d_episode_fail_count) -> (n,) float32 episode returns, the bits
    the rollout launch's own d_episode_return has: -(float)count, -0.0f for 0.  One launch on the current stream."""
    assert counts.dtype == torch.uint8 and counts.is_cuda and counts.is_contiguous()
    if out is None:
        out = torch.empty(counts.shape, dtype=torch.float32, device=counts.device)
    assert out.shape == counts.shape and out.dtype == torch.float32 and out.is_contiguous()
    _lib.check(_lib.lib().bpg_episode_return_from_fail_count(
        C.c_void_p(counts.data_ptr()), C.c_void_p(out.data_ptr()), counts.numel(), counts.device.index,
        C.c_void_p(torch.cuda.current_stream(counts.device).cuda_stream)))
    return out


class RcclComm:
    """The library's own RCCL communicator (include/bpgym.h: bpg_comm_*): collectives issued by libbpgym.so itself on the
    CALLER's stream -- the stream the rollout launch that produced the input ran on, so no event crosses streams -- and
    usable from a host that is not PyTorch (examples/c_abi_demo.cpp).  torch.distributed's nccl backend runs its
    collectives on a stream of its own and orders it against the caller's with events: 12-30 us of serialisation per
    collective per rollout (docs/HISTORY.md section 8), which a per-rollout gather pays every time.

    RcclComm(rank, world, device, id_bytes): every rank passes the same 128 bytes (rank 0's RcclComm.unique_id()).
    RcclComm.from_process_group(device): takes them from rank 0 of the initialised torch.distributed group (any backend:
    the bytes travel as a broadcast object) -- the process group is only the courier."""

    def __init__(self, rank, world, device, id_bytes):
        dev = torch.device(device)
        self.rank, self.world, self.device = int(rank), int(world), dev
        h = C.c_void_p()
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _lib.check(_lib.lib().bpg_comm_create(C.byref(h), self.rank, self.world, buf, dev.index))
        self._h = h

    @staticmethod
    def unique_id():
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.lib().bpg_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device):
        if dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [cls.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
        else:
            rank, world, box = 0, 1, [cls.unique_id()]
        return cls(rank, world, device, box[0])

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            _lib.lib().bpg_comm_destroy(h)

    __del__ = close

    def _stream(self, stream):
        return C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)

    def all_gather(self, local, out=None, stream=None):
        """(n_local,) uint8 or float32 -> (world * n_local,) in rank order (global env order for equal contiguous
        shards): ONE ncclAllGather on `stream` (default: the current stream)."""
        assert local.is_cuda and local.is_contiguous() and local.dim() == 1
        if out is None:
            out = torch.empty(self.world * local.numel(), dtype=local.dtype, device=local.device)
        assert out.numel() == self.world * local.numel() and out.dtype == local.dtype and out.is_contiguous()
        fn = {torch.uint8: _lib.lib().bpg_allgather_u8, torch.float32: _lib.lib().bpg_allgather_f32}[local.dtype]
        _lib.check(fn(self._h, C.c_void_p(local.data_ptr()), C.c_void_p(out.data_ptr()), local.numel(), self._stream(stream)))
        return out

    def all_reduce_sum_(self, counts, stream=None):
        """In-place sum over the ranks of an int64 / uint64 tensor (the [successes, envs] rows): one ncclAllReduce."""
        assert counts.is_cuda and counts.is_contiguous() and counts.element_size() == 8 and not counts.is_floating_point()
        _lib.check(_lib.lib().bpg_allreduce_sum_u64(self._h, C.c_void_p(counts.data_ptr()), counts.numel(), self._stream(stream)))
        return counts
