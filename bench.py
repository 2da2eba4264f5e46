#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched gym_blocks step()/reset() hot
path at 1 048 576 envs per GPU (BASELINE.json metric), one JSON line on rank 0.

A bench "step" = one rollout pass over the batch in ONE kernel launch: reset of
all envs, T=50 fused env steps (contacts read as packed masks resident in HBM --
the contacts-in boundary -- rewards + success/done flags written per env-step)
and the success-count reduction; with N>1 GPUs the (successes, envs) rows are
all-reduced over RCCL in slabs of `--reduce-every` rollouts (default 50 = one epoch of the
reference's training loop, which averages across ranks once per epoch), asynchronously.
Weak scaling by default: every GPU holds `--envs` envs (BASELINE configs[2] per GPU).
`--total-envs E` fixes the JOB size instead and shards it over the ranks (strong scaling).
With N>1 the run has a SECOND timed leg, reported under `extra.config3_as_written`: BASELINE configs[3] as written --
4 194 304 envs in total sharded over the N ranks (524 288 per GPU at N = 8), the rollout launch writes every env's
episode fail count (1 B/env, bpg_rollout_args.d_episode_fail_count: the episode return is its negative; `--gather-dtype f32`
moves the float32 return, 4 B/env, instead), the ranks all-gather those bytes over RCCL once per rollout and expand them to the
float32 returns on the device (`--gpus 8 --total-envs 4194304 --collective allgather --reduce-every 1` makes that leg the
headline instead).  A THIRD and a FOURTH leg, `extra.config3_cabi_collectives[_side_stream]`, repeat it with the collectives issued by
the library itself (bpg_comm_*, include/bpgym.h) instead of torch.distributed: on the launch stream, and on a second stream ordered
by events so that the exchange runs beside the next launch; a watchdog gives them up after --cabi-leg-timeout seconds.
TIMING: `value` = total env-steps of the K timed steps / MAX over ranks of the HIP-event span on the launch stream that covers
them and the completion of their last exchange, measured inside the barrier + synchronize bracket (see rollout_leg); the wall
clock of the bracket is reported beside it (`timing`).  N = 1 and N = 8 are measured the same way.
At N = 1 the line also carries `roofline_sharded`: the same batch run as stream shards (views of the batch on streams of their
own, with arrays of their own and on column ranges of the headline's arrays).

    python bench.py [--gpus N --steps K --warmup W] [--total-envs E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself, the
way the reference's `mpi_fork` does (gym_blocks/train.py:113-119: the parent re-launches the program once
per rank and only waits): N fresh child processes of this file with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, the parent never touches the GPU (no HIP call, no torch import), relays rank 0's JSON line
and exits non-zero if any rank does.  A rank that fails prints a JSON line with an "error" key instead of
dying on a bare assert.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ENV_ID = 'BlocksTouchVariation-v0'
CONFIG3_TOTAL = 4194304        # BASELINE.json configs[3]: 4 194 304 envs over the GPUs of one node


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--envs', type=int, default=1 << 20, help='envs per GPU (weak scaling)')
    p.add_argument('--total-envs', type=int, default=0,
                   help='total envs of the job, sharded over the ranks (strong scaling); overrides --envs')
    p.add_argument('--T', type=int, default=50)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--buffers', type=int, default=4, help='independent mask/output buffer sets used round-robin')
    p.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    p.add_argument('--collective', choices=['allreduce', 'allgather'], default='allreduce',
                   help='per-rollout exchange at N>1: all-reduce of (successes, envs) [16 B] or all-gather of every '
                        'env\'s episode return [4 B/env, written by the rollout launch]')
    p.add_argument('--gather-dtype', choices=['u8', 'f32'], default='u8',
                   help='all-gather legs: what travels per env -- u8 = the episode fail count (bpg_rollout_args.d_episode_fail_count, '
                        '1 B/env, expanded to the float32 return on the device after the gather), f32 = the float32 return itself')
    p.add_argument('--collective-impl', choices=['torch', 'cabi', 'cabi_side'], default='torch',
                   help='who issues the collectives: torch.distributed (its nccl backend runs them on a stream of its own, ordered '
                        'against the launch stream with events) or the library (bpg_comm_*: ncclAllGather / ncclAllReduce on the '
                        'launch stream itself)')
    p.add_argument('--no-cabi-leg', action='store_true',
                   help='N>1: skip the extra configs[3] leg with the library-issued collectives (extra.config3_cabi_collectives)')
    p.add_argument('--cabi-leg-timeout', type=float, default=90.0,
                   help='seconds before the extra library-collective leg is given up (the result line is printed without it)')
    p.add_argument('--config3-total', type=int, default=CONFIG3_TOTAL,
                   help='N>1: total envs of the second timed leg (BASELINE configs[3] as written: sharded over the ranks, '
                        'all-gather of the episode returns once per rollout), reported under extra.config3_as_written')
    p.add_argument('--no-config3', action='store_true', help='N>1: skip that second leg')
    p.add_argument('--reduce-every', type=int, default=50,
                   help='N>1: rollouts per collective (the statistics rows of R rollouts travel in one all-reduce; the '
                        'reference averages across ranks once per epoch of n_cycles = 50 rollouts, train.py:21-26,67-76, '
                        'config.py:39)')
    p.add_argument('--single-device', action='store_true',
                   help='rehearsal only: every rank uses cuda:0 (use with --backend gloo on a 1-GPU box)')
    p.add_argument('--default-stream', action='store_true',
                   help='launch on the legacy default stream instead of a dedicated non-blocking stream')
    p.add_argument('--prewarm', type=int, default=200,
                   help='untimed rollouts run during set-up, before the W warm-up steps, so that the clocks have '
                        'settled when the contract\'s warm-up starts (0.02 s at the default)')
    p.add_argument('--contiguous-rows', action='store_true',
                   help='time-major streams with rows exactly B elements apart (default: bpg_row_stride(B))')
    p.add_argument('--no-extras', action='store_true', help='skip the secondary kernels (step / episode / synth)')
    p.add_argument('--no-sharded', action='store_true',
                   help='skip roofline_sharded (the batch as stream shards: views of the headline batch on streams of their own)')
    p.add_argument('--stride-pads', default='0,256,768', help='row pads (elements) of the row-stride probe')
    p.add_argument('--no-stride-probe', action='store_true',
                   help='skip the set-up probe of the headline launch at row strides B, B + 256, B + 768')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-box-state', action='store_true', help='skip the rocm-smi reading under load after the measurement')
    p.add_argument('--cpu-seconds', type=float, default=12.0)
    p.add_argument('--launch-timeout', type=float, default=900.0,
                   help='self-launched N>1 runs: seconds before the parent ends the rank processes it started')
    return p.parse_args()


def cpu_baseline(T, seconds):
    """The oracle (matrix-form C port of the reference algorithm) timed on ONE host core on a
    bounded sample of the same workload as the GPU headline: same variant and T, contact masks of
    the same synthetic source generated beforehand (untimed) and read from memory, reward float32 +
    flags byte written per env-step."""
    import ctypes as C
    import numpy as np
    from oracle import bpg_oracle as O
    L = O.lib()
    variant = O.VARIANTS[ENV_ID]
    P = C.POINTER

    def run(envs, seed):
        masks = np.empty((T, envs), np.uint32)
        L.orc_fill_masks(seed, 0, envs, T, 6, masks.ctypes.data_as(P(C.c_uint32)))
        rew = np.empty((T, envs), np.float32)
        flg = np.empty((T, envs), np.uint8)
        t0 = time.perf_counter()
        L.orc_bench_rollout_masks(variant, envs, T, 2, masks.ctypes.data_as(P(C.c_uint32)),
                                  rew.ctypes.data_as(P(C.c_float)), flg.ctypes.data_as(P(C.c_uint8)))
        return time.perf_counter() - t0

    dt = run(8192, 0)
    envs = int(max(8192, min(4 << 20, 8192 * seconds / max(dt, 1e-6))))
    dt = run(envs, 0)
    out = dict(value=envs * T / dt, unit='env-steps/s', cores=1, kind='port',
               sample='%d envs x %d steps of %s, pre-generated synthetic contact masks read from memory, '
                      '%.1f s on 1 thread (oracle/bpg_oracle.c matrix-form port)' % (envs, T, ENV_ID, dt))
    # context: the same port on every host core the process may use (ctypes releases the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box's CPU share per GPU
        per = max(8192, int(envs * 3.0 / max(dt, 1e-6)))   # ~3 s of work per thread
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            times = list(ex.map(lambda k: run(per, k), range(cores)))
        wall = time.perf_counter() - t0
        out['all_cores'] = dict(value=cores * per * T / max(times), unit='env-steps/s', cores=cores,
                                sample='%d threads x %d envs x %d steps, slowest thread %.1f s (wall incl. untimed '
                                       'mask generation %.1f s)' % (cores, per, T, max(times), wall))
    except Exception as e:  # noqa: BLE001
        out['all_cores'] = {'error': repr(e)[:200]}
    # the reference's OWN Python on one core, measured in the build container by tests/golden/time_reference.py (the
    # reference cannot travel to this box): quoted verbatim
    rf = os.path.join(ROOT, 'profiles', 'reference_cpu_timing.json')
    if os.path.exists(rf):
        try:
            out['reference_python'] = json.load(open(rf))
        except ValueError as e:
            out['reference_python'] = {'error': repr(e)[:200]}
    return out


def box_state_under_load(cx, h, seconds=0.6):
    """Untimed, after the measurement: the clocks and the power of the card WHILE the headline launch runs back to back (a train
    of launches is enqueued, `rocm-smi` is read once while it drains).  The headline launch takes 82-91 us depending on the box
    (docs/HISTORY.md 4): this records what the box the bench ran on looked like.  Best effort -- {'error': ...} if rocm-smi is missing."""
    torch, _lib, L = cx.torch, cx._lib, cx.L
    try:
        env, T, LD, NBUF = h['env'], cx.T, h['LD'], h['NBUF']
        ng = ((torch.arange(h['B'], device=cx.dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
        cnt = torch.zeros(2, dtype=torch.int64, device=cx.dev)
        args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=h['masks'][k].data_ptr(),
                                 d_reward_tb=h['reward'][k].data_ptr(), d_flags_tb=h['flags'][k].data_ptr(), ld_masks=LD, ld_out=LD,
                                 d_success_count=cnt.data_ptr()) for k in range(NBUF)]
        n = int(seconds / (h['kernel_ms'] * 1e-3))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i % NBUF]), cx.stream))
        e1.record()
        t0 = time.perf_counter()
        smi = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showmemorypartition', '--showcomputepartition', '--showvbios',
                              '--showdriverversion', '--json'],
                             capture_output=True, text=True, timeout=20)
        read_after_s = time.perf_counter() - t0
        still_running = not e1.query()
        torch.cuda.synchronize(cx.dev)
        card, js = {}, None
        try:
            js = json.loads(smi.stdout)
            card = js.get('card0') or next(iter(js.values()))
        except (ValueError, StopIteration):
            card = {'raw': smi.stdout[:400]}
        keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ('clk', 'clock', 'power', 'partition', 'vbios'))}
        if isinstance(js, dict) and isinstance(js.get('system'), dict):
            keep.update({k: v for k, v in js['system'].items() if 'driver' in k.lower()})
        return {'rocm_smi': keep, 'launches_in_the_train': n, 'train_ms_per_launch': e0.elapsed_time(e1) / n,
                'smi_read_took_s': read_after_s, 'train_still_running_when_read': bool(still_running),
                'how': 'a train of headline launches enqueued, rocm-smi read once while it drains; untimed, after the measurement'}
    except Exception as e:  # noqa: BLE001
        try:
            torch.cuda.synchronize(cx.dev)
        except Exception:  # noqa: BLE001
            pass
        return {'error': repr(e)[:300]}


def sharded_roofline(cx, h, steps, warmup):
    """The headline batch as STREAM SHARDS (untimed part of the run, after the measurement; `value` never uses it):
    S = auto_shards(B) views of the headline env (bpg_create_view: contiguous env ranges that share its packed state), each on a
    stream of its own, launching the headline's own launch (one rollout per launch) K times back to back; the shards are forked
    from / joined into the launch stream ONCE around the K steps.  us per rollout of the WHOLE batch = event span / K, algorithmic
    bytes per rollout as in `roofline`: B * (9 T + 12).
      own_arrays     every shard reads and writes (T, n_shard) arrays of its own (the top-level figure: what a pipeline that
                     owns its arrays per shard gets, e.g. distributed.StreamShards);
      shared_arrays  every shard works on its column range of the headline's (T, B) arrays (what a call on the caller's arrays
                     could be turned into: RolloutStudent.evaluate_rollouts_fused(shards=...), off by default -- it gains nothing).
    """
    torch, _lib, L, dev, T = cx.torch, cx._lib, cx.L, cx.dev, cx.T
    import blockpuzzle_gym_amd as bpg
    B, LD, NBUF, env = h['B'], h['LD'], h['NBUF'], h['env']
    S = bpg.rollout.auto_shards(B)
    nbytes = B * (T * 9 + 12)
    out = {'shards': S, 'envs_per_shard': B // max(S, 1), 'bytes_per_rollout': nbytes, 'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'}

    def figure(ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {'ms_per_rollout': ms, 'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS, 'env_steps_per_sec': B * T / (ms * 1e-3)}
    if S < 2:
        out['note'] = 'a batch of %d envs is not split (auto_shards)' % B
        return out
    cur = torch.cuda.current_stream(dev)
    ng = ((torch.arange(B, device=dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
    vs = bpg.rollout._ViewShards(env, S)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    for layout in ('shared_arrays', 'own_arrays'):
        vargs, keep = [], []
        for hv, st, lo, n in vs.views:
            if layout == 'shared_arrays':
                sets = [(h['masks'][k].data_ptr() + 4 * lo, h['reward'][k].data_ptr() + 4 * lo, h['flags'][k].data_ptr() + lo, LD)
                        for k in range(NBUF)]
            else:
                ldv = int(L.bpg_row_stride(n))
                sets = []
                for k in range(NBUF):
                    m = torch.empty((T, ldv), dtype=torch.int32, device=dev)[:, :n]
                    m.copy_(h['masks'][k][:, lo:lo + n])
                    r = torch.empty((T, ldv), dtype=torch.float32, device=dev)[:, :n]
                    f = torch.empty((T, ldv), dtype=torch.uint8, device=dev)[:, :n]
                    keep.append((m, r, f))
                    sets.append((m.data_ptr(), r.data_ptr(), f.data_ptr(), ldv))
            vargs.append([_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr() + 4 * lo, d_masks=pm, d_reward_tb=pr, d_flags_tb=pf,
                                           ld_masks=ld, ld_out=ld, d_success_count=cnt.data_ptr()) for pm, pr, pf, ld in sets])

        def train(n_it, first):
            vs.fork()
            for i in range(n_it):
                for (hv, st, lo, n), va in zip(vs.views, vargs):
                    _lib.check(L.bpg_rollout_v2(hv, C.byref(va[(first + i) % NBUF]), C.c_void_p(st.cuda_stream)))
            vs.join()
        train(max(warmup, 10), 0)
        torch.cuda.synchronize(dev)
        runs = []
        for g in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            train(steps, g * steps)
            e1.record(cur)
            torch.cuda.synchronize(dev)
            runs.append(e0.elapsed_time(e1) / steps)
        out[layout] = dict(figure(min(runs)), runs_ms=runs, launches_per_shard=steps)
        del keep, vargs
    vs.close()
    torch.cuda.empty_cache()
    # ---- the same through the package's worker: RolloutStudent.evaluate_rollouts_fused (the evaluator's test loop, train.py:63-64:
    # K rollouts per call) on per-shard arrays of the shards' own (worker.shard_streams), one launch of K rollouts per shard, forked /
    # joined per call; timed from outside with everything the call does (the K success counts read back); the one launch beside it
    try:
        dims = dict(o=env.dimo, u=4, g=env.dimg, info_is_success=1)
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=dev, seed=cx.a.seed), None, dims, None, T,
                               rollout_batch_size=B, output='torch')
        K = min(steps, 20)
        ngk = ng[None].expand(K, B).contiguous()
        one_m = torch.empty((K * T, LD), dtype=torch.int32, device=dev)[:, :B]
        for k in range(K):
            w.env.seed(cx.a.seed + k)
            _lib.check(L.bpg_synth_masks_ld(w.env._h, 0, T, C.c_void_p(one_m[k * T].data_ptr()), LD, cx.stream))
        w.env.seed(cx.a.seed)
        one_r = torch.empty((K * T, LD), dtype=torch.float32, device=dev)[:, :B]
        one_f = torch.empty((K * T, LD), dtype=torch.uint8, device=dev)[:, :B]
        own_m = w.shard_streams(K * T, torch.int32, S)
        for t, (lo, n) in zip(own_m, w.shard_ranges(S)):
            t.copy_(one_m[:, lo:lo + n])
        own_r = w.shard_streams(K * T, torch.float32, S)
        own_f = [torch.empty((K * T, int(r.stride(0))), dtype=torch.uint8, device=dev)[:, :r.shape[1]] for r in own_r]
        for label, call in (('worker_own_arrays', lambda: w.evaluate_rollouts_fused(own_m, num_grey=ngk, reward_tb=own_r, flags_tb=own_f)),
                            ('worker_one_launch', lambda: w.evaluate_rollouts_fused(one_m, num_grey=ngk, reward_tb=one_r, flags_tb=one_f))):
            call()
            torch.cuda.synchronize(dev)
            runs = []
            for g in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                call()
                e1.record(cur)
                torch.cuda.synchronize(dev)
                runs.append(e0.elapsed_time(e1) / K)
            out[label] = dict(figure(min(runs)), runs_ms=runs, rollouts_per_call=K,
                              how='RolloutStudent.evaluate_rollouts_fused, %d rollouts per call, HIP events around the call, min of 3 calls' % K)
        del w, one_m, one_r, one_f, own_m, own_r, own_f
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        out['worker_own_arrays'] = {'error': repr(e)[:300]}
    out['how'] = ('%d views x K = %d launches each on %d streams, forked / joined once around them; HIP events on the launch stream; '
                  'min of 3 groups; the one launch under the same events: roofline.kernel_ms' % (S, steps, S))
    out.update({k: out['own_arrays'][k] for k in ('ms_per_rollout', 'achieved', 'frac')})
    out['kernel'] = 'k_rollout<masks, 2 envs/lane x 256 lanes, reward+flags, reset> on %d views / streams, arrays of their own' % S
    out['one_launch_frac'] = nbytes / (h['kernel_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


METRIC = 'env_steps_per_sec'


def error_line(msg, a, **kw):
    """What a failed run prints instead of a result: one parseable JSON line."""
    out = {'metric': METRIC, 'value': None, 'unit': 'env-steps/s', 'n_gpus': a.gpus, 'steps': a.steps,
           'warmup': a.warmup, 'error': msg}
    out.update(kw)
    return out


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environments(a, base_env=None, port=None):
    """The environment of each of the N rank processes (the torchrun contract: RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT), one dict per rank.  Pure host logic (CPU-tested)."""
    base = dict(os.environ if base_env is None else base_env)
    port = port or _free_port()
    envs = []
    for r in range(a.gpus):
        e = dict(base)
        e.update(RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_RANK='0' if a.single_device else str(r),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BPG_BENCH_SPAWNED='1')
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it here
        envs.append(e)
    return envs


def launch_ranks(a, argv):
    """The reference's mpi_fork (gym_blocks/train.py:113-119) for `python bench.py --gpus N`: start N
    fresh rank processes of this file and wait.  The parent makes no HIP call and never imports torch;
    nothing is exec'ed from a process that holds the GPU.  Rank 0's stdout (the JSON line) is relayed;
    the other ranks' stdout goes to stderr.  Exit code: 0 only if every rank exited 0 and rank 0 printed a
    JSON line; otherwise a JSON error line is printed (unless rank 0 already printed one) and the code is 1."""
    envs = rank_environments(a)
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    out0_file = tempfile.TemporaryFile(mode='w+')        # rank 0's stdout (a file: no pipe to fill up)
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(cmd, env=e, stdout=out0_file if r == 0 else sys.stderr, stderr=sys.stderr))
    deadline = time.time() + a.launch_timeout
    rcs = [None] * len(procs)
    failed_at = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
        # a rank that died leaves the others waiting in a collective: end exactly the processes started here
        if (failed_at is not None and time.time() - failed_at > 15) or time.time() > deadline:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.wait()
            break
        if any(rc is None for rc in rcs):
            time.sleep(0.05)
    out0_file.seek(0)
    out0 = out0_file.read()
    line = None
    for ln in out0.splitlines():
        try:
            cand = json.loads(ln)
        except ValueError:
            print(ln, file=sys.stderr)
            continue
        if isinstance(cand, dict) and 'metric' in cand:
            line = cand
        else:
            print(ln, file=sys.stderr)
    ok = all(rc == 0 for rc in rcs) and line is not None and 'error' not in line
    if line is None or (not ok and 'error' not in line):
        line = error_line('rank exit codes %s%s' % (rcs, '' if time.time() <= deadline else ' (launch timeout)'), a,
                          rank_exit_codes=rcs)
    line['launcher'] = {'how': 'bench.py started %d rank processes itself (train.py:113-119 mpi_fork)' % a.gpus,
                        'rank_exit_codes': rcs}
    print(json.dumps(line), flush=True)
    return 0 if ok else 1


_RESULT_FD = None


def emit(line):
    """The ONE JSON line of the run, on the process's original stdout."""
    data = (json.dumps(line) + '\n').encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def main():
    global _RESULT_FD
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    # Libraries print to stdout behind our back (RCCL its version banner, gloo its connection report): while the
    # bench runs, file descriptor 1 points at stderr, and only the result line goes to the real stdout.
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    try:
        run(a)
    except SystemExit:
        raise
    except BaseException as e:          # noqa: BLE001  (a parseable line instead of a bare traceback)
        rank = int(os.environ.get('RANK', '0'))
        traceback.print_exc()
        line = error_line('%s: %s' % (type(e).__name__, str(e)[:500]), a, rank=rank)
        if rank == 0:
            emit(line)
        else:
            print(json.dumps(line), file=sys.stderr, flush=True)
        sys.exit(2)


def collective_schedule(n_it, warmup, R, multi):
    """Which rollouts of a leg are followed by a collective, and which exchange carries each rollout's results.  Pure host
    logic (CPU-tested): a collective follows every R-th rollout, the last warm-up rollout and the last rollout;
    grp[i] = number of collectives before rollout i.  Rollouts of group g write the return buffer g % 2; the gather of group
    g must have finished before group g + 2 starts writing it."""
    R = max(1, R)
    coll_after = [bool(multi) and ((i + 1) % R == 0 or i == n_it - 1 or i == warmup - 1) for i in range(n_it)]
    grp, c = [], 0
    for i in range(n_it):
        grp.append(c)
        c += coll_after[i]
    return coll_after, grp


def rollout_leg(cx, total, B, collective, R, steps, warmup, prewarm, stride_probe=False, impl='torch', gather_dtype='u8'):
    """One timed leg: W warm-up + exactly K timed fused rollouts of this rank's B-env shard of a `total`-env job, bracketed
    by barrier + synchronize.  collective (N > 1 or forced): 'allreduce' = the [successes, envs] rows in slabs of R rollouts
    (16 B per rollout); 'allgather' = every env's episode return all-gathered into global env order once per R rollouts --
    the exchange BASELINE configs[3] names -- as the rollout launch's own per-env byte (bpg_rollout_args.d_episode_fail_count,
    1 B/env; expanded to the float32 return on the device behind the gather) or as the float32 return (4 B/env).
    impl = 'torch': torch.distributed issues the collectives (RCCL on its own stream, asynchronous: the exchange of rollout i
    overlaps the kernel of rollout i+1; a buffer is reused only after its gather is done).  impl = 'cabi': the library issues
    them on the launch stream (bpg_comm_*): stream order is the only synchronisation.
    TIMING.  The K timed steps are bracketed by barrier + synchronize on both sides (wall clock `wall_dt`, as before), and
    INSIDE the bracket a pair of HIP events on the launch stream spans them: the first recorded in front of the first timed
    launch, the second behind the last launch AND behind the completion of the last exchange (the launch stream waits for it
    first -- a stream-level wait).  `dt` = MAX over ranks of that device-side span: the same quantity at N = 1 (no collective,
    no barrier) and at N = 8, where the wall figure additionally holds a dist.barrier() and the host's drain of the collective
    handles (about 150 us per leg: 8 % of a 20-step window)."""
    torch, dist, _lib, D, L, a = cx.torch, cx.dist, cx._lib, cx.D, cx.L, cx.a
    dev, stream, rank, world, multi, T = cx.dev, cx.stream, cx.rank, cx.world, cx.multi, cx.T
    env = D.make_sharded(ENV_ID, total, rank, world, device=dev, seed=a.seed)
    assert env.num_envs == B
    # inputs resident in HBM before the timed region: (T, B) contact masks from the synthetic source (global env index
    # keyed), outputs preallocated.  Variation draws num_grey per episode (fetch_env.py:647); fixed explicit draws keep
    # every rollout's episode consistent with the pre-generated contact masks
    num_grey = ((torch.arange(B, device=dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
    env.reset(num_grey=num_grey)
    # NBUF independent input/output buffer sets, used round-robin, so that no launch finds its contact masks in the
    # 256 MiB Infinity Cache from a previous launch: inputs come from HBM (a set is B * 450 bytes: enough sets that two
    # consecutive uses of one are > 2 x 256 MiB apart)
    NBUF = max(a.buffers, int(2.5 * 256 * 2**20 // (B * T * 9)) + 1)
    # time-major (T, B) streams whose rows are LD = bpg_row_stride(B) elements apart (B + 256 for the power-of-two batch
    # sizes): rows a power of two apart land on the same HBM channels.  The pad is never read or written
    # and is not counted in the bytes.  --contiguous-rows: LD = B.
    LD = B if a.contiguous_rows else int(L.bpg_row_stride(B))

    def tm_empty(dtype):
        return torch.empty((T, LD), dtype=dtype, device=dev)[:, :B]
    masks, reward, flags = [], [], []
    for k in range(NBUF):
        env.seed(a.seed + k)
        m = tm_empty(torch.int32)
        _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), LD, stream))
        masks.append(m)
        reward.append(tm_empty(torch.float32))
        flags.append(tm_empty(torch.uint8))
    env.seed(a.seed)
    gather = multi and collective == 'allgather'
    equal_shards = total % world == 0
    # ONE launch per rollout: reset (rollout.py:79) + T steps + success count (rollout.py:163-167) [+ the per-env episode
    # return].  Rollout i adds its success count straight into row i of `stats` = [successes, envs].
    n_it = warmup + steps
    stats = torch.zeros((n_it, 2), dtype=torch.int64, device=dev)
    stats[:, 1] = B
    R = max(1, R)
    # a collective follows every R-th rollout, the last warm-up rollout and the last rollout; grp[i] = number of
    # collectives before rollout i = the exchange that will carry rollout i's results
    coll_after, grp = collective_schedule(n_it, warmup, R, multi)
    u8 = gather and gather_dtype == 'u8'
    wire = torch.uint8 if u8 else torch.float32
    ret = [torch.zeros(B, dtype=wire, device=dev) for _ in range(2)] if gather else None          # what the launch writes
    gathered = [torch.zeros(total, dtype=wire, device=dev) for _ in range(2)] if gather else None  # what the gather delivers
    # the float32 returns of all envs, global env order: the gathered array itself, or its expansion on the device
    returns_f32 = [torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(2)] if u8 else gathered
    comm = cx.comm() if (multi and impl in ('cabi', 'cabi_side')) else None
    side = torch.cuda.Stream(dev) if (comm is not None and impl == 'cabi_side') else None
    overlapped = impl in ('torch', 'cabi_side')      # the exchange of rollout i runs beside the kernel of rollout i + 1

    class _EventWork:              # what torch's Work is for its collectives: wait() = the current stream waits for the event
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream(dev).wait_event(self.ev)

    def per_env_arg(buf):
        return dict(d_episode_fail_count=buf.data_ptr()) if u8 else dict(d_episode_return=buf.data_ptr())
    args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=num_grey.data_ptr(),
                             d_masks=masks[i % NBUF].data_ptr(), d_reward_tb=reward[i % NBUF].data_ptr(),
                             d_flags_tb=flags[i % NBUF].data_ptr(), ld_masks=LD, ld_out=LD,
                             d_success_count=stats.data_ptr() + 16 * i,
                             **(per_env_arg(ret[grp[i] % 2]) if gather else {})) for i in range(n_it)]
    # ONE pair of HIP events on the launch stream around the K timed launches: the kernel's average duration =
    # elapsed / K.  (An event pair around every launch puts two marker packets between consecutive kernels: +2.4 us per
    # launch on this stream, measured with round 2's tools/tune_geometry.)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev_end = torch.cuda.Event(enable_timing=True)   # behind the last launch AND the last exchange: closes the timed span
    works = []                     # all-reduce handles
    gworks = {}                    # group index -> all-gather handle (torch) whose result has not been consumed yet
    pending = [0]                  # first statistics row that has not been all-reduced yet
    ncoll = [0]

    def expand(g):                 # u8 on the wire: the float32 returns on the device, behind the gather, on the launch stream
        if u8:
            D.episode_return_from_fail_count(gathered[g % 2], out=returns_f32[g % 2])

    def settle(g):
        """The launch stream waits for the gather of group g (a stream-level wait, no host block) and expands its bytes (the
        side-stream form has expanded them there already)."""
        w = gworks.pop(g, None)
        if w is not None:
            w.wait()
            if side is None:
                expand(g)

    def one_rollout(i, k=None):
        group_start = gather and overlapped and (i == 0 or coll_after[i - 1])
        if group_start:
            settle(grp[i] - 2)     # this group's buffers were read by the gather two groups ago (settled already, normally)
        if k == 0:
            ev0.record()
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i]), stream))
        if k == steps - 1:
            ev1.record()
        if group_start:
            settle(grp[i] - 1)     # the previous group's gather ran beside this launch: wait + expansion go BEHIND the launch
        if coll_after[i]:
            ncoll[0] += 1
            if not gather:
                # one collective per R rollouts: the [successes, envs] rows not yet reduced
                lo, pending[0] = pending[0], i + 1
                if comm is not None:
                    comm.all_reduce_sum_(stats[lo:i + 1])      # on the launch stream (16 R bytes: nothing to overlap)
                else:
                    works.append(dist.all_reduce(stats[lo:i + 1], async_op=True))
            elif side is not None:     # the library's ncclAllGather + the expansion on a second stream of the caller's,
                g = grp[i]             # ordered by two events: beside the next rollout's kernel, as torch's collectives run
                after_launch = torch.cuda.Event()
                after_launch.record()
                side.wait_event(after_launch)
                with torch.cuda.stream(side):
                    comm.all_gather(ret[g % 2], out=gathered[g % 2], stream=side)
                    expand(g)
                    done = torch.cuda.Event()
                    done.record(side)
                gworks[g] = _EventWork(done)
            elif comm is not None:     # the library's ncclAllGather on the launch stream: stream order is all there is
                g = grp[i]
                comm.all_gather(ret[g % 2], out=gathered[g % 2])
                expand(g)
            elif equal_shards:     # one all_gather_into_tensor straight into the (total,) result
                g = grp[i]
                gworks[g] = D.all_gather_per_env(ret[g % 2], total_envs=total, out=gathered[g % 2], async_op=True)[1]
            else:                  # ragged shards: padded, still one collective, compacted on the device
                D.all_gather_per_env(ret[grp[i] % 2], total_envs=total, out=gathered[grp[i] % 2])
                expand(grp[i])

    def close_span():
        """Everything the K steps started has completed on the launch stream when ev_end fires."""
        for w in works:
            w.wait()               # stream-level: the launch stream waits for RCCL's stream
        for g in sorted(gworks):
            settle(g)
        ev_end.record()

    def drain():
        while works:
            works.pop().wait()
        for g in sorted(gworks):
            settle(g)

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Row-stride probe (set-up, untimed part of the run; `value` always uses LD above): the headline launch at
    # rows B, B + 256 and B + 768 elements apart on THIS box -- docs/HISTORY.md 3 derives bpg_row_stride from a handful
    # of boxes, this records what the box the bench ran on says.
    probe = None
    if stride_probe and rank == 0 and not a.no_stride_probe and B >= (1 << 19):
        probe = {}
        pads = tuple(int(x) for x in a.stride_pads.split(','))
        probes = {}
        try:
            for pad in pads:
                ldp = B + pad
                # exactly the headline's set-up: NBUF independent mask / reward / flag sets used round-robin
                pm = [torch.empty((T, ldp), dtype=torch.int32, device=dev)[:, :B] for _ in range(NBUF)]
                pr = [torch.empty((T, ldp), dtype=torch.float32, device=dev)[:, :B] for _ in range(NBUF)]
                pf = [torch.empty((T, ldp), dtype=torch.uint8, device=dev)[:, :B] for _ in range(NBUF)]
                for k in range(NBUF):
                    env.seed(a.seed + k)
                    _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(pm[k].data_ptr()), ldp, stream))
                sc = torch.zeros(2, dtype=torch.int64, device=dev)
                probes[pad] = (pm, pr, pf, sc,
                               [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=num_grey.data_ptr(), d_masks=pm[k].data_ptr(),
                                                 d_reward_tb=pr[k].data_ptr(), d_flags_tb=pf[k].data_ptr(), ld_masks=ldp, ld_out=ldp,
                                                 d_success_count=sc.data_ptr()) for k in range(NBUF)])
            env.seed(a.seed)
            best = {pad: [] for pad in pads}
            for rep in range(3):                     # interleaved, so that clock ramp-up hits every stride alike;
                for pad in pads:                     # the first repetition is a warm-up
                    pa = probes[pad][4]
                    for i in range(40):
                        _lib.check(L.bpg_rollout_v2(env._h, C.byref(pa[i % NBUF]), stream))
                    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    p0.record()
                    for i in range(60):
                        _lib.check(L.bpg_rollout_v2(env._h, C.byref(pa[i % NBUF]), stream))
                    p1.record()
                    torch.cuda.synchronize(dev)
                    if rep:
                        best[pad].append(p0.elapsed_time(p1) / 60)
            for pad in pads:
                ms = min(best[pad])
                probe['ld_B_plus_%d' % pad] = {'kernel_ms': ms, 'frac_hbm_peak': B * (T * 9 + 12) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               'runs_ms': best[pad]}
        except Exception as e:  # noqa: BLE001
            probe['error'] = repr(e)[:200]
        probe['library_choice'] = int(L.bpg_row_stride(B)) - B
        probe['how'] = 'three interleaved passes over the pads (first discarded), 60 launches each, HIP events; min of the two kept'
        del probes
        torch.cuda.empty_cache()

    scratch = torch.zeros(2, dtype=torch.int64, device=dev)
    pre = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=num_grey.data_ptr(), d_masks=masks[k].data_ptr(),
                            d_reward_tb=reward[k].data_ptr(), d_flags_tb=flags[k].data_ptr(), ld_masks=LD, ld_out=LD,
                            d_success_count=scratch.data_ptr(), **(per_env_arg(ret[0]) if gather else {})) for k in range(NBUF)]
    if prewarm > 0:                        # set-up, not measurement: same launch, scratch statistics row
        for i in range(prewarm):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(pre[i % NBUF]), stream))
        torch.cuda.synchronize(dev)
    for i in range(warmup):
        one_rollout(i)
    drain()
    fence()
    t0 = time.perf_counter()
    for k in range(steps):
        one_rollout(warmup + k, k)
    close_span()
    host_dt = time.perf_counter() - t0          # time to ENQUEUE the K steps (host side only)
    drain()
    fence()
    wall_dt = time.perf_counter() - t0
    span_dt = ev0.elapsed_time(ev_end) * 1e-3    # device-side span of the K steps incl. the completion of the last exchange
    rank_span_ms, rank_wall_ms = [span_dt * 1e3], [wall_dt * 1e3]
    dt = span_dt
    if multi:
        both = torch.tensor([span_dt, wall_dt], dtype=torch.float64, device=dev)
        allb = [torch.zeros_like(both) for _ in range(dist.get_world_size())]
        dist.all_gather(allb, both)
        rank_span_ms = [float(b[0].item()) * 1e3 for b in allb]
        rank_wall_ms = [float(b[1].item()) * 1e3 for b in allb]
        dt = max(rank_span_ms) * 1e-3
        wall_dt = max(rank_wall_ms) * 1e-3
    kernel_ms = ev0.elapsed_time(ev1) / steps
    # Spread (untimed, after the measurement): three more groups of K launches of the same kernel on the same buffers,
    # each between its own event pair, no collectives -- how much the launch duration moves on this box, and at N > 1
    # the launch duration without the waits for the collective that the timed region's events include
    groups = []
    for g in range(3):
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for i in range(steps):
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(pre[(g * steps + i) % NBUF]), stream))
        g1.record()
        torch.cuda.synchronize(dev)
        groups.append(g0.elapsed_time(g1) / steps)
    spread = {'min': min(groups + [kernel_ms]), 'max': max(groups + [kernel_ms]), 'timed_region': kernel_ms, 'later_groups': groups,
              'how': 'the K timed launches + 3 untimed groups of K launches after them, one HIP event pair per group'}
    rank_kernel_ms = [kernel_ms]
    if multi:                                   # every rank's own HIP-event figure (the headline uses rank 0's)
        km = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        allk = [torch.zeros_like(km) for _ in range(dist.get_world_size())]
        dist.all_gather(allk, km)
        rank_kernel_ms = [float(k.item()) for k in allk]
    last = stats[-1].clone()
    if gather:                                  # untimed: the global success rate of the last rollout
        dist.all_reduce(last)
    last = last.tolist()
    succ_rate = last[0] / last[1]               # global: the row was all-reduced when world > 1
    checksum = None
    if gather:
        # the gathered (total,) returns of the LAST rollout, reduced to two integers any single-batch run can reproduce:
        # failed steps summed over all envs, and the same weighted by (global env index mod 65521) + 1
        gl = returns_f32[grp[n_it - 1] % 2]
        nf = (-gl).to(torch.int64)
        wts = (torch.arange(total, device=dev, dtype=torch.int64) % 65521) + 1
        checksum = {'failed_steps': int(nf.sum().item()), 'weighted': int((nf * wts).sum().item()),
                    'all_negative_zero_or_below': bool(torch.signbit(gl).all().item()),
                    'mask_buffer_set': (n_it - 1) % NBUF, 'seed': a.seed + (n_it - 1) % NBUF,
                    'on_the_wire': 'u8 fail counts, expanded on the device' if u8 else 'f32 returns'}
        if u8:      # the bytes that travelled ARE the returns: -(float)count, bit for bit
            checksum['expansion_matches_bytes'] = bool(torch.equal(gl, -(gathered[grp[n_it - 1] % 2].to(torch.float32))))
    return dict(env=env, masks=masks, reward=reward, flags=flags, B=B, total=total, LD=LD, NBUF=NBUF, R=R, dt=dt, wall_dt=wall_dt,
                rank_span_ms=rank_span_ms, rank_wall_ms=rank_wall_ms, impl=impl if multi else None,
                wire_bytes_per_env=(1 if u8 else 4) if gather else None,
                host_dt=host_dt, kernel_ms=kernel_ms, rank_kernel_ms=rank_kernel_ms, spread=spread, succ_rate=succ_rate,
                checksum=checksum, probe=probe, collectives=ncoll[0], gather=gather, steps=steps, warmup=warmup)


def run(a):
    import types
    import torch
    import torch.distributed as dist
    import blockpuzzle_gym_amd as bpg
    from blockpuzzle_gym_amd import _lib, distributed as D

    if a.single_device:
        os.environ['LOCAL_RANK'] = '0'
    rank, world, local = D.init_from_env(backend=a.backend)
    force_dist = os.environ.get('BPG_BENCH_FORCE_DIST') == '1'     # rehearsal: RCCL path on one rank
    if force_dist and world == 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        torch.cuda.set_device(local)
        dist.init_process_group(a.backend or 'nccl', rank=0, world_size=1, device_id=torch.device('cuda', local))
    if world != a.gpus:
        raise RuntimeError('--gpus %d but WORLD_SIZE=%d: launch as `python bench.py --gpus N` (starts the ranks '
                           'itself) or under torchrun --nproc-per-node N' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs a GPU (no CPU fallback)')
    multi = world > 1 or force_dist          # take the collective code path
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    L = _lib.lib()
    T = a.T
    if a.total_envs:
        total = a.total_envs
        lo, hi = D.shard_range(total, rank, world)
        B = hi - lo
    else:
        B = a.envs
        total = B * world
    if not a.default_stream:
        # a dedicated stream: the legacy default stream serialises against the RCCL stream
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cx = types.SimpleNamespace(torch=torch, dist=dist, _lib=_lib, D=D, L=L, a=a, dev=dev, stream=stream, rank=rank,
                               world=world, multi=multi, T=T, _comm=None)

    def comm():         # the library's own RCCL communicator (bpg_comm_*), created on first use: a collective over the ranks
        if cx._comm is None:
            cx._comm = D.RcclComm.from_process_group(dev)
        return cx._comm
    cx.comm = comm
    # which device every rank really runs on (LOCAL_RANK -> cuda index, the visible-device lists, the device's identity)
    props = torch.cuda.get_device_properties(dev)
    mine = {'rank': rank, 'local_rank': local, 'device': 'cuda:%d' % dev.index, 'name': props.name,
            'uuid': str(getattr(props, 'uuid', '')), 'pci_bus_id': getattr(props, 'pci_bus_id', None),
            'HIP_VISIBLE_DEVICES': os.environ.get('HIP_VISIBLE_DEVICES'), 'ROCR_VISIBLE_DEVICES': os.environ.get('ROCR_VISIBLE_DEVICES'),
            'CUDA_VISIBLE_DEVICES': os.environ.get('CUDA_VISIBLE_DEVICES'), 'visible_device_count': torch.cuda.device_count()}
    rank_devices = [mine]
    if dist.is_initialized() and dist.get_world_size() > 1:
        rank_devices = [None] * dist.get_world_size()
        dist.all_gather_object(rank_devices, mine)

    h = rollout_leg(cx, total, B, a.collective, a.reduce_every, a.steps, a.warmup, a.prewarm, stride_probe=True,
                    impl=a.collective_impl, gather_dtype=a.gather_dtype)
    dt, kernel_ms, LD, NBUF, R = h['dt'], h['kernel_ms'], h['LD'], h['NBUF'], h['R']

    # algorithmic bytes of the fused rollout launch: per env-step mask 4 B in,
    # reward 4 B + flags 1 B out; per env the packed state 8 B out (the in-launch Variation reset
    # overwrites the whole state, so the old state is not an input) + the explicit num_grey draw 4 B in
    # (+ the episode return 4 B out when the leg all-gathers it)
    bytes_per_launch = B * (T * 9 + 8 + 4 + (h['wire_bytes_per_env'] or 0))
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tf = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tf):
        t = json.load(open(tf)).get('k_rollout_masks')
        if t and t.get('envs') == B and t.get('T') == T and not h['gather']:
            traffic = t['hbm_bytes_per_launch']
            traffic_source = ('committed constant, NOT measured by this run: profiles/traffic.json <- %s (rocprofv3 --pmc '
                              'FETCH_SIZE / WRITE_SIZE passes of the same launch, corrected as MI355X_MICROARCH.md prescribes; '
                              'counters cannot be collected from inside the bench)' % t.get('source', 'tools/collect_profiles.sh'))

    # rocprofv3 --kernel-trace --stats means of the committed profile of this same command
    # (tools/collect_profiles.sh -> profiles/r0N_pmc/kernel_means.json), quoted beside the live HIP-event figures
    prof_means, mf = {}, None
    for rnd in ('r05', 'r04', 'r03', 'r02'):
        mf = os.path.join(ROOT, 'profiles', rnd + '_pmc', 'kernel_means.json')
        if os.path.exists(mf):
            prof_means = json.load(open(mf))
            break

    def rocprof_of(label, nbytes):
        m = prof_means.get(label)
        if not m:
            return None
        return {'mean_us': m['mean_us'], 'calls': m['calls'],
                'frac_hbm_peak': nbytes / (m['mean_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 'source': os.path.relpath(mf, ROOT)}

    coll = (a.collective if multi else None)
    out = {
        'metric': 'env_steps_per_sec', 'value': total * T * a.steps / dt, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
        'host_enqueue_ms_per_step': h['host_dt'] / a.steps * 1e3,
        # how `value` is timed: see rollout_leg.  The wall figure (perf_counter from in front of the first enqueue to behind the
        # closing synchronize / barrier / synchronize) is what rounds 1-4 reported as ms_per_step.
        'timing': {'value_uses': 'MAX over ranks of the HIP-event span on the launch stream: first timed launch -> behind the last '
                                 'launch and the completion of the last exchange; inside the barrier + synchronize bracket',
                   'span_ms_per_rank': h['rank_span_ms'], 'wall_ms_per_rank': h['rank_wall_ms'],
                   'wall_ms_per_step_incl_sync_and_barrier': h['wall_dt'] / a.steps * 1e3,
                   'value_from_wall_clock': total * T * a.steps / h['wall_dt']},
        'higher_is_better': True, 'scaling': 'strong' if a.total_envs else 'weak', 'vs_baseline': None, 'dtype': 'u32',
        'data': 'synthetic',
        'config': {'workload': '%s, %d envs per GPU x T=%d fused rollout per step, contact masks (T,B) u32 '
                               '(rows %d elements apart) resident in HBM, packed now/ever state, reward f32 + flags u8 written per '
                               'env-step%s' % (ENV_ID, B, T, LD, (', episode %s per env' % ('fail count u8' if h['wire_bytes_per_env'] == 1 else 'return f32'))
                                               if h['gather'] else ''),
                   'envs_per_gpu': B, 'total_envs': total, 'T': T, 'row_stride_elements': LD,
                   'parallelism': 'env-shard x%d' % world,
                   'buffer_sets': NBUF, 'collective': coll,
                   'rollouts_per_collective': (R if multi else None),
                   'success_rate': h['succ_rate']},
        # what the collective path actually ran with (observed, not requested)
        'rccl_world_size': (dist.get_world_size() if dist.is_initialized() else 1),
        'backend': (dist.get_backend() if dist.is_initialized() else None),
        'collective': coll,
        'collective_impl': h['impl'],
        'rank_devices': rank_devices,
        'collectives_issued': (h['collectives'] if multi else None),
        'collective_bytes_per_rollout': (None if not multi else (h['wire_bytes_per_env'] * total if h['gather'] else 16)),
        'gathered_bytes_per_rollout': (h['wire_bytes_per_env'] * total if h['gather'] else None),
        'gathered_return_checksum': h['checksum'],
        'kernel_ms_per_rank': {'min': min(h['rank_kernel_ms']), 'max': max(h['rank_kernel_ms']), 'all': h['rank_kernel_ms']},
        'kernel_ms_spread': h['spread'],
        'roofline': {'bound': 'hbm', 'kernel': 'k_rollout<masks, %s, reward+flags, reset>' % ('2 envs/lane x 1024 lanes' if B >= 524288 else ('2 envs/lane x 256 lanes' if B >= 229376 else '1 env/lane x 256 lanes, pipelined')), 'achieved': achieved,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_source': traffic_source, 'bytes_per_launch': bytes_per_launch,
                     'bytes_per_env_step': bytes_per_launch / (B * T), 'kernel_ms': kernel_ms,
                     'kernel_ms_spread': {'min': h['spread']['min'], 'max': h['spread']['max']},
                     'rocprofv3_committed': rocprof_of('k_rollout_masks_bench', bytes_per_launch) if B == (1 << 20) and T == 50 and not h['gather'] else None},
    }
    # the job ran as described: one rank per GPU, every rank on a device of its own, every rank's figure present
    if out['rccl_world_size'] != world or len(out['kernel_ms_per_rank']['all']) != world or len(rank_devices) != world:
        raise RuntimeError('the process group holds %d ranks, %d kernel figures, %d device records for --gpus %d'
                           % (out['rccl_world_size'], len(out['kernel_ms_per_rank']['all']), len(rank_devices), world))
    if world > 1 and not a.single_device:
        ids = [(d['uuid'] or d['pci_bus_id'] or d['device']) for d in rank_devices]
        if len(set(ids)) != world:
            raise RuntimeError('ranks share a GPU: %s' % ids)
    if h['probe'] is not None:
        out['row_stride_probe'] = h['probe']
    # (not under a profiler: its preloaded library would ride along into the child process)
    profiled = any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'))
    if rank == 0 and world == 1 and not a.no_box_state and not profiled:
        out['box_state_under_load'] = box_state_under_load(cx, h)

    if rank == 0 and world == 1 and not a.no_sharded:
        try:
            out['roofline_sharded'] = sharded_roofline(cx, h, a.steps, a.warmup)
        except Exception as e:  # noqa: BLE001
            traceback.print_exc()
            out['roofline_sharded'] = {'error': repr(e)[:300]}
    if rank == 0 and world == 1 and not a.no_extras:
        out['extra'] = extras(torch, bpg, _lib, h['env'], h['masks'][0], h['reward'][0], h['flags'][0], B, T, dev, stream)
        # SURVEY 8(d)'s per-step accounting (25 B/env-step: state read+written every step) for the
        # one-launch-per-step form of the same path, next to the fused headline above
        one = out['extra'].get('step_masks_single_launch_hbm_cold', {})
        grp = out['extra'].get('step_masks_graph_T_launches', {})
        if 'GB/s' in one:
            out['roofline_per_step_launch'] = {
                'bound': 'hbm', 'kernel': 'k_step_vec4<masks>', 'bytes_per_env_step': 25,
                'achieved': one['GB/s'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': one['GB/s'] / HBM_PEAK_GBS, 'kernel_ms': one['ms'],
                'how': '13 independent 1 M-env batches stepped round-robin (HBM-cold), events around a train of launches',
                'rocprofv3_committed': rocprof_of('k_step_vec4_masks', 25 * B) if B == (1 << 20) else None,
                'graph_replay_of_T_launches': {k: grp.get(k) for k in ('ms', 'GB/s', 'frac_hbm_peak',
                                                                      'env_steps_per_sec')}}

    # BASELINE configs[3] as written, as a second timed leg of the same run whenever the job has a collective path
    # (N > 1, or the one-rank RCCL rehearsal): 4 194 304 envs in total sharded over the ranks, every env's episode return
    # all-gathered once per rollout.  The headline above stays the weak-scaling figure the driver's scaling curve needs.
    def config3_record(c, note):
        cb = c['B'] * (T * 9 + 12 + c['wire_bytes_per_env'])
        return {
            'workload': '%s, %d envs in total = %d per GPU x %d GPUs, T=%d fused rollout per step, every env\'s episode %s '
                        '(written by the rollout launch) all-gathered into global env order once per rollout%s'
                        % (ENV_ID, c['total'], c['B'], world, T, 'fail count (u8)' if c['wire_bytes_per_env'] == 1 else 'return (f32)',
                           ', expanded to the f32 returns on the device' if c['wire_bytes_per_env'] == 1 else ''),
            'collectives_issued_by': note,
            'value': c['total'] * T * c['steps'] / c['dt'], 'unit': 'env-steps/s', 'scaling': 'strong',
            'steps': c['steps'], 'warmup': c['warmup'], 'ms_per_step': c['dt'] / c['steps'] * 1e3,
            'wall_ms_per_step_incl_sync_and_barrier': c['wall_dt'] / c['steps'] * 1e3,
            'span_ms_per_rank': c['rank_span_ms'],
            'host_enqueue_ms_per_step': c['host_dt'] / c['steps'] * 1e3,
            'total_envs': c['total'], 'envs_per_gpu': c['B'], 'row_stride_elements': c['LD'], 'buffer_sets': c['NBUF'],
            'rccl_world_size': dist.get_world_size(), 'backend': dist.get_backend(),
            'collective': 'allgather', 'rollouts_per_collective': 1, 'collectives_issued': c['collectives'],
            'sent_bytes_per_rank_per_rollout': c['wire_bytes_per_env'] * c['B'],
            'gathered_bytes_per_rollout': c['wire_bytes_per_env'] * c['total'],
            'gathered_return_checksum': c['checksum'], 'success_rate': c['succ_rate'],
            'stream_ms_per_rollout_incl_collective_waits': c['kernel_ms'],
            'kernel_ms_no_collective': c['spread']['later_groups'],
            'kernel_ms_per_rank': {'min': min(c['rank_kernel_ms']), 'max': max(c['rank_kernel_ms']), 'all': c['rank_kernel_ms']},
            'roofline_kernel_only': {'bound': 'hbm', 'bytes_per_launch': cb, 'kernel_ms': min(c['spread']['later_groups']),
                                     'frac': cb / (min(c['spread']['later_groups']) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }

    if multi and not a.no_config3 and not (h['gather'] and total == a.config3_total and R == 1):
        del h
        torch.cuda.empty_cache()
        c_lo, c_hi = D.shard_range(a.config3_total, rank, world)
        c = rollout_leg(cx, a.config3_total, c_hi - c_lo, 'allgather', 1, a.steps, a.warmup, min(a.prewarm, 50),
                        impl=a.collective_impl, gather_dtype=a.gather_dtype)
        out.setdefault('extra', {})['config3_as_written'] = config3_record(
            c, {'torch': 'torch.distributed (RCCL on its own stream, asynchronous)', 'cabi': 'libbpgym.so (ncclAllGather on the launch stream)',
                'cabi_side': 'libbpgym.so (ncclAllGather + expansion on a second stream, event-ordered)'}[a.collective_impl])
        del c
        # The same leg with the collectives issued by the OTHER implementation, for the comparison on hardware.  The library's
        # communicator is created here for the first time (ncclCommInitRank over the ranks): should that never return, a
        # watchdog prints the result line without this leg and ends the process -- on every rank alike.
        if not a.no_cabi_leg and a.collective_impl == 'torch' and dist.get_backend() == 'nccl' and not a.single_device:
            import threading
            done = threading.Event()

            def give_up():
                if not done.wait(a.cabi_leg_timeout):
                    out['extra']['config3_cabi_collectives'] = {'error': 'no result within %.0f s (given up by the watchdog)' % a.cabi_leg_timeout}
                    if rank == 0:
                        emit(out)
                    os._exit(0)        # every rank alike: the launcher must not read a lost extra leg as a failed run
            threading.Thread(target=give_up, daemon=True).start()
            try:
                torch.cuda.empty_cache()
                c = rollout_leg(cx, a.config3_total, c_hi - c_lo, 'allgather', 1, a.steps, a.warmup, min(a.prewarm, 50),
                                impl='cabi', gather_dtype=a.gather_dtype)
                out['extra']['config3_cabi_collectives'] = config3_record(c, 'libbpgym.so (ncclAllGather on the launch stream)')
                del c
                torch.cuda.empty_cache()
                c = rollout_leg(cx, a.config3_total, c_hi - c_lo, 'allgather', 1, a.steps, a.warmup, min(a.prewarm, 50),
                                impl='cabi_side', gather_dtype=a.gather_dtype)
                out['extra']['config3_cabi_collectives_side_stream'] = config3_record(
                    c, 'libbpgym.so (ncclAllGather + expansion on a second stream, event-ordered: beside the next launch)')
                del c
            except Exception as e:  # noqa: BLE001
                out['extra']['config3_cabi_collectives'] = {'error': repr(e)[:300]}
            done.set()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(T, a.cpu_seconds)
    if multi:
        dist.barrier()
    if rank == 0:
        emit(out)
    if cx._comm is not None:             # the library's communicator goes before the process group that carried its id
        cx._comm.close()
    if multi:
        dist.destroy_process_group()


def extras(torch, bpg, _lib, env, masks, reward, flags, B, T, dev, stream):
    """Secondary kernels of the same path, each with its own algorithmic-byte
    roofline line (not part of `value`)."""
    L = _lib.lib()
    res = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    def line(ms, nbytes, env_steps, cache_resident=False):
        gbs = nbytes / (ms * 1e-3) / 1e9
        out = {'ms': ms, 'GB/s': gbs, 'bytes': nbytes, 'env_steps_per_sec': env_steps / (ms * 1e-3)}
        if cache_resident:      # working set stays in the 256 MiB Infinity Cache: NOT an HBM figure
            out['infinity_cache_resident'] = True
        else:
            out['frac_hbm_peak'] = gbs / HBM_PEAK_GBS
        return out

    # launch floor of this method (events around a train of back-to-back launches): the library's
    # smallest kernel, a reset of a 1-env batch
    tiny = bpg.make(ENV_ID, num_envs=1, device=dev)
    ng1 = torch.zeros(1, dtype=torch.int32, device=dev)       # explicit draw: the reset is ONE launch
    res['launch_floor_smallest_kernel'] = {
        'ms': timed(lambda: _lib.check(L.bpg_reset(tiny._h, C.c_void_p(ng1.data_ptr()), None, stream)), 200)}
    del tiny

    # (0) on-box streaming ceiling: device-to-device copy of one (T,B) u32 array (read + write)
    src, dst = masks, reward.view(torch.int32)
    src, dst = src.contiguous(), torch.empty((T, B), dtype=torch.int32, device=dev)
    res['copy_ceiling_torch_d2d'] = line(timed(lambda: dst.copy_(src), 20), 2 * masks.numel() * 4, 0)
    del src, dst

    # (0a) the headline kernel at BASELINE.json's other batch sizes (configs[1]: 65 536 envs;
    # configs[3]: 4 194 304 over 8 GPUs = 524 288 per GPU), masks read cold from HBM
    for Bx in (65536, 524288):
        try:
            res['rollout_masks_%d_envs' % Bx] = rollout_at(torch, bpg, _lib, Bx, T, dev, stream, timed, line)
        except Exception as e:  # noqa: BLE001
            res['rollout_masks_%d_envs' % Bx] = {'error': repr(e)[:200]}
    # ... configs[3]'s per-GPU shard with the per-env episode return the 8 GPUs all-gather (4 B/env more), and the headline
    # batch with it
    for Bx in (524288, B, 4 * B):                     # 4 * B = configs[3]'s whole job on ONE GPU at the default batch
        try:
            res['rollout_masks_%d_envs_with_episode_return' % Bx] = rollout_at(torch, bpg, _lib, Bx, T, dev, stream, timed, line,
                                                                               episode_return=True)
        except Exception as e:  # noqa: BLE001
            res['rollout_masks_%d_envs_with_episode_return' % Bx] = {'error': repr(e)[:200]}
    # ... and the evaluator's n_test_rollouts = 10 (config.py:43) rollouts of configs[1] in ONE launch
    try:
        res['rollout_masks_65536_envs_10_rollouts_per_launch'] = rollout_at(torch, bpg, _lib, 65536, T, dev, stream, timed,
                                                                            line, rollouts=10)
    except Exception as e:  # noqa: BLE001
        res['rollout_masks_65536_envs_10_rollouts_per_launch'] = {'error': repr(e)[:200]}
    # ... and of the headline batch: what the kernel streams at once the fixed cost of a launch is spread over 10 rollouts
    try:
        res['rollout_masks_%d_envs_10_rollouts_per_launch' % B] = rollout_at(torch, bpg, _lib, B, T, dev, stream, timed, line,
                                                                            rollouts=10)
    except Exception as e:  # noqa: BLE001
        res['rollout_masks_%d_envs_10_rollouts_per_launch' % B] = {'error': repr(e)[:200]}

    # (1) one env.step launch: 12 B in (now, ever, mask) + 13 B out (now, ever, reward, flags) = 25 B/env-step
    # (SURVEY 8d's per-step form).  One batch stepped over and over keeps its 26 MB in the Infinity Cache ...
    def step_once():
        _lib.check(L.bpg_step_masks(env._h, C.c_void_p(masks.data_ptr()), C.c_void_p(reward.data_ptr()),
                                    C.c_void_p(flags.data_ptr()), stream))
    res['step_masks_single_launch'] = line(timed(step_once, 200), 25 * B, B, cache_resident=True)
    # ... so the HBM figure steps 13 independent batches round-robin (13 x 26 MB of state + streams
    # between two visits of the same batch: nothing survives in the 256 MiB cache)
    try:
        NSET = 13
        sets = []
        for k in range(NSET):
            e = bpg.make(ENV_ID, num_envs=B, device=dev, seed=100 + k)
            e.reset()
            m = torch.empty(B, dtype=torch.int32, device=dev)
            _lib.check(L.bpg_synth_masks(e._h, 0, 1, C.c_void_p(m.data_ptr()), stream))
            sets.append((e, m, torch.empty(B, dtype=torch.float32, device=dev),
                         torch.empty(B, dtype=torch.uint8, device=dev)))
        it = [0]

        def step_cold():
            e, m, r, f = sets[it[0] % NSET]
            it[0] += 1
            _lib.check(L.bpg_step_masks(e._h, C.c_void_p(m.data_ptr()), C.c_void_p(r.data_ptr()),
                                        C.c_void_p(f.data_ptr()), stream))
        res['step_masks_single_launch_hbm_cold'] = line(timed(step_cold, 20 * NSET), 25 * B, B)
        res['step_masks_single_launch_hbm_cold']['batches_round_robin'] = NSET
        del sets
    except Exception as e:  # noqa: BLE001
        res['step_masks_single_launch_hbm_cold'] = {'error': repr(e)[:200]}

    # (1a) the rollout worker's step: step + float32 achieved_goal row + float32 is_success in one launch:
    # 12 B in, 13 + 4 + 144 B out = 173 B/env-step
    # The worker writes the rows of step t into time slot t+1 of the episode's (T+1, B, dimg) slab: a different 151 MB
    # every step.  (Rounds 1-3 timed this launch rewriting ONE (B, dimg) array, whose lines then never leave the
    # Infinity Cache; with the nontemporal stores of round 4 the rows go to HBM either way.)
    agrec = torch.empty((T + 1, B, env.dimg), dtype=torch.float32, device=dev)
    succ_f = torch.empty((T, B), dtype=torch.float32, device=dev)
    sr_it = [0]

    def step_record():
        t = sr_it[0] % T
        sr_it[0] += 1
        sa = _lib.StepArgs(d_masks=masks[t].data_ptr(), d_reward=reward[t].data_ptr(), d_flags=flags[t].data_ptr(),
                           d_success_f32=succ_f[t].data_ptr(), d_ag=agrec[t + 1].data_ptr(), ld_ag=env.dimg)
        _lib.check(L.bpg_step_record_v2(env._h, C.byref(sa), stream))
    res['step_record_single_launch'] = line(timed(step_record, 2 * T), (29 + 4 * env.dimg) * B, B)
    res['step_record_single_launch']['note'] = 'rows into rotating time slots of a (T+1, B, dimg) slab, as the worker writes them'
    del agrec, succ_f

    # (1b) one env.step launch with the contacts as the reference reads them: geom-id pair lists
    # (B, maxcon=16, 2) int16 + ncon; bytes = 8 state + 4 ncon + 64 list in, 13 out = 89 B/env-step; eight list sets in
    # rotation, so the lists come from HBM (the 8 MB of state stay cache-resident)
    try:
        maxcon = 16
        names = bpg.default_geom_names(ENV_ID)[2]
        objs = [i for i, n in enumerate(names) if n and (n == 'table' or n.startswith('object') or 'finger' in n)]
        gsel = torch.tensor(objs, dtype=torch.int16, device=dev)
        csets = [(gsel[torch.randint(0, len(objs), (B, maxcon, 2), device=dev)].contiguous(),
                  torch.randint(0, 9, (B,), dtype=torch.int32, device=dev)) for _ in range(8)]
        c_it = [0]

        def step_contacts():
            pairs, ncon = csets[c_it[0] % len(csets)]
            c_it[0] += 1
            _lib.check(L.bpg_step_contacts(env._h, C.c_void_p(pairs.data_ptr()), C.c_void_p(ncon.data_ptr()),
                                           maxcon, C.c_void_p(reward.data_ptr()), C.c_void_p(flags.data_ptr()),
                                           stream))
        res['step_contacts_single_launch'] = line(timed(step_contacts, 96), 89 * B, B)
        del csets
    except Exception as e:  # noqa: BLE001
        res['step_contacts_single_launch'] = {'error': repr(e)[:200]}

    # (1c) the contact lists of a whole recorded rollout -> the fused rollout's mask stream (bpg_contacts_to_masks_tb):
    # (T, n, 16, 2) int16 + (T, n) int32 in, (T, LD) masks out; 262 144 envs = 0.94 GB of lists per set, two sets
    try:
        nrec, maxcon = 262144, 16
        erec = bpg.make(ENV_ID, num_envs=nrec, device=dev, seed=7)
        ngr = (torch.arange(nrec, device=dev, dtype=torch.int32) * 7) % 3
        tb = [(gsel[torch.randint(0, len(objs), (T, nrec, maxcon, 2), device=dev)].contiguous(),
               torch.randint(0, 9, (T, nrec), dtype=torch.int32, device=dev), bpg.time_major_empty(T, nrec, torch.int32, dev))
              for _ in range(2)]
        t_it = [0]

        def lists_to_masks():
            p_, c_, m_ = tb[t_it[0] % 2]
            t_it[0] += 1
            _lib.check(L.bpg_contacts_to_masks_tb(erec._h, C.c_void_p(p_.data_ptr()), C.c_void_p(c_.data_ptr()), T, maxcon,
                                                  C.c_void_p(ngr.data_ptr()), C.c_void_p(m_.data_ptr()), int(m_.stride(0)), stream))
        res['contacts_to_masks_recorded_rollout_262144_envs'] = line(timed(lists_to_masks, 8), T * nrec * (4 * maxcon + 8) + 4 * nrec, nrec * T)
        del tb, erec
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        res['contacts_to_masks_recorded_rollout_262144_envs'] = {'error': repr(e)[:200]}

    # (2) the same T steps as T launches replayed from a hipGraph (25 B/env-step, streams from HBM,
    # the 8 MB of state cache-resident between the launches)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            cs = C.c_void_p(s.cuda_stream)
            with torch.cuda.graph(g, stream=s):
                for t in range(T):
                    _lib.check(L.bpg_step_masks(env._h, C.c_void_p(masks[t].data_ptr()),
                                                C.c_void_p(reward[t].data_ptr()),
                                                C.c_void_p(flags[t].data_ptr()), cs))
        res['step_masks_graph_T_launches'] = line(timed(g.replay, 20), 25 * B * T, B * T)
    except Exception as e:  # noqa: BLE001
        res['step_masks_graph_T_launches'] = {'error': repr(e)[:200]}

    # (3) fused rollout with in-kernel synthetic contacts (SURVEY 8d's "contacts generated in-kernel"):
    # reward 4 + flags 1 written per env-step + 8 B of state per env (the in-launch reset reads nothing)
    LD = int(masks.stride(0))             # the three streams share the row stride
    assert reward.stride(0) == LD and flags.stride(0) == LD
    a2 = _lib.RolloutArgs(T=T, reset_first=1, d_masks=None, d_reward_tb=reward.data_ptr(),
                          d_flags_tb=flags.data_ptr(), ld_out=LD)

    def synth():
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a2), stream))
    res['rollout_synth_in_kernel'] = line(timed(synth, 20), B * (T * 5 + 8), B * T)

    # (3b) compact outputs: flags alone (bit2 encodes the reward) -> mask 4 in + flags 1 out per env-step
    a2c = _lib.RolloutArgs(T=T, reset_first=1, d_masks=masks.data_ptr(), d_flags_tb=flags.data_ptr(), ld_masks=LD, ld_out=LD)

    def compact():
        _lib.check(L.bpg_rollout_v2(env._h, C.byref(a2c), stream))
    res['rollout_masks_flags_only'] = line(timed(compact, 20), B * (T * 5 + 8), B * T)

    # (3c) float view of the packed state: obs['achieved_goal'] (B, 36) f32
    agv = torch.empty((B, env.dimg), dtype=torch.float32, device=dev)

    def get_ag():
        _lib.check(L.bpg_get_achieved_goal_f32(env._h, C.c_void_p(agv.data_ptr()), env.dimg, stream))
    res['get_achieved_goal_f32'] = line(timed(get_ag, 50), B * (8 + 4 * env.dimg), B)

    # (3d) HER-style batched compute_reward on float rows (4 B rows of 36): 288 B in + 4 B out per row
    nrow = 4 * B
    cr_ag = torch.randint(-1, 2, (nrow, env.dimg), device=dev).to(torch.float32)
    cr_g = env._g[:1].expand(nrow, env.dimg).contiguous()
    cr_out = torch.empty(nrow, dtype=torch.float32, device=dev)

    def cr():
        _lib.check(L.bpg_compute_reward_f32(C.c_void_p(cr_ag.data_ptr()), C.c_void_p(cr_g.data_ptr()), nrow,
                                            env.dimg, C.c_void_p(cr_out.data_ptr()), dev.index, stream))
    res['compute_reward_f32_rows'] = line(timed(cr, 20), nrow * (8 * env.dimg + 4), nrow)

    def cr_goal():      # env.compute_reward(ag, env.goal): one goal row for all rows (fetch_env.py:135-143 with broadcasting)
        _lib.check(L.bpg_compute_reward_f32_goal(C.c_void_p(cr_ag.data_ptr()), C.c_void_p(cr_g.data_ptr()), nrow,
                                                 env.dimg, C.c_void_p(cr_out.data_ptr()), dev.index, stream))
    res['compute_reward_f32_one_goal_row'] = line(timed(cr_goal, 20), nrow * (4 * env.dimg + 4), nrow)
    del cr_ag, cr_g
    # float64 rows: the precision of env.step's own call (a float64 matrix against an int64 goal, robot_env.py:66): 2 B rows
    nrow = 2 * B
    cr_ag = torch.randint(-1, 2, (nrow, env.dimg), device=dev).to(torch.float64)
    cr_g = env._g[:1].to(torch.float64).expand(nrow, env.dimg).contiguous()

    def cr64():
        _lib.check(L.bpg_compute_reward_f64(C.c_void_p(cr_ag.data_ptr()), C.c_void_p(cr_g.data_ptr()), nrow,
                                            env.dimg, C.c_void_p(cr_out.data_ptr()), dev.index, stream))

    def cr64_goal():
        _lib.check(L.bpg_compute_reward_f64_goal(C.c_void_p(cr_ag.data_ptr()), C.c_void_p(cr_g.data_ptr()), nrow,
                                                 env.dimg, C.c_void_p(cr_out.data_ptr()), dev.index, stream))
    res['compute_reward_f64_rows'] = line(timed(cr64, 20), nrow * (16 * env.dimg + 4), nrow)
    res['compute_reward_f64_one_goal_row'] = line(timed(cr64_goal, 20), nrow * (8 * env.dimg + 4), nrow)
    del cr_ag, cr_g, cr_out

    # (3d') the expert's inputs of one step (policy_gradient/rollout.py:105-171, :198): 87 -> 40 (float64) observation
    # columns, 36 -> 16 goal entries twice; 636 B read + 448 B written per env
    try:
        tr_o = torch.zeros((B, 87), dtype=torch.float32, device=dev)
        tr_o[:, 11 + 15 + 2] = 1.0                      # slot 0 green
        tr_o[:, 11 + 19 + 15 + 3] = 1.0                 # slot 1 blue
        tr_g = torch.randint(-1, 2, (B, 36), device=dev).to(torch.float32)
        tr_ag = torch.randint(-1, 2, (B, 36), device=dev).to(torch.float32)
        tr_out = (torch.empty((B, 40), dtype=torch.float64, device=dev), torch.empty((B, 16), dtype=torch.float32, device=dev),
                  torch.empty((B, 16), dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))

        def tr():
            _lib.check(L.bpg_trim_expert_inputs(C.c_void_p(tr_o.data_ptr()), C.c_void_p(tr_g.data_ptr()), C.c_void_p(tr_ag.data_ptr()),
                                                B, 87, 36, 40, 16, 4, 1, C.c_void_p(tr_out[0].data_ptr()),
                                                C.c_void_p(tr_out[1].data_ptr()), C.c_void_p(tr_out[2].data_ptr()),
                                                C.c_void_p(tr_out[3].data_ptr()), dev.index, stream))
        res['trim_expert_inputs_variation'] = line(timed(tr, 20), B * (87 * 4 + 2 * 36 * 4 + 40 * 8 + 2 * 16 * 4), B)
        assert int(tr_out[3].item()) == 0
        del tr_o, tr_g, tr_ag, tr_out
    except Exception as e:  # noqa: BLE001
        res['trim_expert_inputs_variation'] = {'error': repr(e)[:200]}

    # three (reward, G) pairs round-robin: neither the 210 MB read nor the 419 MB written find the
    # previous launch's lines in the 256 MiB Infinity Cache
    NDR = 3
    dr_in = [reward] + [torch.empty((T, LD), dtype=torch.float32, device=dev)[:, :B] for _ in range(NDR - 1)]
    for r_ in dr_in[1:]:
        r_.copy_(reward)
    dr_out = [torch.empty((B, T), dtype=torch.float64, device=dev) for _ in range(NDR)]
    dr_it = [0]

    def dr():
        k = dr_it[0] % NDR
        dr_it[0] += 1
        _lib.check(L.bpg_discounted_return_ld(C.c_void_p(dr_in[k].data_ptr()), B, LD, T, 1.0 - 1.0 / T,
                                              C.c_void_p(dr_out[k].data_ptr()), dev.index, stream))
    for _ in range(NDR):          # first touch of the fresh buffers is not part of the figure
        dr()
    res['discounted_return_f64'] = line(timed(dr, 4 * NDR), B * T * 12, B * T)

    # (3e') the O(T) scan form of the same return (tolerance-based, include/bpgym.h)
    def drs():
        k = dr_it[0] % NDR
        dr_it[0] += 1
        _lib.check(L.bpg_discounted_return_scan(C.c_void_p(dr_in[k].data_ptr()), B, LD, T, 1.0 - 1.0 / T,
                                                C.c_void_p(dr_out[k].data_ptr()), dev.index, stream))
    res['discounted_return_scan_f64'] = line(timed(drs, 4 * NDR), B * T * 12, B * T)

    # (3f) the policy-gradient path (policy_gradient/rollout.py): one rollout + its return G (B, T) float64.
    # round-2 form: reward + flags streams, then the return kernel on the reward stream (9 + 12 = 21 B/env-step);
    # round-3 form: ONE bpg_rollout call with d_G -- flags stream + 8-byte step-ok words, expanded by a second
    # launch of the same call (5 + 8 + 0.3 B/env-step), no reward stream.
    try:
        m_sets = [masks] + [torch.empty((T, LD), dtype=torch.int32, device=dev)[:, :B] for _ in range(2)]
        for m_ in m_sets[1:]:
            m_.copy_(masks)
        pg_it = [0]
        ngp = ((torch.arange(B, device=dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
        a_old = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ngp.data_ptr(), d_masks=m_.data_ptr(), ld_masks=LD,
                                  d_reward_tb=dr_in[k].data_ptr(), d_flags_tb=flags.data_ptr(), ld_out=LD)
                 for k, m_ in enumerate(m_sets)]
        a_new = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ngp.data_ptr(), d_masks=m_.data_ptr(), ld_masks=LD,
                                  d_flags_tb=flags.data_ptr(), ld_out=LD, d_G=dr_out[k].data_ptr(), gamma=1.0 - 1.0 / T)
                 for k, m_ in enumerate(m_sets)]

        def pg_old():
            k = pg_it[0] % NDR
            pg_it[0] += 1
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(a_old[k]), stream))
            _lib.check(L.bpg_discounted_return_ld(C.c_void_p(dr_in[k].data_ptr()), B, LD, T, 1.0 - 1.0 / T,
                                                  C.c_void_p(dr_out[k].data_ptr()), dev.index, stream))

        def pg_new():
            k = pg_it[0] % NDR
            pg_it[0] += 1
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(a_new[k]), stream))
        res['pg_rollout_plus_return_via_reward_stream'] = line(timed(pg_old, 4 * NDR), B * (T * 21 + 12), B * T)
        res['pg_rollout_with_return_from_step_ok_words'] = line(timed(pg_new, 4 * NDR), B * (T * 13 + 12 + 16), B * T)
        del m_sets, a_old, a_new
    except Exception as e:  # noqa: BLE001
        res['pg_rollout_with_return_from_step_ok_words'] = {'error': repr(e)[:300]}
    del dr_in, dr_out

    # (4) episode writer: float32 episode['ag'] (B,T+1,36), ['g'] (B,T,36), ['info_is_success'] (B,T,1)
    try:
        dimg = env.dimg
        ag = torch.empty((B, T + 1, dimg), dtype=torch.float32, device=dev)
        gg = torch.empty((B, T, dimg), dtype=torch.float32, device=dev)
        sc = torch.empty((B, T, 1), dtype=torch.float32, device=dev)
        a3 = _lib.RolloutArgs(T=T, reset_first=1, d_masks=masks.data_ptr(), ld_masks=LD,
                              d_ep_ag=ag.data_ptr(), d_ep_g=gg.data_ptr(), d_ep_success=sc.data_ptr())

        def episode():
            _lib.check(L.bpg_rollout_v2(env._h, C.byref(a3), stream))
        nbytes = B * (16 + T * 4 + 4 * ((T + 1) * dimg + T * dimg + T))
        res['episode_writer_f32'] = line(timed(episode, 5), nbytes, B * T)
        del ag, gg, sc
    except Exception as e:  # noqa: BLE001
        res['episode_writer_f32'] = {'error': repr(e)[:200]}

    # (4b) the batch as S env shards on S streams of the one GPU (distributed.StreamShards): the same partition that spreads a
    # job over GPUs, once more inside one -- independent launches that the hardware runs side by side, so that one shard's
    # ramp-up and drain are covered by the others' streaming.  K rollouts per shard back to back, nothing joins the shards
    # in between; ms = wall time per rollout of ALL shards (the same env-steps and bytes as the headline launch).
    try:
        from blockpuzzle_gym_amd import distributed as D
        for tot, S in ((B, 4), (524288, 2)):
            res['rollout_masks_%d_envs_as_%d_shards_on_%d_streams' % (tot, S, S)] = shards_on_streams(torch, _lib, D, tot, S, T, dev, line)
    except Exception as e:  # noqa: BLE001
        res['rollout_masks_as_shards_on_streams'] = {'error': repr(e)[:300]}

    # (5) BASELINE configs[4]: the rollout worker with a policy in the loop
    for Bx in (65536, B):
        try:
            res['rollout_policy_in_loop_%d_envs' % Bx] = policy_in_loop(torch, bpg, Bx, T, dev)
        except Exception as e:  # noqa: BLE001
            res['rollout_policy_in_loop_%d_envs' % Bx] = {'error': repr(e)[:300]}
    return res


def shards_on_streams(torch, _lib, D, total, S, T, dev, line, reps=120):
    L = _lib.lib()
    sh = D.StreamShards(ENV_ID, total, S, device=dev, seed=7)
    nset = max(4, int(2.5 * 256 * 2**20 // (total * T * 9)) + 1)
    per = []
    for i, (env, st) in enumerate(sh.shards):
        with torch.cuda.stream(st):
            cs = C.c_void_p(st.cuda_stream)
            n = env.num_envs
            ng = ((torch.arange(n, device=dev, dtype=torch.int32) + env.env_id_base) * 7) % 3
            ld = int(L.bpg_row_stride(n))
            sets = []
            for k in range(nset):
                env.seed(7 + k)
                m = torch.empty((T, ld), dtype=torch.int32, device=dev)[:, :n]
                _lib.check(L.bpg_synth_masks_ld(env._h, 0, T, C.c_void_p(m.data_ptr()), ld, cs))
                sets.append((m, torch.empty((T, ld), dtype=torch.float32, device=dev)[:, :n],
                             torch.empty((T, ld), dtype=torch.uint8, device=dev)[:, :n]))
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            args = [_lib.RolloutArgs(T=T, reset_first=1, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), d_reward_tb=r.data_ptr(),
                                     d_flags_tb=f.data_ptr(), ld_masks=ld, ld_out=ld, d_success_count=cnt.data_ptr()) for m, r, f in sets]
            per.append((env, cs, args, (ng, sets, cnt)))
    torch.cuda.synchronize(dev)

    def run(n):
        for i in range(n):
            for env, cs, args, _ in per:
                _lib.check(L.bpg_rollout_v2(env._h, C.byref(args[i % nset]), cs))
    run(40)
    torch.cuda.synchronize(dev)
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(reps)
        torch.cuda.synchronize(dev)
        best.append((time.perf_counter() - t0) / reps * 1e3)
    out = line(min(best), total * (T * 9 + 12), total * T)
    out.update(shards=S, envs_per_shard=[e.num_envs for e, _ in sh.shards], runs_ms=best, buffer_sets_per_shard=nset,
               note='wall time per rollout of all shards over %d rollouts per shard back to back (min of 3 runs); launches of '
                    'different shards overlap, so no per-launch duration is quoted' % reps)
    return out


def rollout_at(torch, bpg, _lib, Bx, T, dev, stream, timed, line, rollouts=1, episode_return=False):
    """The bench headline (one fused reset + T-step + success-count launch, masks from HBM) at
    another batch size; rollouts = K > 1: K consecutive rollouts (the evaluator's loop, train.py:63-64)
    in ONE launch, every figure per rollout."""
    L = _lib.lib()
    K = rollouts
    e = bpg.make(ENV_ID, num_envs=Bx, device=dev, seed=7)
    ng = ((torch.arange(K * Bx, device=dev, dtype=torch.int32) * 7) % 3).reshape(K, Bx)
    e.reset(num_grey=ng[0])
    nset = int(2.5 * 256 * 2**20 // (K * Bx * T * 9)) + 1
    sets = []
    for k in range(nset):
        e.seed(7 + k)
        m = bpg.time_major_empty(K * T, Bx, torch.int32, dev)    # rows bpg_row_stride(Bx) apart, as the headline
        ld = int(m.stride(0))
        _lib.check(L.bpg_synth_masks_ld(e._h, 0, K * T, C.c_void_p(m.data_ptr()), ld, stream))
        sets.append((m, bpg.time_major_empty(K * T, Bx, torch.float32, dev), bpg.time_major_empty(K * T, Bx, torch.uint8, dev)))
    cnt = torch.zeros(max(2, K), dtype=torch.int64, device=dev)
    ret = torch.empty((K, Bx), dtype=torch.float32, device=dev) if episode_return else None
    args = [_lib.RolloutArgs(T=T, reset_first=1, rollouts=K, d_num_grey=ng.data_ptr(), d_masks=m.data_ptr(), ld_masks=ld,
                             ld_out=ld, d_reward_tb=r.data_ptr(), d_flags_tb=f.data_ptr(), d_success_count=cnt.data_ptr(),
                             d_episode_return=ret.data_ptr() if episode_return else None)
            for m, r, f in sets]
    it = [0]

    def go():
        _lib.check(L.bpg_rollout_v2(e._h, C.byref(args[it[0] % nset]), stream))
        it[0] += 1
    out = line(timed(go, max(40, 2 * nset)) / K, Bx * (T * 9 + 12 + (4 if episode_return else 0)), Bx * T)
    out['buffer_sets'] = nset
    out['note'] = 'ms = events around a train of back-to-back launches (includes the launch gaps; see launch_floor_smallest_kernel)'
    if K > 1:
        out['rollouts_per_launch'] = K
        out['note'] = 'per rollout: ' + out['note'] + ' / rollouts_per_launch'
    return out


def policy_in_loop(torch, bpg, Bx, T, dev):
    import numpy as np
    """RolloutStudent.generate_rollouts with a device-resident policy between the env steps
    (rollout.py:75-172): eager (one step launch + policy per t) vs the whole loop replayed from a
    hipGraph.  Two policies: a trivial one (isolates the env side: step kernel writing straight
    into the episode slots + the action copy) and a 3 x 256 MLP on the observation with the input
    contract of the reference's ActorCritic (actor_critic.py:213-255: normalised, clipped `o` ->
    3 hidden layers of 256 -> tanh * max_u), bf16 weights -- torch is only the policy here."""
    dimo, dimg = 87, 36

    class Trivial:
        device_tensors = True

        def __init__(self):
            self.u = torch.zeros((Bx, 4), device=dev)

        def get_actions(self, o, ag, g, **kw):
            return self.u

    class MLP:
        device_tensors = True

        def __init__(self):
            gen = torch.Generator(device=dev).manual_seed(0)
            dims = [dimo, 256, 256, 256, 4]
            self.w = [(torch.randn((dims[i], dims[i + 1]), device=dev, generator=gen) / dims[i] ** 0.5).to(torch.bfloat16)
                      for i in range(4)]
            self.mean = torch.zeros(dimo, device=dev)
            self.inv_std = torch.ones(dimo, device=dev)

        def get_actions(self, o, ag, g, **kw):
            x = torch.clamp((o - self.mean) * self.inv_std, -5.0, 5.0).to(torch.bfloat16)   # Normalizer.normalize
            for w in self.w[:-1]:
                x = torch.relu(x @ w)
            return torch.tanh((x @ self.w[-1]).float())

    # per env-step: now + ever + mask 12 in; now + ever 8, is_success 4, the achieved_goal row 4 * dimg and the action copy 16
    # out (the worker ignores the step reward, rollout.py:123-125, and reads is_success as float: neither the reward nor
    # the flags byte is produced since round 4 -- 189 B before)
    env_bytes = 12 + 8 + 4 + 4 * dimg + 16
    out = {'env_side_bytes_per_env_step': env_bytes,
           'note': 'ms per T=%d rollout of %d envs, reset included; env side = step + f32 achieved_goal row + f32 '
                   'is_success written into the episode slots, actions stored; o is an expanded view (no backend '
                   'writes observations)' % (T, Bx)}
    for name, pol, reps in (('trivial_policy', Trivial(), 5), ('mlp_3x256_bf16', MLP(), 2)):
        w = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=dev), pol,
                               dict(o=dimo, u=4, g=dimg, info_is_success=1), None, T, rollout_batch_size=Bx)
        for mode, fn in (('eager', w.generate_rollouts), ('hipgraph', w.generate_rollouts_graphed)):
            fn()
            fn()                                     # both episode slabs the worker alternates between exist now
            torch.cuda.synchronize(dev)
            each = []
            for _ in range(reps + 2):                # per rollout, median: memory the process released earlier
                t0 = time.perf_counter()             # (torch.cuda.empty_cache) is reclaimed by the driver in the
                fn()                                 # background and stalls a launch for tens of ms now and then
                torch.cuda.synchronize(dev)
                each.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(each)[len(each) // 2]
            out['%s_%s' % (name, mode)] = {'ms': ms, 'env_steps_per_sec': Bx * T / (ms * 1e-3), 'ms_min': min(each),
                                           'ms_max': max(each)}
            if mode == 'hipgraph':
                # the replayed graph alone (reset + T x (policy, step launch)), HIP events around a train of replays: the
                # figure above also holds the host's part of a rollout (graph launch, reading the success count back)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                nrep = 10 * reps
                e0.record()
                for _ in range(nrep):
                    w._graph.replay()
                e1.record()
                torch.cuda.synchronize(dev)
                w._resync_success_total()
                gms = e0.elapsed_time(e1) / nrep
                nbytes = Bx * T * env_bytes
                out['%s_%s' % (name, mode)].update(graph_only_ms=gms, graph_only_us_per_step=gms * 1e3 / T,
                                                   graph_only_env_side_frac_hbm_peak=nbytes / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        del w
        if name == 'trivial_policy':
            # the same replayed rollouts from a worker that never reads the success count back inside generate_rollouts
            # (async_stats: success_history resolves its entries when they are read -- once per epoch in the reference's
            # loop, train.py:67-76): 20 rollouts back to back, ONE synchronisation at the end
            wa = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=dev), pol,
                                    dict(o=dimo, u=4, g=dimg, info_is_success=1), None, T, rollout_batch_size=Bx, async_stats=True)
            for _ in range(3):
                wa.generate_rollouts_graphed()
            torch.cuda.synchronize(dev)
            nroll = 20
            t0 = time.perf_counter()
            for _ in range(nroll):
                wa.generate_rollouts_graphed()
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / nroll * 1e3
            rate = float(np.mean(wa.success_history))
            out['trivial_policy_hipgraph_async_stats'] = {
                'ms': ms, 'env_steps_per_sec': Bx * T / (ms * 1e-3), 'rollouts_back_to_back': nroll,
                'env_side_frac_hbm_peak': Bx * T * env_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'mean_success_rate_read_afterwards': rate}
            del wa
        torch.cuda.empty_cache()
    return out


if __name__ == '__main__':
    main()
