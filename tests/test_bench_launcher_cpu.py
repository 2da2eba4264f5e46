"""`python bench.py --gpus N` starts its own rank processes (the reference's mpi_fork,
gym_blocks/train.py:113-119).  CPU-only checks of the parent: the environment every rank gets, that
the parent neither imports torch nor touches the GPU, that a failing rank produces ONE parseable JSON
error line and a non-zero exit code, and that a rank's own failures are JSON lines too."""
import argparse
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _bench_module():
    spec = importlib.util.spec_from_file_location('bpg_bench_for_test', BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rank_environments_follow_the_torchrun_contract():
    b = _bench_module()
    a = argparse.Namespace(gpus=4, single_device=False)
    envs = b.rank_environments(a, base_env={'PATH': '/bin', 'WORLD_SIZE': 'stale'}, port=12345)
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3']
    assert [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert all(e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '12345'
               and e['PATH'] == '/bin' and e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' for e in envs)
    a1 = argparse.Namespace(gpus=2, single_device=True)
    assert [e['LOCAL_RANK'] for e in b.rank_environments(a1, base_env={}, port=1)] == ['0', '0']


def test_parent_does_not_import_torch_or_touch_the_gpu():
    """The parent path of main() runs before any torch import: importing bench.py and building the rank
    environments must leave torch out of sys.modules."""
    code = ('import sys, importlib.util; sys.argv=["bench.py"];'
            'spec=importlib.util.spec_from_file_location("b", %r); m=importlib.util.module_from_spec(spec);'
            'spec.loader.exec_module(m); import argparse;'
            'm.rank_environments(argparse.Namespace(gpus=2, single_device=False), base_env={}, port=1);'
            'assert "torch" not in sys.modules, "parent imported torch"; print("clean")' % BENCH)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'clean' in out.stdout, out.stderr


def test_self_launch_reports_rank_failures_as_one_json_line():
    """No GPU in this container: every rank fails in set-up.  The parent must still end with exactly one
    JSON line carrying an "error" key and exit non-zero -- never a bare assert / traceback only."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--single-device', '--backend', 'gloo',
                          '--total-envs', '4096', '--steps', '2', '--warmup', '1', '--prewarm', '0', '--no-extras',
                          '--no-cpu-baseline', '--launch-timeout', '240'],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    import torch
    if torch.cuda.is_available():          # on a GPU box this very command works: covered by the -m gpu test
        return
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['metric'] == 'env_steps_per_sec' and line['value'] is None and 'error' in line
    assert line['n_gpus'] == 2 and line['launcher']['rank_exit_codes'] == [2, 2]


def test_world_size_mismatch_is_a_json_error_not_an_assert():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--no-extras', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 2
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert 'error' in line and 'WORLD_SIZE' in line['error']


def test_collective_schedule_of_a_bench_leg():
    """One collective per R rollouts, plus one behind the warm-up and one behind the last rollout; every rollout belongs to
    exactly one exchange group and groups advance by one per collective (the two ping-pong return buffers rely on it)."""
    b = _bench_module()
    ca, grp = b.collective_schedule(8, 2, 3, True)                    # warm-up 2 + 6 timed, every third rollout
    assert [i for i, c in enumerate(ca) if c] == [1, 2, 5, 7] and grp == [0, 0, 1, 2, 2, 2, 3, 3]
    ca, grp = b.collective_schedule(6, 2, 1, True)                    # BASELINE configs[3]: one all-gather per rollout
    assert all(ca) and grp == list(range(6))
    ca, grp = b.collective_schedule(110, 10, 50, True)                # the default headline: once per 50 rollouts
    assert [i for i, c in enumerate(ca) if c] == [9, 49, 99, 109]
    assert b.collective_schedule(5, 1, 2, False) == ([False] * 5, [0] * 5)
    for n_it, w, R in [(7, 3, 2), (12, 0, 5), (3, 1, 1)]:
        ca, grp = b.collective_schedule(n_it, w, R, True)
        assert ca[-1] and all(grp[i + 1] == grp[i] + ca[i] for i in range(n_it - 1))
