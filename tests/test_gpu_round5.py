"""Round 5 GPU tests: the reference's restart semantics replayed from fixtures recorded from the reference itself
(tests/golden/restart_*.npz, pg_restart_*.npz <- make_golden.restart_trace), the sharded default of the fused
workers, the u8 episode-return gather, the curriculum and collective entry points of the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import blockpuzzle_gym_amd as bpg
from blockpuzzle_gym_amd import _lib
from util import GOLDEN, VARIANTS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dims_of(env_id):
    v = VARIANTS[env_id]
    return dict(o=v['dimo'], u=4, g=v['dimg'], info_is_success=1)


class _Log:
    def __init__(self):
        self.lines = []

    def warning(self, m):
        self.lines.append(m)

    info = warning


# --------------------------------------------------------------------------
# rollout.py:136-142 / policy_gradient/rollout.py:237-246, pinned by the reference
# --------------------------------------------------------------------------
class _FaultyBackend:
    """Replays one generate_rollouts call of a restart fixture: the contact lists of the discarded attempt (if any), the
    fault at its last step -- a NaN in env j's observation, or PhysicsError(failed_env=j) with the step's contacts of the
    envs before j -- then the T contact lists of the attempt the reference returned."""

    def __init__(self, B, dimo):
        self.B, self.dimo = B, dimo
        self.resets = 0

    def load(self, d, r, T):
        kind, j, k = (int(x) for x in d['faults'][r])
        self.kind, self.j, self.k, self.T = kind, j, k, T
        dc = d['r%d_discard_contacts' % r].copy()
        dc[dc < 0] = 0
        self.dcon, self.dncon = dc, d['r%d_discard_ncon' % r]
        c = d['r%d_contacts' % r].copy()
        c[c < 0] = 0
        self.con, self.ncon = c, d['r%d_ncon' % r]
        self.attempt_step, self.faulted, self.calls = 0, kind == 0, 0

    def on_reset(self, env, xy, status, env_mask, test):
        self.resets += 1
        self.attempt_step = 0

    def __call__(self, env, actions):
        self.calls += 1
        t = self.attempt_step
        self.attempt_step += 1
        env.observation.zero_()                       # "physics" writes the whole observation every step
        if not self.faulted:
            pairs = torch.as_tensor(np.ascontiguousarray(self.dcon[:, min(t, self.dcon.shape[1] - 1)]), device=DEV)
            ncon = torch.as_tensor(np.ascontiguousarray(self.dncon[:, min(t, self.dncon.shape[1] - 1)]), device=DEV)
            if t == self.k:
                self.faulted = True
                if self.kind == 2:
                    # envs j .. B-1 have not stepped in the reference: their lists of this step do not exist
                    raise bpg.PhysicsError('scripted simulation failure', failed_env=self.j, contacts=(pairs, ncon))
                env.observation[self.j, 0] = float('nan')
            return pairs, ncon
        return (torch.as_tensor(np.ascontiguousarray(self.con[:, t]), device=DEV),
                torch.as_tensor(np.ascontiguousarray(self.ncon[:, t]), device=DEV))


def _replay_restarts(name, pg, nan_check='per_step'):
    env_id = name + '-v0'
    d = np.load(os.path.join(GOLDEN, ('pg_restart_%s.npz' if pg else 'restart_%s.npz') % name))
    B, T = int(d['B']), int(d['T'])
    backend = _FaultyBackend(B, VARIANTS[env_id]['dimo'])
    src = bpg.CallableContacts(backend, on_reset=backend.on_reset, writes_observation=True)

    class Replay:
        kwargs = {}

        def __init__(self):
            self.calls, self.seen = 0, []

        def get_actions(self, o, ag, g, **kw):
            self.seen.append(dict(kw))
            u = self.u[self.calls]
            self.calls += 1
            return (u, u.copy(), np.zeros_like(u)) if pg else u

    policy, log = Replay(), _Log()
    make = lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src)   # noqa: E731
    if pg:
        worker = bpg.PGRolloutStudent(make, policy, None, dims_of(env_id), log, T, rollout_batch_size=B, gamma=1. - 1. / T,
                                      nan_check=nan_check)
    else:
        worker = bpg.RolloutStudent(make, policy, dims_of(env_id), log, T, rollout_batch_size=B, nan_check=nan_check)
    for r in range(int(d['n_calls'])):
        backend.load(d, r, T)
        # the policy's actions in call order: the discarded attempt's, then the returned episode's (time-major)
        policy.u = np.concatenate([d['r%d_discard_u' % r], d['r%d_u' % r].transpose(1, 0, 2)])
        policy.calls, policy.seen = 0, []
        draws = iter(d['r%d_reset_num_objs' % r] - 4)            # one row per reset of this call
        n_before, h_before, resets_before, lines_before = worker.n_episodes, len(worker.success_history), backend.resets, len(log.lines)
        kw = dict(exploit=True) if pg and name == 'BlocksTouch' else {}    # make_golden: exploit_first
        ep = worker.generate_rollouts(num_grey=lambda: next(draws), **kw)
        for k in ('ag', 'g', 'u', 'info_is_success'):
            np.testing.assert_array_equal(ep[k], d['r%d_%s' % (r, k)], err_msg='call %d: %s' % (r, k))
        if pg:
            np.testing.assert_array_equal(ep['G'].view(np.uint64), d['r%d_G' % r].view(np.uint64))
        assert worker.success_history[-1] == pytest.approx(float(d['r%d_success_rate' % r]), abs=1e-7)
        assert worker.n_episodes - n_before == int(d['r%d_n_episodes_added' % r])
        assert len(worker.success_history) - h_before == int(d['r%d_history_added' % r])
        assert policy.calls == int(d['r%d_policy_calls' % r])
        assert backend.resets - resets_before == d['r%d_reset_num_objs' % r].shape[0]
        assert next(draws, None) is None
        assert len(log.lines) - lines_before == (1 if d['faults'][r][0] else 0)
        # the state the envs are left in: the matrix and the latch
        np.testing.assert_array_equal(worker.env.achieved_goal_f64().cpu().numpy(), d['r%d_final_ag' % r])
        np.testing.assert_array_equal(((worker.env.state_ever >> 31) & 1).cpu().numpy().astype(np.uint8),
                                      d['r%d_final_has_succeeded' % r])
        if pg:
            np.testing.assert_array_equal(np.asarray([bool(k.get('exploit')) for k in policy.seen], np.uint8),
                                          d['r%d_exploit_seen' % r])
    assert worker.n_episodes == int(d['n_episodes'])


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation', 'ToppleTower'])
def test_golden_restarts_rollout_student(name):
    """RolloutStudent.generate_rollouts around a NaN observation and a MujocoException (rollout.py:136-142), bit for bit
    what the reference returned -- including what the discarded attempt left in the touch matrix (quirk Q1)."""
    _replay_restarts(name, pg=False)


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation'])
def test_golden_restarts_pg_rollout_student(name):
    """policy_gradient/rollout.py:243-246: reset, then a retry whose `exploit` is back at its default."""
    _replay_restarts(name, pg=True)


def test_per_rollout_nan_check_is_the_documented_deviation():
    """nan_check='per_rollout' reads the flag once after all T steps: the discarded attempt runs to its end, so the
    backend is called T times instead of t+1 and (quirk Q1) the retried episode of a base variant can start from more
    touches than the reference's.  The test pins the call count and that the returned episode is still a clean one."""
    d = np.load(os.path.join(GOLDEN, 'restart_BlocksTouch.npz'))
    B, T = int(d['B']), int(d['T'])
    env_id = 'BlocksTouch-v0'
    backend = _FaultyBackend(B, VARIANTS[env_id]['dimo'])
    src = bpg.CallableContacts(backend, on_reset=backend.on_reset, writes_observation=True)

    class Zero:
        def get_actions(self, o, ag, g, **kw):
            return np.zeros((B, 4), np.float32)

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV, contact_source=src), Zero(),
                           dims_of(env_id), _Log(), T, rollout_batch_size=B, nan_check='per_rollout')
    backend.load(d, 1, T)                              # ('nan', env 2, step 7)
    ep = w.generate_rollouts()
    assert backend.calls == T + T                      # per_step (the reference): 8 + T
    assert not np.isnan(ep['o']).any() and w.n_episodes == B and len(w.success_history) == 1
    with pytest.raises(ValueError):
        bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), Zero(), dims_of(env_id), None, T,
                           rollout_batch_size=B, nan_check='sometimes')


def test_policy_action_width_is_checked():
    """a (B, k != 4) float32 device tensor from the policy must not reach the kernel as a raw pointer."""
    env_id, B, T = 'BlocksTouch-v0', 16, 3

    class Wide:
        device_tensors = True

        def get_actions(self, o, ag, g, **kw):
            return torch.zeros((B, 5), device=DEV)

    w = bpg.RolloutStudent(lambda num_envs: bpg.make(env_id, num_envs=num_envs, device=DEV), Wide(), dims_of(env_id), None, T,
                           rollout_batch_size=B, output='torch')
    with pytest.raises(ValueError):
        w.generate_rollouts()
    with pytest.raises(ValueError):
        w.env._step_with_args(torch.zeros((B, 3), device=DEV), _lib.StepArgs())
