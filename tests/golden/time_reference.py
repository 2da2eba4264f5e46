#!/usr/bin/env python3
"""Time the REFERENCE ITSELF on the CPU of the build container and write profiles/reference_cpu_timing.json.

Run in the build container only (needs /root/reference; it never travels to the GPU box):

    python tests/golden/time_reference.py [--steps 20000]

What is timed (one core, one env object, exactly the objects tests/golden/make_golden.py drives):

  step        `env.step(a)` of the registered BlocksTouchVariation-v0 through the TimeLimit stand-in -- the whole
              RobotEnv.step template (gym_blocks/envs/robot_env.py:57-69: clip, _set_action, sim.step, _step_callback,
              _get_obs, _is_success, compute_reward) with the third-party pieces replaced by the inert stand-ins of
              _ref_harness.py: sim.step() installs a pre-generated scripted contact list (NO physics is simulated -- MuJoCo's
              20 substeps, the real cost of the reference's step, are absent here and everywhere else in this image),
              mat2euler / ctrl_set_action / mocap_set_action are trivial lambdas.
  goal_logic  the reference-owned functions behind the contacts-in boundary, called as the step template calls them,
              on the same env object: _step_callback + _check_goal (as _get_obs runs it, fetch_env.py:219) + _is_success
              + compute_reward (fetch_env.py:119-167, 275-281) -- what the HIP path replaces.
  reset       `env.reset()` (robot_env.py:71-82, fetch_env.py:646-679), every 50 steps inside `step`'s loop (TimeLimit).

The contact lists come from the harness's seeded ContactScript and are generated BEFORE the timed loops (the timed
region replays them), so the generator's own cost is not in the figures.  bench.py quotes the JSON verbatim as
`cpu_baseline.reference_python` (where: "build container"): it is context for the GPU figures, measured on other
hardware than the GPU box, on the reference's per-env Python objects.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402

ENV_ID = 'BlocksTouchVariation-v0'


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def replay_scripts(env, seed, n):
    """Pre-generate n contact lists per sim with the harness's seeded script; every sim then replays its list."""
    u = env.unwrapped
    sims = list(getattr(u, 'sims', None) or [u.sim])
    if u.sim not in sims:
        sims.append(u.sim)
    ncon = []
    for k, sim in enumerate(sims):
        script = H.ContactScript(sim.model, seed * 17 + k)
        lists = [script() for _ in range(n)]
        ncon.append(float(np.mean([len(p) for p in lists])))
        state = {'i': 0}

        def run(lists=lists, state=state):
            p = lists[state['i'] % len(lists)]
            state['i'] += 1
            return p
        sim.script = run
    return ncon


def best_of(fn, repeats):
    ts = []
    for _ in range(repeats):
        ts.append(fn())
    return min(ts), ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20000)
    ap.add_argument('--repeats', type=int, default=3)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'reference_cpu_timing.json'))
    a = ap.parse_args()
    N, T = a.steps, 50
    env = H.make(ENV_ID)
    u = env.unwrapped
    np.random.seed(0)
    env.seed(0)
    mean_ncon = replay_scripts(env, 3, 4096)
    actions = np.random.RandomState(1).uniform(-1, 1, size=(N, 4)).astype(np.float32)
    env.reset()

    def time_step():
        env.reset()
        n_reset = 0
        t_reset = 0.0
        t0 = time.perf_counter()
        for i in range(N):
            if i and i % T == 0:
                r0 = time.perf_counter()
                env.reset()
                t_reset += time.perf_counter() - r0
                n_reset += 1
            env.step(actions[i])
        dt = time.perf_counter() - t0
        time_step.reset_us = t_reset / max(n_reset, 1) * 1e6
        return (dt - t_reset) / N * 1e6

    def time_goal_logic():
        env.reset()
        sim = u.sim
        t0 = time.perf_counter()
        for i in range(N):
            sim.step()                                   # installs the next scripted contact list (no physics)
            u._step_callback()                           # fetch_env.py:148-167
            u._check_goal()                              # :119-124, as _get_obs runs it (:219)
            ag = u.achieved_goal.ravel()
            u._is_success(ag, u.goal)                    # :275-281 (recomputes the reward, latches)
            u.compute_reward(ag, u.goal, {})             # :135-143, robot_env.py:66
        return (time.perf_counter() - t0) / N * 1e6

    def time_sim_step_only():
        env.reset()
        sim = u.sim
        t0 = time.perf_counter()
        for i in range(N):
            sim.step()
        return (time.perf_counter() - t0) / N * 1e6

    step_us, step_runs = best_of(time_step, a.repeats)
    reset_us = time_step.reset_us
    goal_us, goal_runs = best_of(time_goal_logic, a.repeats)
    stub_us, _ = best_of(time_sim_step_only, a.repeats)
    out = {
        'where': 'build container (NOT the GPU box: the reference cannot travel)',
        'env_id': ENV_ID, 'steps_timed': N, 'repeats': a.repeats, 'cores': 1,
        'cpu_model': cpu_model(), 'python': platform.python_version(), 'numpy': np.__version__,
        'step_us': step_us, 'step_runs_us': step_runs,
        'step_env_steps_per_sec': 1e6 / step_us,
        'goal_logic_us': goal_us - stub_us, 'goal_logic_runs_us_incl_stub': goal_runs, 'contact_stub_us': stub_us,
        'goal_logic_env_steps_per_sec': 1e6 / (goal_us - stub_us),
        'reset_us': reset_us,
        'mean_contacts_per_step': mean_ncon,
        'what_step_is': 'gym_blocks RobotEnv.step (robot_env.py:57-69) of BlocksTouchVariationEnv through the TimeLimit '
                        'stand-in, reset every 50 steps (reset time excluded and reported separately)',
        'what_goal_logic_is': '_step_callback + _check_goal + _is_success + compute_reward (fetch_env.py:119-167, 275-281) on '
                              'the same env object, minus the cost of the stub that installs the scripted contact list',
        'stubbed': ['mujoco_py.MjSim.step -> installs a pre-generated contact list, NO physics (the real cost of the '
                    "reference's step is absent)", 'gym.envs.robotics.utils / rotations -> inert lambdas',
                    'gym TimeLimit -> counting stand-in', 'gym.utils.seeding -> np.random.RandomState(seed)'],
        'script': 'tests/golden/time_reference.py',
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
        f.write('\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
