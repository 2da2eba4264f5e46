#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's gym_blocks package through the stand-in modules of
_ref_harness.py, builds the real registered env classes
(gym_blocks/envs/tasks.py) on a scripted fake sim, and records what the
reference's own step()/reset()/compute_reward()/RolloutStudent produce for
seeded scripted contact lists.  Outputs are DATA ONLY (json / npz); neither the
reference nor this harness is needed to run the test-suite.

Explicit inputs recorded alongside the outputs (because the third-party RNG /
physics cannot be reproduced): the contact list of every step (geom id pairs),
`num_objs` of every Variation episode, the actions of every rollout.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402

MAXCON = H.FakeData.MAXCON
ENV_IDS = ['GripperTouch-v0', 'BlocksTouch-v0', 'ToppleTower-v0',
           'BlocksTouchCurriculum-v0', 'BlocksTouchChoose-v0',
           'BlocksTouchChooseCurriculum-v0', 'BlocksTouchVariation-v0']


def f32bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def attach_scripts(env, seed, log, p_pair=0.15):
    """Give every fake sim of `env` a seeded contact script writing to `log`."""
    u = env.unwrapped
    sims = getattr(u, 'sims', None) or [u.sim]
    if u.sim not in sims:
        sims = sims + [u.sim]
    for k, sim in enumerate(sims):
        script = H.ContactScript(sim.model, seed * 17 + k, p_pair=p_pair)

        def run(script=script):
            pairs = script()
            log.append(pairs)
            return pairs
        sim.script = run


def pad_contacts(pairs):
    out = -np.ones((MAXCON, 2), np.int16)
    for k, (a, b) in enumerate(pairs):
        out[k] = (a, b)
    return out


def variant_info(env_id):
    env = H.make(env_id)
    u = env.unwrapped
    np.random.seed(1)
    env.seed(1)
    obs = env.reset()
    info = dict(
        env_id=env_id,
        dimo=int(obs['observation'].shape[0]),
        dimg=int(obs['desired_goal'].shape[0]),
        block_gripper=bool(u.block_gripper),
        obj_colors=[int(c) for c in u.obj_colors],
        goal=[int(v) for v in obs['desired_goal']],
        desired_goal_dtype=str(obs['desired_goal'].dtype),
        achieved_goal_dtype=str(obs['achieved_goal'].dtype),
        observation_dtype=str(obs['observation'].dtype),
        max_episode_steps=int(env._max_episode_steps),
        action_shape=list(env.action_space.shape),
    )
    sims = getattr(u, 'sims', None) or [u.sim]
    info['models'] = []
    for sim in sims:
        names = sim.model.geom_names
        # the reference's own _geom2objid over the real geom names
        u_sim_backup = u.sim
        u.sim = sim
        ids = [u._geom2objid(i) for i in range(sim.model.ngeom)]
        u.sim = u_sim_backup
        info['models'].append(dict(
            xml=os.path.relpath(sim.model.path, H.ASSET_DIR),
            geom_names=names,
            id2obj=[-1 if v is None else int(v) for v in ids]))
    # curriculum surface (fetch_env.py:93-101, 351-358, 419-432, 623-630)
    for name in ['set_test', 'increase_difficulty']:
        try:
            getattr(u, name)()
            info[name] = 'ok'
        except NotImplementedError:
            info[name] = 'NotImplementedError'
        except AttributeError:
            info[name] = 'AttributeError'
    return info


def difficulty_trace(env_id, n=12):
    env = H.make(env_id)
    u = env.unwrapped
    rows = []
    for _ in range(n):
        try:
            r = u.increase_difficulty()
        except (NotImplementedError, AttributeError) as e:
            return dict(error=type(e).__name__)
        rows.append(dict(ret=bool(r), difficulty=int(u.get_difficulty()),
                         obj_range=float(u.obj_range),
                         wrong_obj_range=(float(u.wrong_obj_range) if hasattr(u, 'wrong_obj_range') else None)))
    return dict(rows=rows)


def step_trace(env_id, seed, lens, test_eps=()):
    env = H.make(env_id)
    u = env.unwrapped
    log = []
    attach_scripts(env, seed, log)
    np.random.seed(seed)
    env.seed(seed)
    arng = np.random.RandomState(seed + 99)
    E, Tmax = len(lens), max(lens)
    dimg = len(env.reset()['desired_goal'])
    out = dict(
        lens=np.asarray(lens, np.int32),
        contacts=-np.ones((E, Tmax, MAXCON, 2), np.int16),
        ncon=np.zeros((E, Tmax), np.int32),
        model_idx=np.zeros(E, np.int32),
        num_objs=np.zeros(E, np.int32),
        test_mode=np.zeros(E, np.uint8),
        ag_reset=np.zeros((E, dimg)), g=np.zeros((E, dimg)),
        ag=np.zeros((E, Tmax, dimg)),
        reward_bits=np.zeros((E, Tmax), np.uint32),
        success=np.zeros((E, Tmax), np.uint8),
        done=np.zeros((E, Tmax), np.uint8),
        actions=np.zeros((E, Tmax, 4), np.float32),
    )
    sims = getattr(u, 'sims', None)
    for e, L in enumerate(lens):
        obs = env.reset()
        if e in test_eps:
            obs = u.set_test()
            out['test_mode'][e] = 1
        out['ag_reset'][e] = obs['achieved_goal']
        out['g'][e] = obs['desired_goal']
        out['num_objs'][e] = u.num_objs
        out['model_idx'][e] = sims.index(u.sim) if sims else 0
        for t in range(L):
            a = arng.uniform(-1.5, 1.5, size=4).astype(np.float32)  # exercises the clip
            n0 = len(log)
            o, r, d, info = env.step(a)
            assert len(log) == n0 + 1
            out['actions'][e, t] = a
            out['contacts'][e, t] = pad_contacts(log[-1])
            out['ncon'][e, t] = len(log[-1])
            out['ag'][e, t] = o['achieved_goal']
            assert (o['desired_goal'] == out['g'][e]).all()
            assert r.dtype == np.float32
            out['reward_bits'][e, t] = f32bits(r)
            out['success'][e, t] = bool(info['is_success'])
            out['done'][e, t] = bool(d)
    return out


def obs_trace(env_id, seed, n_eps=3, n_steps=6):
    """Reference _get_obs assembly (fetch_env.py:187-228, Variation :567-621) on seeded random
    site / robot values.  Records the raw inputs the reference read and the observation."""
    env = H.make(env_id)
    u = env.unwrapped
    log = []
    attach_scripts(env, seed, log)
    sims = getattr(u, 'sims', None) or [u.sim]
    rich = np.random.RandomState(seed + 7)
    for sim in sims + [u.sim]:
        sim.data.rich = rich
    np.random.seed(seed)
    env.seed(seed)
    NB = 4 if 'Variation' in env_id else u.num_objs - 2
    rows = dict(grip_pos=[], grip_velp=[], gripper_state=[], gripper_vel=[], obj_pos=[], obj_rot=[],
                obj_velp=[], obj_velr=[], num_objs=[], obs=[])

    def record(obs):
        d = u.sim.data
        nb = u.num_objs - 2
        rows['grip_pos'].append(d.sites['robot0:grip']['xpos'])
        rows['grip_velp'].append(d.sites['robot0:grip']['xvelp'])
        rows['gripper_state'].append(d.robot[0][-2:])
        rows['gripper_vel'].append(d.robot[1][-2:])
        z = np.zeros((NB, 3))
        pos, rot, velp, velr = z.copy(), z.copy(), z.copy(), z.copy()
        for i in range(nb):
            st = d.sites['object%d' % i]
            pos[i], rot[i], velp[i], velr[i] = st['xpos'], st['xmat'][0], st['xvelp'], st['xvelr']
        rows['obj_pos'].append(pos); rows['obj_rot'].append(rot)
        rows['obj_velp'].append(velp); rows['obj_velr'].append(velr)
        rows['num_objs'].append(u.num_objs)
        rows['obs'].append(obs['observation'])

    arng = np.random.RandomState(seed)
    acts, ctrl = [], []
    for _ in range(n_eps):
        record(env.reset())
        for _ in range(n_steps):
            a = arng.uniform(-1.6, 1.6, size=4).astype(np.float32)     # beyond the clip range
            del H.CTRL_LOG[:]
            o, _, _, _ = env.step(a)
            record(o)
            acts.append(a)
            assert len(H.CTRL_LOG) == 1 and H.CTRL_LOG[0].dtype == np.float64
            ctrl.append(H.CTRL_LOG[0])
    out = {k: np.asarray(v) for k, v in rows.items()}
    out['actions'] = np.asarray(acts)              # float32 (n, 4) as a policy hands them over
    out['set_action'] = np.asarray(ctrl)           # float64 (n, 9): _set_action's vector (fetch_env.py:170-185)
    out['block_gripper'] = np.bool_(u.block_gripper)
    out['dt'] = np.float64(u.sim.nsubsteps * u.sim.model.opt.timestep)
    return out


def placement_trace(env_id, seed, n_resets=40):
    """Reference _randomize_objects (through the real reset()/set_test()/increase_difficulty())
    with both RNG streams wrapped so every consumed standard variate is recorded."""
    env = H.make(env_id)
    u = env.unwrapped
    E = H.RecordingRandom(seed)
    Gs = H.RecordingRandom(seed + 1)
    u.np_random = E
    saved = (np.random.normal, np.random.uniform)
    np.random.normal, np.random.uniform = Gs.normal, Gs.uniform      # the global stream (quirk Q7)
    rows = dict(grip_xy=[], eu=[], gz=[], gu=[], xy=[], num_objs=[], test=[], obj_range=[], max_obj_range=[],
                wrong_obj_range=[], difficulty=[], challenge=[])
    nbmax = 4 if 'Variation' in env_id else u.num_objs - 2
    try:
        for k in range(n_resets):
            E.clear(); Gs.clear()
            test = False
            if hasattr(u, 'challenge'):
                u.challenge = (k % 11 == 10)
            # the gripper start position is a physics output; move it next to the table edge
            # on some episodes so the out-of-table rejection branches are exercised
            base = H.GRIP_XPOS0[:2]
            u.initial_gripper_xpos = np.array(
                [base[0] + [0, 0.2, -0.21][k % 3], base[1] + [0, 0.3, -0.31][k % 3], H.GRIP_XPOS0[2]])
            env.reset()
            if k % 4 == 3 and variants_can_test(u):
                # set_test re-randomises at maximum difficulty (fetch_env.py:365-368, :443-446)
                E.clear(); Gs.clear()
                u.set_test()
                test = env_id != 'BlocksTouchVariation-v0'     # Variation's set_test does not re-place
                if not test:
                    continue
            xy = np.full((nbmax, 2), np.nan)
            for i in range(u.num_objs - 2):
                xy[i] = u.sim.data.qpos['object%d:joint' % i][:2]
            rows['eu'].append(list(E.u)); rows['gz'].append(list(Gs.z)); rows['gu'].append(list(Gs.u))
            rows['xy'].append(xy); rows['num_objs'].append(u.num_objs); rows['test'].append(test)
            rows['grip_xy'].append(u.initial_gripper_xpos[:2].copy())
            rows['obj_range'].append(u.obj_range)
            rows['max_obj_range'].append(getattr(u, 'max_obj_range', np.nan))
            rows['wrong_obj_range'].append(getattr(u, 'wrong_obj_range', np.nan))
            rows['difficulty'].append(u.difficulty)
            rows['challenge'].append(bool(getattr(u, 'challenge', False)))
            if k % 5 == 4:
                try:
                    u.increase_difficulty()
                except (NotImplementedError, AttributeError):
                    pass
    finally:
        np.random.normal, np.random.uniform = saved

    def pad(lists):
        m = max(1, max(len(x) for x in lists))
        out = np.full((len(lists), m), np.nan)
        for i, x in enumerate(lists):
            out[i, :len(x)] = x
        return out, np.asarray([len(x) for x in lists], np.int32)
    out = {}
    for k in ('eu', 'gz', 'gu'):
        out[k], out['n_' + k] = pad(rows[k])
    for k in ('grip_xy', 'xy', 'num_objs', 'test', 'obj_range', 'max_obj_range', 'wrong_obj_range', 'difficulty', 'challenge'):
        out[k] = np.asarray(rows[k])
    return out


def variants_can_test(u):
    try:
        type(u).set_test
    except AttributeError:
        return False
    import gym_blocks.envs.fetch_env as fe
    return type(u).set_test is not fe.BlocksEnv.set_test


def compute_reward_cases(rng):
    """The reference's vectorised compute_reward on ternary inputs
    (config.py:110-111 call shape: (batch, N^2) in, (batch,) float32 out)."""
    out = {}
    for env_id in ['GripperTouch-v0', 'BlocksTouch-v0', 'BlocksTouchChoose-v0',
                   'ToppleTower-v0', 'BlocksTouchVariation-v0']:
        env = H.make(env_id)
        g = np.asarray(env.reset()['desired_goal'])
        n2 = g.shape[0]
        key = env_id.split('-')[0]
        ag = rng.randint(-1, 2, size=(257, n2)).astype(np.float32)
        # make a good share of rows succeed
        nz = g != 0
        ag[::3][:, nz] = g[nz]
        out[key + '_ag'] = ag
        out[key + '_g'] = np.broadcast_to(g, ag.shape).astype(np.float32).copy()
        r = env.compute_reward(ag, out[key + '_g'], None)
        assert r.dtype == np.float32 and r.shape == (257,)
        out[key + '_rbits'] = f32bits(r)
        # random ternary goals too (HER relabels g with achieved goals)
        g2 = rng.randint(-1, 2, size=(257, n2)).astype(np.float32)
        g2[::4] = ag[::4] * (rng.rand(*ag[::4].shape) < 0.3)
        out[key + '_g2'] = g2
        out[key + '_rbits2'] = f32bits(env.compute_reward(ag, g2, None))
        # extra leading axis + float64 inputs
        ag3 = ag[:240].reshape(4, 60, n2).astype(np.float64)
        g3 = g2[:240].reshape(4, 60, n2).astype(np.float64)
        r3 = env.compute_reward(ag3, g3, None)
        assert r3.shape == (4, 60)
        out[key + '_rbits3'] = f32bits(r3)
    return out


def compute_reward_order_cases():
    """The reference's compute_reward (fetch_env.py:135-143) on rows that are NOT ternary, float32 and
    float64: np.sum's pairwise order then decides whether d == c.  Every row's exact sum equals
    count_nonzero(g); large cancelling products at random positions make a sequential sum, NumPy's pairwise
    sum and the exact sum disagree on a good share of the rows.  (The env and HER only ever produce ternary
    rows, for which any order is exact; this fixture pins the order for everything else.)"""
    rng = np.random.RandomState(314159)
    env = H.make('BlocksTouchVariation-v0')
    env.reset()
    out = {}
    for dt, big_exp in ((np.float32, (24, 28)), (np.float64, (53, 57))):
        for L in (7, 16, 36, 100, 129, 159):
            n = 150
            g = rng.choice(np.array([0, 0, 1, -1, 2, -2, 0.5, -0.5]), size=(n, L))
            prod = np.zeros((n, L))
            for r in range(n):
                nz = np.flatnonzero(g[r])
                c = len(nz)
                if c < 4:
                    g[r, :4] = [1, -1, 2, 0.5]
                    nz = np.flatnonzero(g[r])
                    c = len(nz)
                k = int(rng.randint(1, max(2, min(4, c // 2))))            # cancelling pairs
                pos = rng.permutation(nz)
                for j in range(k):
                    b = float(2.0 ** rng.randint(*big_exp)) * rng.choice([1.0, 3.0])
                    prod[r, pos[2 * j]], prod[r, pos[2 * j + 1]] = b, -b
                rest = pos[2 * k:]
                if len(rest):
                    # small exact values that add up to c (multiples of 1/4)
                    v = rng.randint(-8, 9, size=len(rest)) / 4.0
                    v[-1] += c - v.sum()
                    prod[r, rest] = v
                else:
                    prod[r, pos[0]] += c
            ag = np.where(g != 0, prod / np.where(g != 0, g, 1.0), rng.normal(size=(n, L)))
            ag, gg = ag.astype(dt), g.astype(dt)
            assert np.array_equal((ag.astype(np.float64) * gg.astype(np.float64))[g != 0], prod[g != 0])   # exact products
            key = '%s_%d' % (np.dtype(dt).name, L)
            out[key + '_ag'], out[key + '_g'] = ag, gg
            r = env.compute_reward(ag, gg, None)
            assert r.dtype == np.float32
            out[key + '_rbits'] = f32bits(r)
            seq = np.zeros(n, dt)
            for kk in range(L):
                seq = (seq + ag[:, kk] * gg[:, kk]).astype(dt)
            c_all = np.count_nonzero(gg, axis=-1)
            out[key + '_differs_from_sequential'] = np.array(int(((seq != c_all) != (r != 0)).sum()))
    return out


class RandomPolicy:
    """Seeded random-action policy with the call signature rollout.py:92-97 uses."""

    def __init__(self, seed, log):
        self.rng = np.random.RandomState(seed)
        self.log = log
        self.kwargs = {}
        self.qlog = []

    def get_actions(self, o, ag, g, **kw):
        u = self.rng.uniform(-1, 1, size=(o.shape[0], 4)).astype(np.float32)
        self.log.append(u)
        if 'exploit' in kw:      # policy_gradient/rollout.py:202-204
            return u, u.copy(), np.zeros_like(u)
        if kw.get('compute_Q'):  # rollout.py:99-103: (u, Q)
            Q = self.rng.standard_normal((o.shape[0], 1)).astype(np.float32)
            self.qlog.append(Q)
            return u, Q
        return u


class NullLogger:
    def warning(self, *a):
        pass
    info = warning


def rollout_trace(env_id, seed, B, modes, pg=False, compute_Q=False):
    """Reference RolloutStudent.generate_rollouts (rollout.py:75-172 or the
    policy_gradient variant) over B scripted envs."""
    H.install()
    if pg:
        from gym_blocks.policy_gradient.rollout import RolloutStudent
    else:
        from gym_blocks.rollout import RolloutStudent
    logs = []

    def make_env():
        env = H.make(env_id)
        log = []
        attach_scripts(env, seed + 1000 * len(logs), log, p_pair=0.012)
        # RolloutStudent.__init__ resets every env BEFORE worker.seed() runs (rollout.py:33-45):
        # with the per-env stream still seeded from OS entropy the first placement -- and through its
        # rejection loops the position of the GLOBAL np.random stream (quirk Q7) -- differed from run
        # to run, and with it every later cube position (the `o` arrays).  Seed at construction.
        env.seed(seed + 1000 * len(logs))
        logs.append(log)
        return env

    np.random.seed(seed)
    probe = H.make(env_id)
    probe.seed(seed + 999)        # see expert_rollout_trace: an unseeded probe can move the global stream
    o = probe.reset()
    dims = dict(o=o['observation'].shape[0], u=4, g=o['desired_goal'].shape[0],
                info_is_success=1)
    T = probe._max_episode_steps
    ulog = []
    policy = RandomPolicy(seed, ulog)
    kw = dict(make_env=make_env, policy=policy, dims=dims, logger=NullLogger(), T=T,
              rollout_batch_size=B)
    if pg:
        kw.update(expert=None, gamma=1. - 1. / T)
    if compute_Q:
        kw.update(compute_Q=True)
    worker = RolloutStudent(**kw)
    worker.seed(seed)
    out = dict(T=np.int32(T), B=np.int32(B), modes=np.asarray(modes, np.uint8))
    for r, test in enumerate(modes):
        for log in logs:
            del log[:]
        del ulog[:]
        del policy.qlog[:]
        ep = worker.generate_rollouts(test=bool(test))
        for k, v in ep.items():
            out['r%d_%s' % (r, k)] = v
        if compute_Q:            # rollout.py:168-169, :182-184, :197-198
            out['r%d_Q' % r] = np.asarray(policy.qlog)                       # (T, B, 1) float32, as the policy returned them
            out['r%d_Q_history_last' % r] = np.asarray(worker.Q_history[-1])  # np.mean(Qs): a float32 scalar
            out['r%d_current_mean_Q' % r] = np.asarray(worker.current_mean_Q())
        con = -np.ones((B, T, MAXCON, 2), np.int16)
        ncon = np.zeros((B, T), np.int32)
        for b, log in enumerate(logs):
            assert len(log) == T, (len(log), T)
            for t, pairs in enumerate(log):
                con[b, t] = pad_contacts(pairs)
                ncon[b, t] = len(pairs)
        out['r%d_contacts' % r] = con
        out['r%d_ncon' % r] = ncon
        out['r%d_num_objs' % r] = np.asarray([e.unwrapped.num_objs for e in worker.envs], np.int32)
        sims = getattr(worker.envs[0].unwrapped, 'sims', None)
        out['r%d_model_idx' % r] = np.asarray(
            [e.unwrapped.sims.index(e.unwrapped.sim) if sims else 0 for e in worker.envs], np.int32)
        out['r%d_success_rate' % r] = np.float64(worker.success_history[-1])
    out['n_episodes'] = np.int32(worker.n_episodes)
    return out


def restart_trace(env_id, seed, B, faults, pg=False, exploit_first=False):
    """What the reference's RolloutStudent does when a rollout goes wrong (rollout.py:136-142; policy_gradient/rollout.py
    :237-246): `faults` is one entry per generate_rollouts call -- None (a normal rollout), ('nan', env j, step k): env
    j's observation of step k carries a NaN (FakeSim.fault), or ('raise', j, k): env j's sim.step() of step k raises
    MujocoException.  Recorded per call: the episode the reference finally returns, the contact lists / actions /
    num_objs of the DISCARDED attempt as well (with quirk Q1 the base variants keep what the discarded steps touched,
    fetch_env.py:247-255), how many steps every env took in it, the number of resets, and the bookkeeping afterwards."""
    H.install()
    if pg:
        from gym_blocks.policy_gradient.rollout import RolloutStudent
    else:
        from gym_blocks.rollout import RolloutStudent
    logs, envs, fault_dicts = [], [], []

    def make_env():
        env = H.make(env_id)
        log = []
        attach_scripts(env, seed + 1000 * len(logs), log, p_pair=0.05)
        env.seed(seed + 1000 * len(logs))
        fd = {'calls': 0}
        u = env.unwrapped
        for sim in set((getattr(u, 'sims', None) or []) + [u.sim]):
            sim.fault = fd
        logs.append(log)
        envs.append(env)
        fault_dicts.append(fd)
        return env

    np.random.seed(seed)
    probe = H.make(env_id)
    probe.seed(seed + 999)
    o = probe.reset()
    dims = dict(o=o['observation'].shape[0], u=4, g=o['desired_goal'].shape[0], info_is_success=1)
    T = probe._max_episode_steps
    ulog = []
    policy = RandomPolicy(seed, ulog)
    policy.kwlog = []
    inner = policy.get_actions

    def get_actions(o, ag, g, **kw):
        policy.kwlog.append(dict(kw))
        return inner(o, ag, g, **kw)
    policy.get_actions = get_actions
    kw = dict(make_env=make_env, policy=policy, dims=dims, logger=NullLogger(), T=T, rollout_batch_size=B)
    if pg:
        kw.update(expert=None, gamma=1. - 1. / T)
    worker = RolloutStudent(**kw)
    worker.seed(seed)
    out = dict(T=np.int32(T), B=np.int32(B), n_calls=np.int32(len(faults)),
               faults=np.asarray([[0, -1, -1] if f is None else [1 if f[0] == 'nan' else 2, f[1], f[2]] for f in faults], np.int32))
    for r, fault in enumerate(faults):
        for log in logs:
            del log[:]
        del ulog[:]
        del policy.kwlog[:]
        for e in envs:
            del e.reset_log[:]
        for fd in fault_dicts:
            fd.clear()
            fd['calls'] = 0
        if fault is not None:
            fault_dicts[fault[1]]['nan_at' if fault[0] == 'nan' else 'raise_at'] = fault[2] + 1
        n_before, hist_before = worker.n_episodes, len(worker.success_history)
        if pg:
            ep = worker.generate_rollouts(exploit=bool(exploit_first))
        else:
            ep = worker.generate_rollouts()
        for k, v in ep.items():
            out['r%d_%s' % (r, k)] = v
        # steps of the discarded attempt per env = everything the env's script emitted before the last T lists
        nd = np.asarray([len(log) - T for log in logs], np.int32)
        assert (nd >= 0).all() and (fault is not None or (nd == 0).all())
        out['r%d_discard_steps' % r] = nd
        D = int(nd.max()) if len(nd) else 0
        con = -np.ones((B, T, MAXCON, 2), np.int16)
        ncon = np.zeros((B, T), np.int32)
        dcon = -np.ones((B, D, MAXCON, 2), np.int16)
        dncon = np.zeros((B, D), np.int32)
        for b, log in enumerate(logs):
            for t, pairs in enumerate(log[:nd[b]]):
                dcon[b, t] = pad_contacts(pairs)
                dncon[b, t] = len(pairs)
            for t, pairs in enumerate(log[nd[b]:]):
                con[b, t] = pad_contacts(pairs)
                ncon[b, t] = len(pairs)
        out['r%d_contacts' % r], out['r%d_ncon' % r] = con, ncon
        out['r%d_discard_contacts' % r], out['r%d_discard_ncon' % r] = dcon, dncon
        # the policy was asked once per step of either attempt; the last T calls belong to the returned episode
        nu = len(ulog) - T
        out['r%d_discard_u' % r] = np.asarray(ulog[:nu], np.float32).reshape((nu, B, 4))
        out['r%d_policy_calls' % r] = np.int32(len(ulog))
        if pg:      # the retry is `self.generate_rollouts(render, test)`: `exploit` falls back to its default
            out['r%d_exploit_seen' % r] = np.asarray([bool(k.get('exploit')) for k in policy.kwlog], np.uint8)
        # num_objs after every reset of this call, per env: (n_resets, B)
        nres = {len(e.reset_log) for e in envs}
        assert len(nres) == 1
        out['r%d_reset_num_objs' % r] = np.asarray([e.reset_log for e in envs], np.int32).T
        out['r%d_num_objs' % r] = np.asarray([e.unwrapped.num_objs for e in worker.envs], np.int32)
        sims = getattr(worker.envs[0].unwrapped, 'sims', None)
        out['r%d_model_idx' % r] = np.asarray(
            [e.unwrapped.sims.index(e.unwrapped.sim) if sims else 0 for e in worker.envs], np.int32)
        out['r%d_success_rate' % r] = np.float64(worker.success_history[-1])
        out['r%d_n_episodes_added' % r] = np.int32(worker.n_episodes - n_before)
        out['r%d_history_added' % r] = np.int32(len(worker.success_history) - hist_before)
        out['r%d_final_ag' % r] = np.asarray([e.unwrapped.achieved_goal.ravel() for e in worker.envs], np.float64)
        out['r%d_final_has_succeeded' % r] = np.asarray([bool(e.unwrapped.has_succeeded) for e in worker.envs], np.uint8)
    out['n_episodes'] = np.int32(worker.n_episodes)
    return out


def _worker_inputs(env_id, B, seed, steps_of):
    """(o, g, ag) as the pg worker holds them (policy_gradient/rollout.py:64-66, :188-191: float32 arrays filled
    from the env's float64 observation dicts), env b advanced steps_of(b) scripted steps after its reset."""
    o = g = ag = None
    for b in range(B):
        env = H.make(env_id)
        attach_scripts(env, seed + 1000 * b, [], p_pair=0.1)
        env.seed(seed + 1000 * b)
        obs = env.reset()
        rng = np.random.RandomState(seed + b)
        for _ in range(steps_of(b)):
            obs = env.step(rng.uniform(-1, 1, 4).astype(np.float32))[0]
        if o is None:
            o = np.empty((B, obs['observation'].shape[0]), np.float32)
            g = np.empty((B, obs['desired_goal'].shape[0]), np.float32)
            ag = np.empty_like(g)
        o[b], g[b], ag[b] = obs['observation'], obs['desired_goal'], obs['achieved_goal']
    return o, g, ag


def trim_cases():
    """RolloutStudent.trim of the policy-gradient worker (policy_gradient/rollout.py:105-171, batched branch): what
    the expert of a smaller puzzle is handed.  Inputs are observations of the reference envs; expected outputs are
    the reference's, dtype included (the Variation branch concatenates a float64 array and returns float64)."""
    import types
    H.install()
    from gym_blocks.policy_gradient.rollout import RolloutStudent
    out = {}
    cases = [('variation', 'BlocksTouchVariation-v0', 40, 16, 4, 12),   # 4 slots, 2 coloured -> BlocksTouch's dims
             ('choose', 'BlocksTouchChoose-v0', 40, 16, 4, 6),          # plain prefix of o, 4 x 4 corner of the 5 x 5 goal
             ('tower', 'ToppleTower-v0', 25, 9, 3, 5),                  # another num_objs
             ('same', 'BlocksTouch-v0', 40, 16, 4, 3)]                  # nothing to trim: the arrays themselves
    for name, env_id, dimo, dimg, num_objs, B in cases:
        o, g, ag = _worker_inputs(env_id, B, 4100 + len(out), lambda b: b % 4)
        fake = types.SimpleNamespace(kwargs={'info': {'env_name': env_id}})
        o_, g_, ag_ = RolloutStudent.trim(fake, o, g, ag, dimo, dimg, num_objs=num_objs)
        out.update({name + '_o': o, name + '_g': g, name + '_ag': ag, name + '_dims': np.asarray([dimo, dimg, num_objs], np.int32),
                    name + '_o_out': np.asarray(o_), name + '_g_out': np.asarray(g_), name + '_ag_out': np.asarray(ag_),
                    name + '_same_objects': np.asarray([o_ is o, g_ is g, ag_ is ag])})
    return out


class RecordingExpert:
    """The expert of policy_gradient/rollout.py:198-200: sees the TRIMMED inputs, returns (u, raw)."""

    def __init__(self, seed, dimo, dimg, log):
        self.rng, self.dimo, self.dimg, self.log = np.random.RandomState(seed), dimo, dimg, log

    def get_actions(self, o, ag, g, compute_raw=False):
        assert compute_raw
        u = self.rng.uniform(-1, 1, size=(o.shape[0], 4)).astype(np.float32)
        self.log.append(dict(o=np.array(o), ag=np.array(ag), g=np.array(g), u=u))
        return u, (u * np.float32(0.5))


def expert_rollout_trace(env_id, seed, B, dimo, dimg, n_rollouts=2):
    """policy_gradient generate_rollouts with an expert in charge on some steps (np.random.rand() < beta,
    :196-206).  The coin flips come from the GLOBAL NumPy stream, which the reference's placement loops also
    consume (quirk Q7), so the fixture records which steps the expert took."""
    H.install()
    from gym_blocks.policy_gradient.rollout import RolloutStudent
    logs = []

    def make_env():
        env = H.make(env_id)
        log = []
        attach_scripts(env, seed + 1000 * len(logs), log, p_pair=0.012)
        env.seed(seed + 1000 * len(logs))
        logs.append(log)
        return env

    np.random.seed(seed)
    probe = H.make(env_id)
    probe.seed(seed + 999)        # an unseeded probe draws from OS entropy and, through its placement loops, moves the GLOBAL stream (Q7)
    o = probe.reset()
    dims = dict(o=o['observation'].shape[0], u=4, g=o['desired_goal'].shape[0], info_is_success=1)
    T = probe._max_episode_steps
    ulog, xlog = [], []
    policy = RandomPolicy(seed, ulog)
    policy.kwargs = {'info': {'env_name': env_id}}
    worker = RolloutStudent(make_env=make_env, policy=policy, expert=RecordingExpert(seed + 1, dimo, dimg, xlog), dims=dims,
                            logger=NullLogger(), T=T, rollout_batch_size=B, gamma=1. - 1. / T, beta_final=0.35,
                            annealing_coeff=2.0)
    worker.seed(seed)
    out = dict(T=np.int32(T), B=np.int32(B), n_rollouts=np.int32(n_rollouts), expert_dims=np.asarray([dimo, dimg], np.int32))
    for r in range(n_rollouts):
        for log in logs:
            del log[:]
        del ulog[:], xlog[:]
        out['r%d_beta' % r] = np.float64(worker.beta())
        ep = worker.generate_rollouts()
        worker.anneal()
        for k, v in ep.items():
            out['r%d_%s' % (r, k)] = v
        # which steps the expert took: its raw actions are u / 2, the policy's are u
        con = -np.ones((B, T, MAXCON, 2), np.int16)
        ncon = np.zeros((B, T), np.int32)
        for b, log in enumerate(logs):
            for t, pairs in enumerate(log):
                con[b, t] = pad_contacts(pairs)
                ncon[b, t] = len(pairs)
        out['r%d_contacts' % r], out['r%d_ncon' % r] = con, ncon
        out['r%d_num_objs' % r] = np.asarray([e.unwrapped.num_objs for e in worker.envs], np.int32)
        in_charge = np.zeros(T, np.uint8)
        xi = pi = 0
        for t in range(T):
            raw_t = ep['u'][:, t]
            if xi < len(xlog) and np.array_equal(raw_t, xlog[xi]['u'] * np.float32(0.5)):
                in_charge[t] = 1
                xi += 1
            else:
                assert np.array_equal(raw_t, ulog[pi])
                pi += 1
        assert xi == len(xlog) and pi == len(ulog)
        out['r%d_expert_in_charge' % r] = in_charge
        out['r%d_policy_u' % r] = np.asarray(ulog, np.float32).reshape(len(ulog), B, 4)
        out['r%d_expert_u' % r] = np.asarray([x['u'] for x in xlog], np.float32).reshape(len(xlog), B, 4)
        for key in ('o', 'ag', 'g'):
            arr = np.asarray([x[key] for x in xlog])
            out['r%d_expert_saw_%s' % (r, key)] = arr.reshape((len(xlog), B, -1))
        out['r%d_success_rate' % r] = np.float64(worker.success_history[-1])
    return out


def main(HERE=HERE):
    H.install()
    from gym_blocks.envs import fetch_env
    rng = np.random.RandomState(20261004)

    registry = [dict(id=r['id'], entry_point=r['entry_point'], kwargs=r.get('kwargs', {}),
                     max_episode_steps=r['max_episode_steps']) for r in H.REGISTRY]
    variants = {eid: variant_info(eid) for eid in ENV_IDS}
    for eid in ENV_IDS:
        variants[eid]['num_objs'] = int(H.make(eid).unwrapped.num_objs)
        variants[eid]['difficulty'] = difficulty_trace(eid)

    pts = rng.uniform([0.9, 0.3], [1.7, 1.2], size=(200, 2))
    # points exactly on / next to the table edge
    edge = [[fetch_env.TABLE_X + s * fetch_env.TABLE_W, fetch_env.TABLE_Y] for s in (-1, 1)]
    edge += [[fetch_env.TABLE_X, fetch_env.TABLE_Y + s * fetch_env.TABLE_H] for s in (-1, 1)]
    edge += [[np.nextafter(e[0], 9), np.nextafter(e[1], 9)] for e in list(edge)]
    pts = np.concatenate([pts, np.asarray(edge)])
    misc = dict(
        constants=dict(GREY=fetch_env.GREY, RED=fetch_env.RED, GREEN=fetch_env.GREEN,
                       BLUE=fetch_env.BLUE, NUM_COLORS=fetch_env.NUM_COLORS,
                       BLOCK_SIZE=fetch_env.BLOCK_SIZE, MIN_BLOCK_DIST=fetch_env.MIN_BLOCK_DIST,
                       TABLE_W=fetch_env.TABLE_W, TABLE_H=fetch_env.TABLE_H,
                       TABLE_X=fetch_env.TABLE_X, TABLE_Y=fetch_env.TABLE_Y),
        out_of_table=dict(points=pts.tolist(),
                          expected=[bool(fetch_env.out_of_table(p)) for p in pts]),
        one_hot_color=[fetch_env.one_hot_color(c).tolist() for c in range(4)],
    )
    with open(os.path.join(HERE, 'registry.json'), 'w') as f:
        json.dump(registry, f, indent=1)
    with open(os.path.join(HERE, 'variants.json'), 'w') as f:
        json.dump(variants, f, indent=1)
    with open(os.path.join(HERE, 'misc.json'), 'w') as f:
        json.dump(misc, f, indent=1)

    lens = [50, 50, 7, 50, 1, 60, 50, 23]
    for k, eid in enumerate(ENV_IDS):
        can_test = variants[eid]['set_test'] == 'ok'
        tr = step_trace(eid, seed=100 + k, lens=lens, test_eps=(3, 6) if can_test else ())
        np.savez_compressed(os.path.join(HERE, 'trace_%s.npz' % eid.split('-')[0]), **tr)

    np.savez_compressed(os.path.join(HERE, 'compute_reward.npz'), **compute_reward_cases(rng))
    np.savez_compressed(os.path.join(HERE, 'compute_reward_order.npz'), **compute_reward_order_cases())
    for k, eid in enumerate(ENV_IDS):
        np.savez_compressed(os.path.join(HERE, 'obs_%s.npz' % eid.split('-')[0]), **obs_trace(eid, 300 + k))
    for k, eid in enumerate(ENV_IDS):
        np.savez_compressed(os.path.join(HERE, 'placement_%s.npz' % eid.split('-')[0]),
                            **placement_trace(eid, 900 + k))

    for k, eid in enumerate(ENV_IDS):
        can_test = variants[eid]['set_test'] == 'ok'
        modes = [0, 0, 1, 0] if can_test else [0, 0, 0]
        ro = rollout_trace(eid, seed=500 + k, B=5, modes=modes)
        np.savez_compressed(os.path.join(HERE, 'rollout_%s.npz' % eid.split('-')[0]), **ro)
    for k, eid in enumerate(['BlocksTouch-v0', 'BlocksTouchVariation-v0', 'ToppleTower-v0', 'GripperTouch-v0',
                             'BlocksTouchChoose-v0']):
        ro = rollout_trace(eid, seed=700 + k, B=4, modes=[0, 0], pg=True)
        np.savez_compressed(os.path.join(HERE, 'pg_rollout_%s.npz' % eid.split('-')[0]), **ro)
    np.savez_compressed(os.path.join(HERE, 'trim.npz'), **trim_cases())
    np.savez_compressed(os.path.join(HERE, 'pg_expert_rollout_BlocksTouchVariation.npz'),
                        **expert_rollout_trace('BlocksTouchVariation-v0', seed=810, B=4, dimo=40, dimg=16))
    np.savez_compressed(os.path.join(HERE, 'pg_expert_rollout_BlocksTouchChoose.npz'),
                        **expert_rollout_trace('BlocksTouchChoose-v0', seed=811, B=3, dimo=40, dimg=16))
    np.savez_compressed(os.path.join(HERE, 'rollout_Q_BlocksTouch.npz'),
                        **rollout_trace('BlocksTouch-v0', seed=860, B=5, modes=[0, 0, 0], compute_Q=True))
    # restarts (rollout.py:136-142, policy_gradient/rollout.py:243-246): a base variant (quirk Q1: the matrix survives
    # resets, so what the discarded steps touched shows in the retried episode) and Variation (three num_grey draws)
    nan_raise = [None, ('nan', 2, 7), None, ('raise', 3, 4), ('nan', 0, 0), ('raise', 0, 49), None]
    np.savez_compressed(os.path.join(HERE, 'restart_BlocksTouch.npz'), **restart_trace('BlocksTouch-v0', 910, 5, nan_raise))
    np.savez_compressed(os.path.join(HERE, 'restart_BlocksTouchVariation.npz'),
                        **restart_trace('BlocksTouchVariation-v0', 911, 5, nan_raise))
    np.savez_compressed(os.path.join(HERE, 'restart_ToppleTower.npz'),
                        **restart_trace('ToppleTower-v0', 912, 4, [None, ('nan', 3, 49), ('raise', 1, 0), None]))
    np.savez_compressed(os.path.join(HERE, 'restart_BlocksTouchChooseCurriculum.npz'),
                        **restart_trace('BlocksTouchChooseCurriculum-v0', 913, 3, [('raise', 2, 10), None, ('nan', 1, 25), None]))
    np.savez_compressed(os.path.join(HERE, 'restart_GripperTouch.npz'),
                        **restart_trace('GripperTouch-v0', 914, 4, [('nan', 3, 3), ('nan', 0, 48), ('raise', 3, 49), None]))
    nan_only = [None, ('nan', 1, 12), None, ('nan', 3, 0), None]
    np.savez_compressed(os.path.join(HERE, 'pg_restart_BlocksTouch.npz'),
                        **restart_trace('BlocksTouch-v0', 920, 4, nan_only, pg=True, exploit_first=True))
    np.savez_compressed(os.path.join(HERE, 'pg_restart_BlocksTouchVariation.npz'),
                        **restart_trace('BlocksTouchVariation-v0', 921, 4, nan_only, pg=True))
    print('golden fixtures written to', HERE)


def check():
    """Regenerates every fixture into a temporary directory and compares it with the committed
    one (arrays bit for bit, json after parsing).  Exit code 1 on any difference."""
    import tempfile
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        main(tmp)
        for name in sorted(os.listdir(tmp)):
            new, old = os.path.join(tmp, name), os.path.join(HERE, name)
            if not os.path.exists(old):
                bad.append('%s: not committed' % name)
            elif name.endswith('.json'):
                if json.load(open(new)) != json.load(open(old)):
                    bad.append('%s: differs' % name)
            else:
                a, b = np.load(new), np.load(old)
                if sorted(a.files) != sorted(b.files):
                    bad.append('%s: keys differ' % name)
                    continue
                for k in a.files:
                    x, y = a[k], b[k]
                    if x.shape != y.shape or x.dtype != y.dtype or x.tobytes() != y.tobytes():
                        bad.append('%s[%s]: differs' % (name, k))
    for line in bad:
        print('MISMATCH', line)
    print('checked: %s' % ('%d mismatches' % len(bad) if bad else 'every fixture regenerates bit-identically'))
    return 1 if bad else 0


if __name__ == '__main__':
    if '--check' in sys.argv[1:]:
        sys.exit(check())
    main()
