"""Stub-import harness used ONLY by tests/golden/make_golden.py (build container).

Purpose: exercise the reference-owned goal logic of /root/reference (gym_blocks)
without gym / mujoco_py / baselines / tensorflow, none of which exist in this
image.  Nothing in here is a restatement of the
reference: it only provides *inert stand-ins for the third-party modules the
reference imports* plus a scripted "physics" object whose contact list is read
by the reference's own `_step_callback` (fetch_env.py:148-167).

The reference itself is imported from /root/reference at generation time; it
never travels to the GPU box.  Only the *data* written by make_golden.py does.

Stand-ins and what they stand for (all third-party, parity unpinned):
  gym.GoalEnv / spaces.Box / spaces.Dict / utils.EzPickle     -> inert classes
  gym.utils.seeding.np_random(seed) -> (np.random.RandomState(seed), seed)
        (the real one hashes the seed first; recorded as an explicit input)
  gym.envs.registration.register    -> records the kwargs
  gym TimeLimit                     -> TimeLimitStandIn below (done at t>=50)
  gym.envs.robotics.utils / rotations -> kinematic no-physics fakes
  mujoco_py.load_model_from_path / MjSim / modder.TextureModder -> FakeSim
  baselines.her.util                -> the reference's own vendored copy
        gym_blocks/util.py, imported with inert tensorflow/baselines stubs
"""
import copy
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

REF_ROOT = '/root/reference'
ASSET_DIR = os.path.join(REF_ROOT, 'gym_blocks', 'envs', 'assets')

# Stand-in for the settled gripper site position MuJoCo would report after
# `_env_setup` (fetch_env.py:300-301).  Physics output, not reference-owned.
GRIP_XPOS0 = np.array([1.3419, 0.7491, 0.555])


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


# --------------------------------------------------------------------------
# XML -> geom name list in document order (includes expanded in place)
# --------------------------------------------------------------------------
def geom_names_from_xml(path):
    """Names (or None) of every <geom> under <worldbody>, document order."""
    root = ET.parse(path).getroot()
    base = os.path.dirname(path)
    names = []

    def walk(elem):
        for child in expand_once(elem):
            if child.tag == 'geom':
                names.append(child.get('name'))
            walk(child)

    def expand_once(elem):
        res = []
        for child in list(elem):
            if child.tag == 'include':
                sub = ET.parse(os.path.join(base, child.get('file'))).getroot()
                res.extend(expand_once(sub))
            else:
                res.append(child)
        return res

    for wb in root.iter('worldbody'):
        walk(wb)
    return names


# --------------------------------------------------------------------------
# Fake MuJoCo
# --------------------------------------------------------------------------
class FakeContact:
    __slots__ = ('geom1', 'geom2')

    def __init__(self, g1=0, g2=0):
        self.geom1, self.geom2 = g1, g2


class FakeModel:
    def __init__(self, path):
        self.path = path
        self.geom_names = geom_names_from_xml(path)
        self.ngeom = len(self.geom_names)
        self.opt = types.SimpleNamespace(timestep=0.002)
        self.nblocks = sum(1 for n in self.geom_names
                           if n is not None and n.startswith('object'))

    def geom_id2name(self, i):
        return self.geom_names[i]

    def body_name2id(self, name):
        return 0


class FakeData:
    MAXCON = 16

    def __init__(self, sim):
        self._sim = sim
        self.ncon = 0
        # stale entries beyond ncon must be ignored by the reference's [:ncon] loop
        self.contact = [FakeContact(1, 1) for _ in range(self.MAXCON)]
        self.qpos = {}
        self.grip = GRIP_XPOS0.copy()
        self.mocap_pos = np.zeros(3)
        self.mocap_quat = np.zeros(4)
        # "rich" mode: every site / robot quantity _get_obs reads is a fresh seeded random
        # value per step, so the reference's observation ASSEMBLY is exercised on
        # non-trivial inputs (the values themselves are physics stand-ins)
        self.rich = None
        self.sites = {}
        self.robot = (np.zeros(15), np.zeros(15))
        self.nan_now = False    # scripted fault (FakeSim.fault): this step's gripper site position is NaN

    def regenerate(self):
        if self.rich is not None:
            self.sites = {}
            self.robot = (self.rich.normal(size=15), self.rich.normal(size=15))

    def _site(self, name):
        if name not in self.sites:
            r = self.rich
            self.sites[name] = dict(xpos=r.uniform(-1, 2, size=3), xmat=r.normal(size=(3, 3)),
                                    xvelp=r.normal(size=3), xvelr=r.normal(size=3))
        return self.sites[name]

    def set_joint_qpos(self, name, value):
        self.qpos[name] = np.array(value, dtype=np.float64).copy()

    def get_joint_qpos(self, name):
        return np.array(self.qpos[name], dtype=np.float64).copy()

    def set_mocap_pos(self, name, v):
        self.mocap_pos = np.array(v, dtype=np.float64)

    def set_mocap_quat(self, name, v):
        self.mocap_quat = np.array(v, dtype=np.float64)

    def get_site_xpos(self, name):
        if self.rich is not None:
            return self._site(name)['xpos'].copy()
        if name == 'robot0:grip':
            if self.nan_now:           # what a diverged simulation reports: the reference's np.isnan(o_new).any()
                bad = self.grip.copy()  # (rollout.py:139) is the only reader that cares
                bad[0] = np.nan
                return bad
            return self.grip
        return np.array(self.qpos[name + ':joint'][:3])

    def get_site_xmat(self, name):
        return self._site(name)['xmat'].copy() if self.rich is not None else np.eye(3)

    def get_site_xvelp(self, name):
        return self._site(name)['xvelp'].copy() if self.rich is not None else np.zeros(3)

    def get_site_xvelr(self, name):
        return self._site(name)['xvelr'].copy() if self.rich is not None else np.zeros(3)


class FakeSim:
    """Scripted stand-in for mujoco_py.MjSim: `step()` installs the next
    scripted contact list; nothing is simulated."""

    def __init__(self, model, nsubsteps=1):
        self.model = model
        self.nsubsteps = nsubsteps
        self.data = FakeData(self)
        self.script = None      # callable() -> list[(g1, g2)]
        self.n_forward = 0
        # scripted faults, shared by every sim of one env: {'calls': n, 'raise_at': k, 'nan_at': k} -- the k-th
        # sim.step() of the env (counted from 1 since the dict was installed) raises MujocoException before anything
        # is simulated (as mujoco_py does from inside mj_step), or reports a NaN gripper position for that step
        self.fault = None

    def step(self):
        f = self.fault
        if f is not None:
            f['calls'] += 1
            if f.get('raise_at') == f['calls']:
                raise sys.modules['mujoco_py'].MujocoException('scripted simulation failure')
            self.data.nan_now = (f.get('nan_at') == f['calls'])
        self.data.regenerate()
        if self.script is not None:
            pairs = self.script()
            assert len(pairs) <= FakeData.MAXCON
            self.data.ncon = len(pairs)
            for k, (a, b) in enumerate(pairs):
                self.data.contact[k].geom1 = int(a)
                self.data.contact[k].geom2 = int(b)

    def forward(self):
        self.n_forward += 1

    def get_state(self):
        return (copy.deepcopy(self.data.qpos), self.data.grip.copy())

    def set_state(self, st):
        self.data.qpos = copy.deepcopy(st[0])
        self.data.grip = st[1].copy()
        self.data.nan_now = False
        self.data.regenerate()


class FakeTextureModder:
    def __init__(self, sim):
        pass

    def set_rgb(self, name, color):
        pass


# --------------------------------------------------------------------------
# gym stand-ins
# --------------------------------------------------------------------------
class GoalEnvStandIn:
    metadata = {}

    @property
    def unwrapped(self):
        return self


class BoxStandIn:
    def __init__(self, low, high, shape=None, dtype='float32'):
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=dtype)
        self.high = np.full(self.shape, high, dtype=dtype)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState(0)

    def sample(self):
        return self._rng.uniform(-1, 1, size=self.shape).astype(self.dtype)


class DictStandIn:
    def __init__(self, spaces):
        self.spaces = spaces


class EzPickleStandIn:
    def __init__(self, *a, **k):
        pass


class TimeLimitStandIn:
    """What `gym.make` wraps around an env registered with max_episode_steps
    (gym_blocks/__init__.py:6-53).  Third-party behaviour recalled from gym
    0.10: done = elapsed_steps >= max_episode_steps."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reset_log = []     # num_objs after every reset (Variation draws it in _reset_sim, fetch_env.py:647-649)

    @property
    def unwrapped(self):
        return self.env

    def reset(self):
        self._elapsed_steps = 0
        obs = self.env.reset()
        self.reset_log.append(int(self.env.num_objs))
        return obs

    def step(self, action):
        obs, r, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            done = True
        return obs, r, done, info

    def seed(self, seed=None):
        return self.env.seed(seed)

    def compute_reward(self, achieved_goal, desired_goal, info):
        return self.env.compute_reward(achieved_goal, desired_goal, info)

    def render(self, *a, **k):
        pass


REGISTRY = []
CTRL_LOG = []        # every 9-vector _set_action hands to the robotics utils (fetch_env.py:181-185)


def install():
    """Inject the stand-in modules and import the reference package."""
    if 'gym_blocks' in sys.modules:
        return sys.modules['gym_blocks']
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    gym = _mod('gym', GoalEnv=GoalEnvStandIn)
    gym.error = _mod('gym.error', DependencyNotInstalled=type('DependencyNotInstalled', (Exception,), {}))
    gym.spaces = _mod('gym.spaces', Box=BoxStandIn, Dict=DictStandIn)
    gym.utils = _mod('gym.utils', EzPickle=EzPickleStandIn)
    gym.utils.seeding = _mod('gym.utils.seeding',
                             np_random=lambda seed=None: (np.random.RandomState(seed), seed))
    gym.envs = _mod('gym.envs')
    gym.envs.registration = _mod('gym.envs.registration',
                                 register=lambda **kw: REGISTRY.append(kw))

    def mocap_set_action(sim, action):
        # kinematic stand-in: the gripper site follows the position delta
        sim.data.grip = sim.data.grip + np.asarray(action[:3])

    robotics_utils = _mod(
        'gym.envs.robotics.utils',
        ctrl_set_action=lambda sim, action: CTRL_LOG.append(np.array(action, copy=True)),
        mocap_set_action=mocap_set_action,
        reset_mocap_welds=lambda sim: None,
        robot_get_obs=lambda sim: (sim.data.robot[0].copy(), sim.data.robot[1].copy()))
    # mat2euler stand-in: a deterministic function of the matrix (its first row) in rich
    # mode, so the fixture can hand the same "euler angles" to the build's assembly kernel
    rotations = _mod('gym.envs.robotics.rotations',
                     mat2euler=lambda m: np.asarray(m)[0].copy() if np.any(np.asarray(m) != np.eye(3))
                     else np.zeros(3))
    gym.envs.robotics = _mod('gym.envs.robotics', utils=robotics_utils, rotations=rotations)

    mj = _mod('mujoco_py',
              MujocoException=type('MujocoException', (Exception,), {}),
              load_model_from_path=lambda p: FakeModel(p),
              MjSim=FakeSim, MjViewer=object)
    mj.modder = _mod('mujoco_py.modder', TextureModder=FakeTextureModder)

    # baselines.her.util -> the reference's vendored copy (gym_blocks/util.py)
    tf = _mod('tensorflow')
    bl = _mod('baselines')
    bl.common = _mod('baselines.common')
    bl.common.tf_util = _mod('baselines.common.tf_util')
    bl.her = _mod('baselines.her')

    import gym_blocks  # noqa: F401  (runs the 7 register() calls)
    import gym_blocks.util as vendored_util
    sys.modules['baselines.her.util'] = vendored_util
    bl.her.util = vendored_util
    return gym_blocks


def make(env_id):
    """gym.make stand-in: instantiate the registered entry point and wrap it
    in the TimeLimit stand-in."""
    install()
    import importlib
    for kw in REGISTRY:
        if kw['id'] == env_id:
            modname, clsname = kw['entry_point'].split(':')
            cls = getattr(importlib.import_module(modname), clsname)
            env = cls(**kw.get('kwargs', {}))
            return TimeLimitStandIn(env, kw['max_episode_steps'])
    raise KeyError(env_id)


class ContactScript:
    """Seeded generator of plausible contact lists over a model's geoms.
    Records everything it emits so the fixture can replay it."""

    def __init__(self, model, seed, p_pair=0.15):
        self.rng = np.random.RandomState(seed)
        names = model.geom_names
        self.ngeom = len(names)
        self.fingers = [i for i, n in enumerate(names) if n and 'finger' in n]
        self.table = [i for i, n in enumerate(names) if n == 'table']
        self.objects = [i for i, n in enumerate(names) if n and n.startswith('object')]
        self.other = [i for i, n in enumerate(names)
                      if i not in self.fingers + self.table + self.objects]
        self.p_pair = p_pair
        self.log = []

    def __call__(self):
        r = self.rng
        pairs = []
        # cubes mostly rest on the table
        for o in self.objects:
            if r.rand() < 0.8:
                pairs.append((self.table[0], o) if r.rand() < 0.5 else (o, self.table[0]))
        # object-object and finger-object touches
        movers = self.fingers + self.objects
        for a in range(len(movers)):
            for b in range(a + 1, len(movers)):
                if r.rand() < self.p_pair:
                    pairs.append((movers[a], movers[b]))
        # contacts involving geoms that map to None (robot links, floor, mocap)
        if r.rand() < 0.3:
            pairs.append((self.other[r.randint(len(self.other))],
                          movers[r.randint(len(movers))]))
        if r.rand() < 0.1:
            a, b = r.randint(len(self.other)), r.randint(len(self.other))
            pairs.append((self.other[a], self.other[b]))
        # rare: finger-finger (quirk Q5 -> diagonal [0][0]) and self pairs
        if r.rand() < 0.03:
            pairs.append((self.fingers[0], self.fingers[1]))
        if r.rand() < 0.02:
            pairs.append((self.fingers[0], self.table[0]))
        idx = r.permutation(len(pairs))
        pairs = [pairs[i] for i in idx][:FakeData.MAXCON]
        self.log.append(pairs)
        return pairs


class RecordingRandom:
    """Wraps a numpy RandomState and logs the STANDARD variates behind every uniform() /
    normal() call (uniform(lo,hi) == lo + (hi-lo)*random_sample(), normal() ==
    standard_normal() for NumPy's legacy RandomState -- checked against a clone on every
    call), so a fixture can hand the identical draws to the build."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.u, self.z, self.ints = [], [], []

    def _clone(self):
        c = np.random.RandomState()
        c.set_state(self.rs.get_state())
        return c

    def uniform(self, low=0.0, high=1.0, size=None):
        ref = self._clone().uniform(low, high, size)
        u = self.rs.random_sample(size)
        out = low + (high - low) * u
        assert np.array_equal(np.asarray(out), np.asarray(ref))
        self.u.extend(np.atleast_1d(u).tolist())
        return out

    def normal(self, loc=0.0, scale=1.0, size=None):
        assert loc == 0.0 and scale == 1.0
        ref = self._clone().normal(size=size)
        z = self.rs.standard_normal(size)
        assert np.array_equal(np.asarray(z), np.asarray(ref))
        self.z.extend(np.atleast_1d(z).tolist())
        return z

    def randint(self, *a, **k):
        v = self.rs.randint(*a, **k)
        self.ints.append(int(v))
        return v

    def clear(self):
        del self.u[:], self.z[:], self.ints[:]
