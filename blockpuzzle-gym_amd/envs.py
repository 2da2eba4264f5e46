"""Batched, GPU-resident mirror of the reference env surface.

Mirrors (paths relative to the reference root):
  gym_blocks/envs/robot_env.py  RobotEnv.step/reset/seed, action/observation spaces
  gym_blocks/envs/fetch_env.py  BlocksEnv + the five variants: compute_reward,
                                set_test, increase_difficulty, get_difficulty
  gym TimeLimit                 `_max_episode_steps`, done after 50 steps
                                (gym_blocks/__init__.py registers every id with 50)

One `BatchedBlocksEnv` replaces the reference's `[make_env() for _ in range(B)]`
(rollout.py:33): all B envs live packed in HBM behind one C-ABI handle and are
stepped by one HIP kernel launch.

Boundary: contacts-in.  MuJoCo (robot_env.py:60) is a closed third-party binary
outside the reference tree; each step consumes the step's contact set, either
  * packed pair masks           step(u, contact_masks=uint32[B])
  * geom-id pair lists          step(u, contacts=(int16[B,maxcon,2], int32[B]))   padded, or
                                step(u, contacts=(int16[total,2], int64[B+1]))    ragged (CSR offsets)
    (what fetch_env.py:158-167 reads from sim.data.contact), or
  * the env's `contact_source`  (default: the build-defined synthetic source).
`observation` has no physics behind it: reset() rewrites it to what the reference
fills without physics (Variation: obs[0] = num_blocks and the colour one-hots,
fetch_env.py:579-600) and zeros elsewhere; a backend writes the rest
(`assemble_observation`, from its on_reset and on every step).
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))

ENV_FEATURES = 10          # actor_critic.py:4-7 / fetch_env.py:222
BLOCK_BASE_FEATURES = 15
COLOR_FEATURES = 4
BLOCK_FEATURES = BLOCK_BASE_FEATURES + COLOR_FEATURES

# scene file of each variant (tasks.py) -> default geom name list (assets/fetch_geoms.json)
_MODEL_XML = {
    'GripperTouch-v0': ['fetch/1block.xml'],
    'BlocksTouch-v0': ['fetch/2blocks.xml'],
    'BlocksTouchCurriculum-v0': ['fetch/2blocks.xml'],
    'BlocksTouchChoose-v0': ['fetch/3blocks.xml'],
    'BlocksTouchChooseCurriculum-v0': ['fetch/3blocks.xml'],
    'ToppleTower-v0': ['fetch/4blocks.xml'],
    # fetch_env.py:528-530: Variation builds one sim per block count
    'BlocksTouchVariation-v0': ['fetch/2blocks.xml', 'fetch/3blocks.xml', 'fetch/4blocks.xml'],
}


def default_geom_names(env_id):
    with open(os.path.join(_HERE, 'assets', 'fetch_geoms.json')) as f:
        models = json.load(f)['models']
    return [models[x] for x in _MODEL_XML[env_id]]


def geom2objid(name):
    """BlocksEnv._geom2objid (fetch_env.py:106-117); None for unmapped geoms."""
    r = _lib.lib().bpg_geom2objid(None if name is None else name.encode())
    return None if r < 0 else r


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency)."""

    def __init__(self, low, high, shape, dtype='float32'):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)


class DictSpace:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class _Spec:
    def __init__(self, env_id, max_episode_steps):
        self.id = env_id
        self.max_episode_steps = max_episode_steps


class _DevicePtr:
    """Zero-copy torch view of a raw device pointer (CUDA array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False),
                                             version=3, strides=None)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class PhysicsError(RuntimeError):
    """What a physics backend raises when its simulation blows up: the stand-in for
    mujoco_py.MujocoException, which the rollout worker answers by starting the rollout over
    (rollout.py:136-137).

    The reference steps its envs one after the other (rollout.py:121-137): when env i raises, envs 0 .. i-1 HAVE
    taken the step and envs i .. B-1 have not -- and with quirk Q1 (the base variants keep the touch matrix across
    resets, fetch_env.py:247-255) what those envs touched in that step shows in the retried episode.  A backend that
    knows which env failed says so: failed_env = i and, for the envs before it, the step's contacts (`masks` (B,)
    packed pair masks, or `contacts` = (pairs, ncon) geom-id lists; entries from i on are ignored); the worker then
    steps exactly envs 0 .. i-1 before it starts over (BatchedBlocksEnv.step_prefix).  Without failed_env no env takes
    the step."""

    def __init__(self, *args, failed_env=None, masks=None, contacts=None):
        super().__init__(*args)
        self.failed_env, self.masks, self.contacts = failed_env, masks, contacts


class SyntheticContacts:
    """Build-defined physics stand-in (no reference counterpart): contacts are a
    counter-based pseudo-random function of (seed, global env id, step) evaluated
    inside the step kernel; table-block pairs always touch, every other pair is
    Bernoulli(5/32).  See DESIGN.md section 5."""
    kind = 'synthetic'

    def next(self, env, actions):
        return None


class ScriptedContacts:
    """Replays pre-recorded contacts: `masks` (T, B) uint32, or
    `pairs` (T, B, maxcon, 2) int16 with `ncon` (T, B) int32."""

    def __init__(self, masks=None, pairs=None, ncon=None):
        self.masks, self.pairs, self.ncon = masks, pairs, ncon
        self.t = 0

    def on_reset(self, env, env_mask=None, **kw):
        if env_mask is None:
            self.t = 0                  # a full reset starts the script over

    def next(self, env, actions):
        t = self.t
        self.t += 1
        if self.masks is not None:
            return ('masks', self.masks[t])
        return ('pairs', self.pairs[t], self.ncon[t])


class CallableContacts:
    """Wraps a user physics backend.

    fn(env, actions) -> uint32 masks (B,)  or  (int16 pairs (B,maxcon,2), int32 ncon (B,)).
        `actions` are the policy's (B, 4) actions as given; `env.set_action_vector(actions)` is
        robot_env.py:58's clip + _set_action in one kernel and returns the control vector.
    on_reset(env, xy, status, env_mask, test): optional; called by reset() / set_test() with the
        freshly sampled cube positions `xy` (B, nb, 2) float64 -- _randomize_objects of the
        reference's _reset_sim (fetch_env.py:247-255) and set_test (:365-368, :443-446) -- the
        sampler's per-env `status` (0 = ok), the partial-reset mask (None = all envs) and whether
        the maximum-difficulty ranges were used.  Rows of `xy` outside the mask keep their previous
        positions.
    writes_observation: the backend rewrites every entry of `env.observation` on each step
        (env.assemble_observation does); the rollout workers then point `env.observation` at the
        episode's time slot so the write lands there directly, and watch it for NaNs
        (rollout.py:139-142)."""

    def __init__(self, fn, on_reset=None, writes_observation=False):
        self.fn = fn
        self._on_reset = on_reset
        self.wants_positions = on_reset is not None
        self.writes_observation = bool(writes_observation)

    def on_reset(self, env, xy=None, status=None, env_mask=None, test=False):
        if self._on_reset is not None:
            self._on_reset(env, xy, status, env_mask, test)

    def next(self, env, actions):
        out = self.fn(env, actions)
        if isinstance(out, (tuple, list)):
            return ('pairs', out[0], out[1])
        return ('masks', out)


class BatchedBlocksEnv:
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 25}
    reward_range = (-float('inf'), float('inf'))

    def __init__(self, env_id, num_envs=1, device=None, seed=None, max_episode_steps=None,
                 contact_source=None, env_id_base=0, scalar_api=False, reward_type='sparse'):
        L = _lib.lib()
        variant = L.bpg_variant_from_id(env_id.encode())
        if variant < 0:
            raise KeyError('No registered env with id: %s' % env_id)
        if not torch.cuda.is_available():
            raise RuntimeError('blockpuzzle-gym_amd needs a ROCm GPU; there is no CPU fallback')
        self.env_id = env_id
        self.info = _lib.variant_info(variant)
        self.variant = variant
        self.num_envs = int(num_envs)
        self.scalar_api = bool(scalar_api)
        if self.scalar_api and self.num_envs != 1:
            raise ValueError('scalar_api needs num_envs == 1')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.reward_type = reward_type
        self._max_episode_steps = int(max_episode_steps or self.info.max_episode_steps)
        self.spec = _Spec(env_id, self._max_episode_steps)
        self.env_id_base = int(env_id_base)
        self._seed_value = 0 if seed is None else int(seed)
        h = C.c_void_p()
        _lib.check(L.bpg_create(C.byref(h), variant, self.num_envs, self.device.index,
                                self._max_episode_steps, C.c_uint64(self._seed_value & (2**64 - 1)),
                                self.env_id_base))
        self._h = h
        self._L = L
        # fetch_env.py:64-86 constructor state
        i = self.info
        self.num_objs = i.num_objs
        self.side = i.side
        self.dimg, self.dimo, self.dimu = i.dimg, i.dimo, i.dimu
        self.block_gripper = bool(i.block_gripper)
        self.obj_colors = list(i.obj_colors[:i.num_objs])
        g = np.array(i.goal[:i.dimg], dtype=np.float64 if i.desired_goal_is_float else np.int64)
        self.goal = g
        if self.env_id == 'BlocksTouchVariation-v0':
            self.max_num_blocks = 4                # fetch_env.py:523
        self.contact_source = contact_source if contact_source is not None else SyntheticContacts()
        B, dev = self.num_envs, self.device
        self._g = torch.empty(B, self.dimg, dtype=torch.float32, device=dev)
        self._static_obs_stale = False
        self.observation = torch.zeros(B, self.dimo, dtype=torch.float32, device=dev)
        # cube xy of the current episode (_randomize_objects; filled when a contact source asks for it)
        self.block_xy = torch.full((B, self.num_objs - 2, 2), float('nan'), dtype=torch.float64, device=dev)
        self.placement_status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.grip_xy = (1.3419, 0.7491)            # initial_gripper_xpos[:2] of the settled Fetch arm (physics)
        _lib.check(L.bpg_get_desired_goal_f32(self._h, _ptr(self._g), self.dimg, _stream(dev)))
        now, ever = C.c_void_p(), C.c_void_p()
        _lib.check(L.bpg_state_ptrs(self._h, C.byref(now), C.byref(ever)))
        self.state_now = torch.as_tensor(_DevicePtr(now.value, B, '<i4'), device=dev)
        self.state_ever = torch.as_tensor(_DevicePtr(ever.value, B, '<i4'), device=dev)
        cnt = C.c_void_p()
        _lib.check(L.bpg_counter_ptr(self._h, C.byref(cnt)))
        self.state_counters = torch.as_tensor(_DevicePtr(cnt.value, 2, '<i4'), device=dev)
        self.set_geom_names(default_geom_names(env_id))
        # robot_env.py:38-44
        self.action_space = Box(-1., 1., shape=(self.dimu,), dtype='float32')
        self.observation_space = DictSpace(dict(
            desired_goal=Box(-np.inf, np.inf, shape=(self.dimg,), dtype='float32'),
            achieved_goal=Box(-np.inf, np.inf, shape=(self.dimg,), dtype='float32'),
            observation=Box(-np.inf, np.inf, shape=(self.dimo,), dtype='float32')))
        self._fill_static_observation()

    # ------------------------------------------------------------------ misc
    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                self._L.bpg_destroy(h)
            except Exception:
                pass

    close = __del__

    @property
    def unwrapped(self):
        return self

    def render(self, mode='human'):
        raise NotImplementedError('rendering needs the MuJoCo viewer (robot_env.py:89-105): out of scope')

    def with_num_envs(self, num_envs, **kw):
        """A fresh env of the same id/config with another batch size."""
        args = dict(device=self.device, seed=self._seed_value, max_episode_steps=self._max_episode_steps,
                    env_id_base=self.env_id_base, contact_source=self.contact_source)
        args.update(kw)
        return BatchedBlocksEnv(self.env_id, num_envs=num_envs, **args)

    # ----------------------------------------------------- curriculum surface
    # fetch_env.py:340-348 (BlocksTouch), :404-417 (Choose), :561-563 (Variation), tasks.py (obj_range = 0.15): the
    # difficulty counter and the placement ranges live in the C-ABI handle (bpg_curriculum: one copy for the batch, as
    # rollout.py:66-73 raises every env alike); the attributes the reference's envs carry are views of it.
    def _curriculum(self):
        c = _lib.Curriculum()
        _lib.check(self._L.bpg_get_curriculum(self._h, C.byref(c)))
        return c

    def _curriculum_attr(name, needs_step=False, kind=float):     # noqa: N805  (a property factory, not a method)
        def get(self):
            c = self._curriculum()
            if needs_step and not c.has_step:       # the preset's constructor never set it (quirk Q6, fetch_env.py:413-415)
                raise AttributeError("'%s' env has no attribute '%s'" % (self.env_id, name))
            return kind(getattr(c, name))

        def set_(self, value):
            c = self._curriculum()
            setattr(c, name, kind(value))
            if needs_step:
                c.has_step = 1
            _lib.check(self._L.bpg_set_curriculum(self._h, C.byref(c)))
        return property(get, set_)

    difficulty = _curriculum_attr('difficulty', kind=int)
    obj_range = _curriculum_attr('obj_range')
    obj_range_step = _curriculum_attr('obj_range_step', needs_step=True)
    max_obj_range = _curriculum_attr('max_obj_range')
    wrong_obj_range = _curriculum_attr('wrong_obj_range')
    wrong_obj_range_step = _curriculum_attr('wrong_obj_range_step', needs_step=True)
    challenge = _curriculum_attr('challenge', kind=bool)
    del _curriculum_attr

    def get_difficulty(self):
        return int(self._L.bpg_get_difficulty(self._h))          # fetch_env.py:96-97

    def increase_difficulty(self):
        """fetch_env.py:351-358 / :419-432 / :623-630 in the handle (bpg_increase_difficulty); True at the maximum level.
        NotImplementedError (:93-94) and AttributeError (quirk Q6) as the reference raises them."""
        top = C.c_int(0)
        rc = self._L.bpg_increase_difficulty(self._h, C.byref(top))
        if rc == _lib.ERR_UNSUPPORTED:
            raise NotImplementedError()
        if rc == _lib.ERR_ATTRIBUTE:
            raise AttributeError("'%s' env has no attribute 'obj_range_step'" % self.env_id)
        _lib.check(rc)
        return bool(top.value)

    def set_test(self, variates=None, env_mask=None, achieved_goal_out=None):
        if not self.info.set_test_supported:
            raise NotImplementedError()             # fetch_env.py:100-101
        # fetch_env.py:365-368 / :443-446 re-place the cubes at the maximum ranges and return
        # _get_obs(); Variation :641-644 only returns it (its re-placement is commented out).
        if not self.info.is_variation:
            self._place(test=True, env_mask=self._to_dev(env_mask, torch.uint8), variates=variates)
        return self._get_obs(ag_out=achieved_goal_out)

    def _place(self, test, env_mask, variates=None):
        """_randomize_objects for the envs being reset, handed to the contact source's on_reset
        (the physics backend sets the cubes' qpos[:2] from it).  Skipped when nothing consumes it."""
        src = self.contact_source
        if getattr(src, 'wants_positions', False):
            self.randomize_objects(test=test, grip_xy=self.grip_xy, variates=variates, out=self.block_xy,
                                   status=self.placement_status, env_mask=env_mask)
            src.on_reset(self, xy=self.block_xy, status=self.placement_status, env_mask=env_mask, test=test)
        elif hasattr(src, 'on_reset'):
            src.on_reset(self, env_mask=env_mask, test=test)

    # ------------------------------------------------------------ id2obj
    def set_geom_names(self, models):
        """models: one geom-name list per scene (3 for Variation).  Builds the
        id2obj tables exactly as fetch_env.py:284 / :651 do."""
        if models and (models[0] is None or isinstance(models[0], str)):
            models = [models]
        self.id2obj = []
        for k, names in enumerate(models):
            ids = [geom2objid(n) for n in names]
            self.id2obj.append(ids)
            arr = np.array([-1 if v is None else v for v in ids], dtype=np.int8)
            _lib.check(self._L.bpg_set_id2obj(self._h, k, arr.ctypes.data_as(C.POINTER(C.c_int8)), len(arr)))

    # ------------------------------------------------------------ gym API
    def seed(self, seed=None):
        """robot_env.py:53-55 (returns [seed]); re-keys the counter-based generator.  Env i of the batch
        (global index env_id_base + i) is keyed by seed + 1000 * i -- the per-env seeds RolloutStudent.seed
        hands out (rollout.py:206-210): env i seeded s draws what env 0 seeded s + 1000 * i draws."""
        self._seed_value = 0 if seed is None else int(seed)
        _lib.check(self._L.bpg_seed(self._h, C.c_uint64(self._seed_value & (2**64 - 1))))
        self.action_space.seed(seed)
        return [seed]

    def _to_dev(self, x, dtype):
        if x is None:
            return None
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.ascontiguousarray(x))
        return x.to(device=self.device, dtype=dtype).contiguous()

    def reset(self, num_grey=None, env_mask=None, variates=None, observation_out=None, achieved_goal_out=None):
        """robot_env.py:71-82.  num_grey: optional explicit (B,) draws for Variation's
        np_random.randint(3) (fetch_env.py:647); env_mask: optional (B,) partial reset (the
        reference's reset_rollout(i)); variates: optional explicit placement draws (see
        randomize_objects); observation_out: optional (B, dimo) float32 buffer that becomes
        `env.observation`; achieved_goal_out: optional (B, dimg) float32 view (unit inner stride, e.g.
        time slot 0 of an episode) that receives obs['achieved_goal'].  One launch resets the state and rewrites the observation rows of the envs
        being reset (the entries the reference fills without physics, zeros in the physics-owned ones:
        a backend writes those from its on_reset); when the contact source consumes positions (`wants_positions`) the cubes are
        re-placed on the same stream and handed to its on_reset, as _reset_sim does
        (fetch_env.py:247-255 -> _randomize_objects)."""
        ng = self._to_dev(num_grey, torch.int32)
        em = self._to_dev(env_mask, torch.uint8)
        if observation_out is not None:
            # the caller's buffer (e.g. time slot 0 of an episode) becomes the env's observation: the
            # reset launch writes the reset observation straight into it
            assert observation_out.shape == (self.num_envs, self.dimo) and observation_out.dtype == torch.float32 \
                and observation_out.stride(1) == 1 and env_mask is None
            self.observation = observation_out
        if em is None:
            self._static_obs_stale = False          # the reset launch rewrites every row
        obs = self.observation                      # (a partial reset of a stale buffer refreshes the other rows first)
        _lib.check(self._L.bpg_reset_obs(self._h, _ptr(ng), _ptr(em), _ptr(obs), int(obs.stride(0)),
                                         _stream(self.device)))
        self._place(test=False, env_mask=em, variates=variates)
        return self._get_obs(ag_out=achieved_goal_out)

    def _step_core(self, action, contact_masks=None, contacts=None, ag_out=None, succ_out=None, reward_out=None,
                   u_out=None, count_out=None, want_reward=True, want_flags=True, return_acc=None):
        """Launches one env step; returns (reward, flags, achieved_goal) device tensors.
        want_reward / want_flags = False: that output is not produced at all (None is returned in its place) -- the rollout
        worker ignores the step reward (rollout.py:123-125) and reads is_success from succ_out.
        return_acc: (B,) float32 device tensor, += the step reward (the per-env episode return of a per-step rollout).
        count_out: int64 device tensor; the launch adds the number of envs whose has_succeeded latch is set after the
        step (the rollout worker passes it with the last step of a rollout, rollout.py:163-167)."""
        a = self._to_dev(action, torch.float32)
        if a is not None:
            a = a.reshape(self.num_envs, self.dimu)
        # robot_env.py:58's clip happens where the action is consumed: k_set_action (set_action_vector)
        # clips in the kernel that builds the physics control vector; nothing behind the contacts-in
        # boundary reads the action itself
        self.last_action = a
        if contact_masks is None and contacts is None:
            nxt = self.contact_source.next(self, a)
            if nxt is not None:
                if nxt[0] == 'masks':
                    contact_masks = nxt[1]
                else:
                    contacts = (nxt[1], nxt[2])
        st = _stream(self.device)
        m = self._masks_of(contact_masks, contacts, st)
        # fresh output tensors every call (the reference returns copies, fetch_env.py:225-227);
        # the kernels write straight into them, so this costs an allocation, not a copy
        if reward_out is not None:
            reward = reward_out
            assert reward.shape == (self.num_envs,) and reward.dtype == torch.float32 and reward.is_contiguous()
        else:
            reward = torch.empty(self.num_envs, dtype=torch.float32, device=self.device) if want_reward else None
        flags = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device) if want_flags else None
        if ag_out is None:
            ag = torch.empty(self.num_envs, self.dimg, dtype=torch.float32, device=self.device)
        else:
            ag = ag_out
            assert ag.shape == (self.num_envs, self.dimg) and ag.dtype == torch.float32 and ag.stride(1) == 1
        return self._launch_step_record(m, reward, flags, succ_out, ag, u_out, a, count_out, return_acc, st)

    def _masks_of(self, contact_masks, contacts, st):
        """The step's contacts as (B,) packed pair masks on the device, or None for the synthetic source."""
        m = None
        if contacts is not None:
            # geom-id contact lists as the reference reads them (fetch_env.py:158-166): one launch maps
            # them through id2obj to pair masks, the step launch below consumes those
            pairs = self._to_dev(contacts[0], torch.int16)
            m = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
            if pairs.dim() == 2:
                # ragged (CSR) form: pairs (total, 2), second element = offsets (B+1,) int64
                offsets = self._to_dev(contacts[1], torch.int64)
                assert pairs.shape[1] == 2 and offsets.shape == (self.num_envs + 1,)
                _lib.check(self._L.bpg_contacts_csr_to_masks(self._h, _ptr(pairs), _ptr(offsets), int(pairs.shape[0]),
                                                             _ptr(m), st))
            else:
                ncon = self._to_dev(contacts[1], torch.int32)
                assert pairs.dim() == 3 and pairs.shape[0] == self.num_envs and pairs.shape[2] == 2
                _lib.check(self._L.bpg_contacts_to_masks(self._h, _ptr(pairs), _ptr(ncon), int(pairs.shape[1]), _ptr(m), st))
        elif contact_masks is not None:
            m = contact_masks.to(self.device).contiguous() if (
                torch.is_tensor(contact_masks) and contact_masks.dtype in (torch.int32, torch.uint32)
            ) else self._to_dev(contact_masks, torch.int32)
            assert m.numel() == self.num_envs
        return m

    def _launch_step_record(self, m, reward, flags, succ_out, ag, u_out, a, count_out, return_acc, st):
        if succ_out is not None:
            assert succ_out.numel() == self.num_envs and succ_out.is_contiguous() and succ_out.dtype == torch.float32
        u_in = None
        if u_out is not None:        # the episode's action slot is filled by the step launch itself
            assert a is not None and u_out.shape == a.shape and u_out.is_contiguous() and self.dimu == 4
            u_in = a if a.is_contiguous() else a.contiguous()
        # step + float achieved_goal (+ float success, + the action store) in one launch; NULL masks =
        # the synthetic source
        if count_out is not None:
            assert count_out.dtype == torch.int64 and count_out.numel() >= 1 and count_out.is_cuda
        if return_acc is not None:
            assert return_acc.shape == (self.num_envs,) and return_acc.dtype == torch.float32 and return_acc.is_contiguous()
        dp = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        a = _lib.StepArgs(d_masks=dp(m), d_reward=dp(reward), d_flags=dp(flags), d_success_f32=dp(succ_out), d_ag=dp(ag),
                          ld_ag=int(ag.stride(0)), d_u_in=dp(u_in), d_u_out=dp(u_out), d_success_count=dp(count_out),
                          d_return_acc=dp(return_acc))
        _lib.check(self._L.bpg_step_record_v2(self._h, C.byref(a), st))
        return reward, flags, ag

    def _step_with_args(self, action, sa):
        """The rollout worker's step: `sa` is a bpg_step_args the worker filled with the addresses of the episode's time
        slots (plain integers: no tensor views are made per step); this adds the step's contacts from the contact
        source and the action pointer, and launches.  `action`: (B, 4) float32 contiguous device tensor."""
        if action.shape != (self.num_envs, self.dimu) or action.dtype != torch.float32 or not action.is_contiguous() \
                or action.device != self.device:
            raise ValueError('step: actions must be a contiguous (%d, %d) float32 tensor on %s, got %s %s on %s'
                             % (self.num_envs, self.dimu, self.device, tuple(action.shape), action.dtype, action.device))
        self.last_action = action
        nxt = self.contact_source.next(self, action)
        st = _stream(self.device)
        m = None
        if nxt is not None:
            m = self._masks_of(nxt[1], None, st) if nxt[0] == 'masks' else self._masks_of(None, (nxt[1], nxt[2]), st)
        sa.d_masks = None if m is None else m.data_ptr()
        sa.d_u_in = action.data_ptr() if sa.d_u_out else None
        _lib.check(self._L.bpg_step_record_v2(self._h, C.byref(sa), st))

    def step_prefix(self, count, action=None, contact_masks=None, contacts=None):
        """One env step for envs 0 .. count-1 ONLY -- the state the reference's per-env loop (rollout.py:121-137) leaves
        behind when env `count` raises in the middle of a step.  An error path (PhysicsError.failed_env): the whole
        batch is stepped by the usual launch and the packed state of envs count .. B-1 is put back; no outputs."""
        count = int(count)
        if count <= 0:
            return
        if contact_masks is None and contacts is None:
            raise ValueError('step_prefix needs the contacts of the envs that took the step')
        keep_now, keep_ever = self.state_now[count:].clone(), self.state_ever[count:].clone()
        if action is None:
            action = torch.zeros(self.num_envs, self.dimu, dtype=torch.float32, device=self.device)
        self._step_core(action, contact_masks, contacts, want_reward=False, want_flags=False)
        self.state_now[count:].copy_(keep_now)
        self.state_ever[count:].copy_(keep_ever)

    def step(self, action, contact_masks=None, contacts=None, ag_out=None):
        """robot_env.py:57-69 (+ TimeLimit).  Returns (obs, reward, done, info).
        ag_out: optional (B, dimg) float32 view with unit inner stride (e.g. a time slot of
        an episode buffer) that receives obs['achieved_goal'] directly."""
        reward, flags, ag = self._step_core(action, contact_masks, contacts, ag_out)
        obs = self._obs_dict(ag)
        done = (flags & _lib.FLAG_DONE) != 0
        succ = (flags & _lib.FLAG_SUCCESS) != 0
        if self.scalar_api:
            return obs, np.float32(reward[0].item()), bool(done[0].item()), {'is_success': bool(succ[0].item())}
        return obs, reward, done, {'is_success': succ}

    def _obs_dict(self, ag):
        if self.scalar_api:
            # reference dtypes (quirk Q9): float64 obs/ag, int64 goal (Variation float64)
            return {'observation': self.observation[0].double().cpu().numpy(),
                    'achieved_goal': ag[0].double().cpu().numpy(),
                    'desired_goal': self.goal.copy()}
        # achieved_goal is a fresh tensor; observation / desired_goal are the env's shared buffers
        return {'observation': self.observation, 'achieved_goal': ag, 'desired_goal': self._g}

    def _get_obs(self, ag_out=None):
        """fetch_env.py:219-228 (achieved_goal / desired_goal parts)."""
        if ag_out is None:
            ag = torch.empty(self.num_envs, self.dimg, dtype=torch.float32, device=self.device)
        else:
            ag = ag_out
            assert ag.shape == (self.num_envs, self.dimg) and ag.dtype == torch.float32 and ag.stride(1) == 1
        _lib.check(self._L.bpg_get_achieved_goal_f32(self._h, _ptr(ag), int(ag.stride(0)), _stream(self.device)))
        return self._obs_dict(ag)

    def achieved_goal_f64(self):
        out = torch.empty(self.num_envs, self.dimg, dtype=torch.float64, device=self.device)
        _lib.check(self._L.bpg_get_achieved_goal_f64(self._h, _ptr(out), self.dimg, _stream(self.device)))
        return out

    def _fill_static_observation(self):
        """The part of Variation's _get_obs that needs no physics (fetch_env.py:579-607):
        obs[0] = num_blocks, colour one-hot of every present block, zero padding -- from the
        state as it stands (reset() does the same inside its own launch)."""
        self._static_obs_stale = False
        obs = self._observation
        _lib.check(self._L.bpg_fill_static_obs(self._h, _ptr(obs), int(obs.stride(0)), _stream(self.device)))

    def _mark_static_observation_stale(self):
        """The launches just enqueued reset envs without rewriting their observation rows (K rollouts in one launch,
        RolloutStudent.evaluate_rollouts_fused): the physics-free entries (Variation's num_blocks / colour one-hots) are
        refreshed when somebody next looks at `env.observation` -- 365 MB at 1 M envs that an evaluator loop which only keeps the
        success rates never pays."""
        if self.info.is_variation:
            self._static_obs_stale = True

    @property
    def observation(self):
        if self._static_obs_stale:
            self._fill_static_observation()
        return self._observation

    @observation.setter
    def observation(self, t):
        self._observation = t

    def set_action_vector(self, action):
        """robot_env.py:58 + fetch_env.py:170-185: (B,4) actions -> the (B,9) float64 vector
        [pos delta, fixed quaternion, finger targets] a physics backend consumes."""
        a = self._to_dev(action, torch.float32).reshape(self.num_envs, self.dimu)
        out = torch.empty(self.num_envs, 9, dtype=torch.float64, device=self.device)
        _lib.check(self._L.bpg_set_action_f64(self._h, _ptr(a), _ptr(out), _stream(self.device)))
        return out

    PHYS_FIELDS = ('grip_pos', 'grip_velp', 'gripper_state', 'gripper_vel',
                   'obj_pos', 'obj_rot', 'obj_velp', 'obj_velr')

    def assemble_observation(self, phys, dt=0.04, dtype=torch.float32, out=None, nan_flag=None):
        """_get_obs' observation vector (fetch_env.py:187-228, Variation :567-621) from a
        physics backend's state: `phys` maps PHYS_FIELDS to float64 arrays/tensors shaped
        (B,3),(B,3),(B,2),(B,2),(B,nb,3)x4 (euler angles in obj_rot; velocities unscaled);
        dt = nsubsteps * timestep = 20 * 0.002 for every preset (tasks.py).  float64 output
        is bit-identical to the reference; float32 is what rollout.py stores.  With
        out=None and float32 the env's shared `observation` buffer is updated.
        nan_flag: optional int32 device tensor, |= 1 when a written value is a NaN (the worker's
        np.isnan(o_new).any(), rollout.py:139-142, inside this launch).  A rollout worker that watches
        `env.observation` installs its flag as `env.observation_nan_flag`; writes into that buffer
        use it by default and tell the worker that the step's observation has been checked."""
        t = {k: self._to_dev(phys[k], torch.float64) for k in self.PHYS_FIELDS}
        nb = self.num_objs - 2
        for k in ('grip_pos', 'grip_velp'):
            assert t[k].shape == (self.num_envs, 3), k
        for k in ('gripper_state', 'gripper_vel'):
            assert t[k].shape == (self.num_envs, 2), k
        for k in ('obj_pos', 'obj_rot', 'obj_velp', 'obj_velr'):
            assert t[k].shape == (self.num_envs, nb, 3), k
        if out is None:
            out = self.observation if dtype == torch.float32 else torch.empty(
                self.num_envs, self.dimo, dtype=torch.float64, device=self.device)
        watched = out.data_ptr() == self.observation.data_ptr()
        if nan_flag is None and watched:
            nan_flag = getattr(self, 'observation_nan_flag', None)
        if nan_flag is not None:
            assert nan_flag.dtype == torch.int32 and nan_flag.numel() >= 1 and nan_flag.device == self.device
            if watched:
                self._observation_checked = True
        ph = _lib.PhysState(dt=float(dt), d_nan_flag=None if nan_flag is None else nan_flag.data_ptr(),
                            **{'d_' + k: t[k].data_ptr() for k in self.PHYS_FIELDS})
        fn = self._L.bpg_assemble_obs_f32 if out.dtype == torch.float32 else self._L.bpg_assemble_obs_f64
        _lib.check(fn(self._h, C.byref(ph), _ptr(out), int(out.stride(0)), _stream(self.device)))
        return out

    def contacts_to_masks(self, pairs, ncon, num_grey=None, out=None):
        """The contact lists of a whole recorded rollout -> the (T, B) mask stream the fused rollout reads
        (generate_rollouts_fused(masks=...)): `pairs` (T, B, maxcon, 2) int16 geom-id pairs and `ncon` (T, B) int32, as
        a physics trace stores sim.data.contact of every step (fetch_env.py:158-166), mapped through id2obj on the
        device in one launch.  num_grey: the (B,) draws of the episode the trace belongs to (Variation picks the sim and
        its id2obj table by num_objs = 4 + num_grey, fetch_env.py:647-651) when the rollout's reset has not happened
        yet; None = the model of the env's current state.  Returns a (T, B) int32 device tensor with the row stride
        bpg_row_stride(B)."""
        p = self._to_dev(pairs, torch.int16)
        nc = self._to_dev(ncon, torch.int32)
        assert p.dim() == 4 and p.shape[1] == self.num_envs and p.shape[3] == 2 and nc.shape == p.shape[:2]
        T, maxcon = int(p.shape[0]), int(p.shape[2])
        ng = self._to_dev(num_grey, torch.int32)
        if out is None:
            ld = int(self._L.bpg_row_stride(self.num_envs))
            out = torch.empty((T, ld), dtype=torch.int32, device=self.device)[:, :self.num_envs]
        assert out.shape == (T, self.num_envs) and out.stride(1) == 1 and out.element_size() == 4
        _lib.check(self._L.bpg_contacts_to_masks_tb(self._h, _ptr(p), _ptr(nc), T, maxcon, _ptr(ng), _ptr(out),
                                                    int(out.stride(0)) if T > 1 else self.num_envs, _stream(self.device)))
        return out

    def randomize_objects(self, test=False, grip_xy=(1.3419, 0.7491), variates=None, out=None, status=None,
                          env_mask=None):
        """_randomize_objects for the whole batch (fetch_env.py:328-336, 370-399, 448-517,
        697-764, 777-787): (B, nb, 2) float64 cube positions for the physics backend, using the
        env's current curriculum state (obj_range, ...).  grip_xy = initial_gripper_xpos[:2]
        (a physics output; the default is the settled Fetch gripper position).  variates:
        optional dict eu (B,n_eu), gz (B,n_gz), gu (B,n_gu) of explicit standard draws in the
        reference's order; default = the counter-based generator.  env_mask: optional (B,) uint8,
        only those envs are placed (rows of `out` elsewhere keep their values).
        Returns (xy, status)."""
        nb = self.num_objs - 2
        if out is None:
            out = torch.full((self.num_envs, nb, 2), float('nan'), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        em = self._to_dev(env_mask, torch.uint8)
        a = _lib.PlacementArgs(grip_x=float(grip_xy[0]), grip_y=float(grip_xy[1]),
                               d_env_mask=None if em is None else em.data_ptr())
        # obj_range / max_obj_range / wrong_obj_range / challenge as the handle holds them.  (`test` is passed on as given:
        # the sampler's test mode is also used by variants whose set_test() raises, through randomize_objects(test=True).)
        _lib.check(self._L.bpg_placement_from_curriculum(self._h, 0, C.byref(a)))
        a.test = int(bool(test))
        keep = [em]
        if variates is not None:
            for k in ('eu', 'gz', 'gu'):
                t = self._to_dev(variates[k], torch.float64)
                assert t.dim() == 2 and t.shape[0] == self.num_envs
                keep.append(t)
                setattr(a, 'd_' + k, t.data_ptr())
                setattr(a, 'n_' + k, int(t.shape[1]))
        _lib.check(self._L.bpg_randomize_objects(self._h, C.byref(a), _ptr(out), _ptr(status), _stream(self.device)))
        return out, status

    def num_objs_per_env(self):
        return (self.state_ever >> 21) & 7

    # ------------------------------------------------------------ reward
    def compute_reward(self, achieved_goal, goal=None, info=None, desired_goal=None):
        """fetch_env.py:135-143, vectorised over leading axes; accepts the `goal`
        (definition) and `desired_goal` (config.py:110-111 call site) spellings."""
        if goal is None:
            goal = desired_goal
        return compute_reward(achieved_goal, goal, device=self.device)

    def success_count(self):
        """Number of envs whose has_succeeded latch is set (rollout.py:163-167)."""
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(self._L.bpg_success_count(self._h, _ptr(cnt), _stream(self.device)))
        return cnt

    # ------------------------------------------------------------ state I/O
    def state_dict(self):
        c = self._curriculum()
        return {'now': self.state_now.clone(), 'ever': self.state_ever.clone(),
                'counters': self.state_counters.clone(),
                # the handle's difficulty and placement ranges, field order of bpg_curriculum (float64 holds the ints exactly)
                'curriculum': torch.tensor([float(getattr(c, name)) for name, _ in _lib.Curriculum._fields_], dtype=torch.float64)}

    def load_state_dict(self, sd):
        self.state_now.copy_(sd['now'])
        self.state_ever.copy_(sd['ever'])
        if 'counters' in sd:
            self.state_counters.copy_(sd['counters'])
        if 'curriculum' in sd:
            c = _lib.Curriculum()
            for (name, ctype), v in zip(_lib.Curriculum._fields_, sd['curriculum'].tolist()):
                setattr(c, name, int(v) if ctype is C.c_int32 else v)
            _lib.check(self._L.bpg_set_curriculum(self._h, C.byref(c)))
        self._fill_static_observation()


def mat2euler(mat, device=None):
    """gym.envs.robotics.rotations.mat2euler on the GPU: (..., 3, 3) float64 -> (..., 3).
    Third-party formula restated from memory (parity unpinned, see include/bpgym.h)."""
    dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
    m = (mat if torch.is_tensor(mat) else torch.as_tensor(np.asarray(mat))).to(dev, torch.float64).contiguous()
    assert m.shape[-2:] == (3, 3)
    n = int(m.numel() // 9)
    out = torch.empty(m.shape[:-2] + (3,), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().bpg_mat2euler_f64(_ptr(m), n, _ptr(out), dev.index, _stream(dev)))
    return out


def compute_reward(achieved_goal, goal, device=None):
    """Stateless BlocksEnv.compute_reward on the GPU.  numpy in -> numpy float32 out,
    torch in -> torch float32 out (on `device`).  The products and their sum are taken in the precision NumPy
    would use for `achieved_goal * goal` (fetch_env.py:141) -- np.result_type of the two arrays: float64 for the
    reference's own call (a float64 matrix, robot_env.py:66), float32 for HER's float32 replay arrays -- and in
    NumPy's summation order; for the ternary rows of the env any order is exact.  Rows of up to 256 entries."""
    L = _lib.lib()
    np_in = not torch.is_tensor(achieved_goal)
    dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
    ag = torch.as_tensor(np.asarray(achieved_goal) if np_in else achieved_goal)
    g = torch.as_tensor(np.asarray(goal) if not torch.is_tensor(goal) else goal)
    # the precision of `achieved_goal * goal` by NumPy's promotion rules (float32 x int8 / int16 / bool / float16 stays
    # float32; float32 x int32 / int64 / float64 is float64; integer x integer stays integer -- exact, run in float64)
    try:
        rt = np.result_type(*(np.dtype(str(t.dtype).replace('torch.', '')) for t in (ag, g)))
    except TypeError:
        raise TypeError('compute_reward: %s x %s arrays have no NumPy counterpart' % (ag.dtype, g.dtype)) from None
    if rt == np.float16:
        raise NotImplementedError('compute_reward on float16 x float16 rows (NumPy would sum in float16)')
    wide = rt != np.float32
    ft = torch.float64 if wide else torch.float32
    ag = ag.to(dev, ft).contiguous()
    g = g.to(dev, ft)
    lead = ag.shape[:-1]
    n = int(np.prod(lead, dtype=np.int64)) if len(lead) else 1
    out = torch.empty(n, dtype=torch.float32, device=dev)
    fn_row, fn_rows = (L.bpg_compute_reward_f64_goal, L.bpg_compute_reward_f64) if wide else \
        (L.bpg_compute_reward_f32_goal, L.bpg_compute_reward_f32)
    # one goal row for every row (a 1-D goal, or leading axes that are all broadcast: size 1 / stride 0)
    # goes to the kernel as ONE row; anything else row by row
    one_row = g.shape[-1] == ag.shape[-1] and all(sz == 1 or st == 0 for sz, st in zip(g.shape[:-1], g.stride()[:-1]))
    if n and one_row:
        row = g.reshape(-1)[:ag.shape[-1]] if g.dim() == 1 else g[(0,) * (g.dim() - 1)]
        row = row.contiguous()
        _lib.check(fn_row(_ptr(ag), _ptr(row), n, int(ag.shape[-1]), _ptr(out), dev.index, _stream(dev)))
    elif n:
        g = g.expand_as(ag).contiguous()
        _lib.check(fn_rows(_ptr(ag), _ptr(g), n, int(ag.shape[-1]), _ptr(out), dev.index, _stream(dev)))
    out = out.reshape(lead)
    return out.cpu().numpy() if np_in else out
