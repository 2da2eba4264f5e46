"""ctypes loader for the CPU oracle (oracle/bpg_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libbpg_oracle.so')

MAX_G = 36
VARIANTS = {
    'GripperTouch-v0': 0, 'BlocksTouch-v0': 1, 'ToppleTower-v0': 2,
    'BlocksTouchCurriculum-v0': 3, 'BlocksTouchChoose-v0': 4,
    'BlocksTouchChooseCurriculum-v0': 5, 'BlocksTouchVariation-v0': 6,
}


class Phys(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in
                ('grip_pos', 'grip_velp', 'gripper_state', 'gripper_vel',
                 'obj_pos', 'obj_rot', 'obj_velp', 'obj_velr')]


class Placement(C.Structure):
    _fields_ = [('grip_x', C.c_double), ('grip_y', C.c_double), ('obj_range', C.c_double),
                ('max_obj_range', C.c_double), ('wrong_obj_range', C.c_double),
                ('test', C.c_int), ('challenge', C.c_int),
                ('eu', C.POINTER(C.c_double)), ('n_eu', C.c_int),
                ('gz', C.POINTER(C.c_double)), ('n_gz', C.c_int),
                ('gu', C.POINTER(C.c_double)), ('n_gu', C.c_int)]


class Env(C.Structure):
    _fields_ = [
        ('variant', C.c_int), ('num_objs', C.c_int), ('side', C.c_int),
        ('is_variation', C.c_int),
        ('achieved_goal', C.c_double * MAX_G),
        ('goal', C.c_longlong * MAX_G),
        ('obj_colors', C.c_int * 6),
        ('has_succeeded', C.c_int), ('elapsed_steps', C.c_int),
        ('max_episode_steps', C.c_int),
    ]


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                    for f in ('bpg_oracle.c', 'bpg_oracle.h'))
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
            build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.orc_geom2objid.argtypes = [C.c_char_p]
        L.orc_geom2objid.restype = C.c_int
        L.orc_sample_colors.argtypes = [C.c_int, P(C.c_int)]
        L.orc_sample_goal.argtypes = [P(C.c_int), C.c_int, P(C.c_longlong)]
        L.orc_env_init.argtypes = [P(Env), C.c_int, C.c_int]
        L.orc_env_reset.argtypes = [P(Env), C.c_int]
        L.orc_step_callback.argtypes = [P(Env), P(C.c_int), P(C.c_int), C.c_int]
        L.orc_step_callback_mask.argtypes = [P(Env), C.c_uint32]
        L.orc_check_goal.argtypes = [P(Env)]
        L.orc_compute_reward.argtypes = [P(C.c_double), P(C.c_double), C.c_int]
        L.orc_compute_reward.restype = C.c_float
        L.orc_compute_reward_batch.argtypes = [P(C.c_double), P(C.c_double), C.c_int, C.c_int, P(C.c_float)]
        L.orc_compute_reward_f32_batch.argtypes = [P(C.c_float), P(C.c_float), C.c_int, C.c_int, P(C.c_float), P(C.c_float)]
        L.orc_rowsum_f64.argtypes = [P(C.c_double), C.c_int, C.c_int, P(C.c_double)]
        L.orc_rowsum_f32.argtypes = [P(C.c_float), C.c_int, C.c_int, P(C.c_float)]
        L.orc_is_success.argtypes = [P(Env)]
        L.orc_env_step.argtypes = [P(Env), P(C.c_int), P(C.c_int), C.c_int,
                                   P(C.c_float), P(C.c_int), P(C.c_int)]
        L.orc_env_step_mask.argtypes = [P(Env), C.c_uint32, P(C.c_float), P(C.c_int), P(C.c_int)]
        L.orc_get_achieved_goal.argtypes = [P(Env), P(C.c_double)]
        L.orc_set_action.argtypes = [P(C.c_float), C.c_int, P(C.c_double)]
        L.orc_get_obs.argtypes = [P(Env), P(Phys), C.c_double, P(C.c_double)]
        L.orc_randomize_objects.argtypes = [P(Env), P(Placement), P(C.c_double), P(C.c_int)]
        L.orc_pair_index.argtypes = [C.c_int, C.c_int]
        L.orc_contacts_to_mask.argtypes = [P(C.c_int), P(C.c_int), C.c_int]
        L.orc_contacts_to_mask.restype = C.c_uint32
        L.orc_discounted_return.argtypes = [P(C.c_double), C.c_int, C.c_double, P(C.c_double)]
        L.orc_out_of_table.argtypes = [C.c_double, C.c_double]
        L.orc_synth_mask.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
        L.orc_synth_mask.restype = C.c_uint32
        L.orc_hash32.argtypes = [C.c_uint32]
        L.orc_hash32.restype = C.c_uint32
        L.orc_generate_rollouts.argtypes = [P(Env), C.c_int, C.c_int, P(C.c_uint32), P(C.c_int),
                                            P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float)]
        L.orc_generate_rollouts.restype = C.c_double
        L.orc_env_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_env_key.restype = C.c_uint32
        L.orc_num_grey_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_num_grey_draw.restype = C.c_int
        L.orc_bench_rollout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.orc_bench_rollout.restype = C.c_uint64
        L.orc_fill_masks.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, P(C.c_uint32)]
        L.orc_bench_rollout_masks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_uint32), P(C.c_float),
                                              P(C.c_uint8)]
        L.orc_bench_rollout_masks.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OracleEnv:
    """One reference-form env (matrix state) behind the contacts-in boundary."""

    def __init__(self, env_id, max_episode_steps=50):
        self.L = lib()
        self.e = Env()
        self.L.orc_env_init(C.byref(self.e), VARIANTS[env_id], max_episode_steps)

    @property
    def dimg(self):
        return self.e.side * self.e.side

    def reset(self, num_grey=0):
        self.L.orc_env_reset(C.byref(self.e), int(num_grey))
        return self.achieved_goal()

    def achieved_goal(self):
        out = np.zeros(self.dimg)
        self.L.orc_get_achieved_goal(C.byref(self.e), _p(out, C.c_double))
        return out

    def goal(self):
        return np.array(self.e.goal[:self.dimg], dtype=np.int64)

    def step(self, id2obj, contacts):
        """contacts: (ncon, 2) geom ids.  Returns (ag, reward f32, done, is_success)."""
        id2obj = np.ascontiguousarray(id2obj, np.int32)
        con = np.ascontiguousarray(contacts, np.int32).reshape(-1, 2)
        r, d, s = C.c_float(), C.c_int(), C.c_int()
        self.L.orc_env_step(C.byref(self.e), _p(id2obj, C.c_int), _p(con, C.c_int), len(con),
                            C.byref(r), C.byref(d), C.byref(s))
        return self.achieved_goal(), np.float32(r.value), bool(d.value), bool(s.value)

    def get_obs(self, phys, dt):
        """phys: dict of float64 arrays named as Phys fields (one env).  Returns observation."""
        keep = {k: np.ascontiguousarray(phys[k], np.float64) for k, _ in Phys._fields_}
        ph = Phys(**{k: _p(v, C.c_double) for k, v in keep.items()})
        out = np.zeros(128)
        n = self.L.orc_get_obs(C.byref(self.e), C.byref(ph), float(dt), _p(out, C.c_double))
        return out[:n].copy()

    def randomize_objects(self, grip_xy, obj_range, max_obj_range, wrong_obj_range, test, challenge,
                          eu, gz, gu):
        """_randomize_objects with explicit standard variates.  Returns (xy (nb,2), used[3], fail)."""
        eu, gz, gu = (np.ascontiguousarray(a, np.float64) for a in (eu, gz, gu))
        pl = Placement(float(grip_xy[0]), float(grip_xy[1]), float(obj_range), float(max_obj_range),
                       float(wrong_obj_range), int(test), int(challenge),
                       _p(eu, C.c_double), len(eu), _p(gz, C.c_double), len(gz), _p(gu, C.c_double), len(gu))
        xy = np.full((self.e.num_objs - 2, 2), np.nan)
        used = (C.c_int * 3)()
        fail = self.L.orc_randomize_objects(C.byref(self.e), C.byref(pl), _p(xy, C.c_double), used)
        return xy, list(used), bool(fail)

    def step_mask(self, mask):
        r, d, s = C.c_float(), C.c_int(), C.c_int()
        self.L.orc_env_step_mask(C.byref(self.e), int(mask), C.byref(r), C.byref(d), C.byref(s))
        return self.achieved_goal(), np.float32(r.value), bool(d.value), bool(s.value)


def set_action(action, block_gripper):
    a = np.ascontiguousarray(action, np.float32)
    out = np.zeros(9)
    lib().orc_set_action(_p(a, C.c_float), int(block_gripper), _p(out, C.c_double))
    return out


def compute_reward(ag, g):
    ag = np.ascontiguousarray(ag, np.float64)
    g = np.ascontiguousarray(np.broadcast_to(g, ag.shape), np.float64)
    n2 = ag.shape[-1]
    out = np.zeros(ag.shape[:-1], np.float32)
    lib().orc_compute_reward_batch(_p(ag, C.c_double), _p(g, C.c_double),
                                   int(np.prod(ag.shape[:-1], dtype=np.int64)), n2,
                                   _p(out, C.c_float))
    return out


def compute_reward_f32(ag, g):
    """float32 arrays: float32 products, NumPy's float32 summation order."""
    ag = np.ascontiguousarray(ag, np.float32)
    g = np.ascontiguousarray(np.broadcast_to(g, ag.shape), np.float32)
    out = np.zeros(ag.shape[:-1], np.float32)
    lib().orc_compute_reward_f32_batch(_p(ag, C.c_float), _p(g, C.c_float), int(np.prod(ag.shape[:-1], dtype=np.int64)),
                                       ag.shape[-1], _p(out, C.c_float), None)
    return out


def rowsum(a):
    """np.sum(a, axis=-1) as the oracle restates NumPy's pairwise order (float32 or float64 rows)."""
    a = np.ascontiguousarray(a)
    out = np.zeros(a.shape[:-1], a.dtype)
    n = int(np.prod(a.shape[:-1], dtype=np.int64))
    if a.dtype == np.float32:
        lib().orc_rowsum_f32(_p(a, C.c_float), n, a.shape[-1], _p(out, C.c_float))
    else:
        lib().orc_rowsum_f64(_p(a, C.c_double), n, a.shape[-1], _p(out, C.c_double))
    return out


def contacts_to_mask(id2obj, contacts):
    id2obj = np.ascontiguousarray(id2obj, np.int32)
    con = np.ascontiguousarray(contacts, np.int32).reshape(-1, 2)
    return int(lib().orc_contacts_to_mask(_p(id2obj, C.c_int), _p(con, C.c_int), len(con)))


def synth_masks(seed, env_ids, T, num_objs):
    """(T, B) uint32 masks of the build-defined synthetic contact source."""
    L = lib()
    env_ids = np.asarray(env_ids, np.uint64)
    num_objs = np.broadcast_to(np.asarray(num_objs, np.int32), env_ids.shape)
    out = np.zeros((T, len(env_ids)), np.uint32)
    for t in range(T):
        for b, (eid, n) in enumerate(zip(env_ids, num_objs)):
            out[t, b] = L.orc_synth_mask(int(seed) & (2**64 - 1), int(eid), t, int(n))
    return out


# policy_gradient/rollout.py:8-26 constants of the observation layout
ENV_FEATURES, BLOCK_BASE_FEATURES, COLOR_FEATURES = 10, 15, 4
BLOCK_FEATURES = BLOCK_BASE_FEATURES + COLOR_FEATURES
GREEN, BLUE = 2, 3


def trim(o, g, ag, dimo, dimg, is_variation, num_objs=4):
    """RolloutStudent.trim of the policy-gradient worker (policy_gradient/rollout.py:105-171), batched branch,
    restated with array indexing: returns (o_, g_, ag_) with the reference's dtypes.  Pinned by
    tests/golden/trim.npz (outputs of the reference's own trim)."""
    o, g, ag = np.asarray(o), np.asarray(g), np.asarray(ag)
    if o.shape[-1] == dimo:                                   # :107-108
        return o, g, ag
    assert o.ndim == 2                                        # the 1-D branch is `assert(False)`, :114-115
    max_num_objs = int(g.shape[1] ** 0.5)                     # :139
    keep = [i for i in range(g.shape[1])
            if i // max_num_objs < num_objs and i % max_num_objs < num_objs]          # :140-145
    assert len(keep) == dimg                                  # :146
    g_, ag_ = g[:, keep].copy(), ag[:, keep].copy()           # np.asarray of the gathered scalars keeps the dtype
    if is_variation:                                          # :149-168
        B = o.shape[0]
        num_blocks, max_num_blocks = num_objs - 2, max_num_objs - 2
        feats = np.empty((B, BLOCK_BASE_FEATURES * num_blocks))                       # float64, :157
        for i in range(B):
            rows = []
            for j in range(max_num_blocks):
                start = ENV_FEATURES + 1 + j * BLOCK_FEATURES
                color = int(np.argmax(o[i, start + BLOCK_BASE_FEATURES:start + BLOCK_FEATURES]))   # get_color, :24-26
                if color in (GREEN, BLUE):
                    rows.append(o[i, start:start + BLOCK_BASE_FEATURES])
            if len(rows) != num_blocks:
                raise ValueError('row %d has %d coloured blocks, the expert takes %d' % (i, len(rows), num_blocks))   # :166
            feats[i] = np.concatenate(rows) if rows else np.zeros(0)
        o_ = np.concatenate([o[:, 1:ENV_FEATURES + 1], feats], axis=1)                # float64, :167
    else:
        o_ = o[:, :dimo]                                      # :169-170
    assert o_.shape[1] == dimo                                # :171
    return o_, g_, ag_


def discounted_return(r, gamma):
    r = np.ascontiguousarray(r, np.float64)
    G = np.zeros_like(r)
    lib().orc_discounted_return(_p(r, C.c_double), len(r), float(gamma), _p(G, C.c_double))
    return G


def episode_return(reward_tb):
    """The undiscounted return of every env's episode from the (T, B) float32 step rewards: the reference's own
    accumulation of `returns[0]` (policy_gradient/rollout.py:255-258: `returns.append(r_new.copy())`, then
    `returns[0] += gamma ** t * r_new` for t = 1 .. T-1) at gamma = 1, i.e. r[0] + r[1] + ... in step order, in
    float32.  With rewards in {-1.0, -0.0} that is -(number of failed steps), and -0.0 when no step failed (a sum
    that STARTS from r[0] keeps the sign; np.sum, which starts from +0.0 for short rows, gives +0.0 there)."""
    r = np.asarray(reward_tb, np.float32)
    acc = r[0].copy()
    for t in range(1, r.shape[0]):
        acc = (acc + r[t]).astype(np.float32)
    return acc


def generate_rollouts(env_id, B, T, masks, num_grey=None, envs=None, want_reward=True):
    """Batch rollout (rollout.py:75-172).  Returns dict + the env array so a
    caller can chain rollouts (quirk Q1 persistence)."""
    L = lib()
    if envs is None:
        envs = (Env * B)()
        for b in range(B):
            L.orc_env_init(C.byref(envs[b]), VARIANTS[env_id], T)
    dimg = envs[0].side ** 2 if not envs[0].is_variation else 36
    masks = np.ascontiguousarray(masks, np.uint32)
    assert masks.shape == (T, B)
    ag = np.zeros((B, T + 1, dimg), np.float32)
    g = np.zeros((B, T, dimg), np.float32)
    succ = np.zeros((B, T, 1), np.float32)
    rew = np.zeros((T, B), np.float32)
    ng = None if num_grey is None else np.ascontiguousarray(num_grey, np.int32)
    rate = L.orc_generate_rollouts(envs, B, T, _p(masks, C.c_uint32),
                                   None if ng is None else _p(ng, C.c_int),
                                   _p(ag, C.c_float), _p(g, C.c_float), _p(succ, C.c_float),
                                   _p(rew, C.c_float) if want_reward else None)
    out = dict(ag=ag, g=g, info_is_success=succ, reward=rew, success_rate=rate)
    if want_reward:
        out['episode_return'] = episode_return(rew)
    return out, envs
