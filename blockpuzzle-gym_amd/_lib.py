"""ctypes binding of include/bpgym.h (the C-ABI drop-in boundary).

The shared library is built in-tree by `build.build()` (hipcc, gfx950).  There
is NO fallback: if the library is missing or a launch fails, the caller gets an
exception -- nothing here (or anywhere in this package) routes to a CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BPGYM_LIB: another build of the same sources (an A/B variant made with build.build(extra_flags=..., out=...)); tools only
LIB_PATH = os.environ.get('BPGYM_LIB') or os.path.join(_HERE, 'lib', 'libbpgym.so')

OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_ATTRIBUTE = 0, -1, -2, -3, -4
MAX_OBJS, NUM_PAIRS, MAX_GEOMS, MAX_MODELS = 6, 21, 64, 3
FLAG_SUCCESS, FLAG_DONE, FLAG_STEP_OK = 1, 2, 4
ARGS_MAGIC = 0xB5A90000          # BPG_ARGS_MAGIC
COMM_ID_BYTES = 128              # BPG_COMM_ID_BYTES


def args_tag(size):
    """BPG_ARGS_TAG(size): the first word of a versioned argument struct."""
    return (int(size) & 0xFFFF) | ARGS_MAGIC


class VariantInfo(C.Structure):
    _fields_ = [
        ('variant', C.c_int32), ('num_objs', C.c_int32), ('side', C.c_int32),
        ('dimg', C.c_int32), ('dimo', C.c_int32), ('dimu', C.c_int32),
        ('block_gripper', C.c_int32), ('is_variation', C.c_int32),
        ('max_episode_steps', C.c_int32), ('set_test_supported', C.c_int32),
        ('increase_difficulty_status', C.c_int32), ('desired_goal_is_float', C.c_int32),
        ('obj_colors', C.c_int32 * MAX_OBJS), ('goal', C.c_int32 * (MAX_OBJS * MAX_OBJS)),
        ('want_touch', C.c_uint32), ('want_never', C.c_uint32), ('valid_pairs', C.c_uint32),
        ('env_id', C.c_char * 48),
    ]


class RolloutArgs(C.Structure):
    """bpg_rollout_args; struct_size is filled in with this mirror's size (BPG_ROLLOUT_ARGS_INIT) unless given."""
    _fields_ = [
        ('struct_size', C.c_uint32), ('T', C.c_int32), ('reset_first', C.c_int32), ('rollouts', C.c_int32),
        ('d_num_grey', C.c_void_p),
        ('d_masks', C.c_void_p), ('d_reward_tb', C.c_void_p), ('d_flags_tb', C.c_void_p),
        ('d_ep_ag', C.c_void_p), ('d_ep_g', C.c_void_p), ('d_ep_success', C.c_void_p),
        ('d_success_count', C.c_void_p), ('d_zero_count', C.c_void_p),
        ('ld_masks', C.c_int64), ('ld_out', C.c_int64),
        ('d_G', C.c_void_p), ('gamma', C.c_double), ('d_episode_return', C.c_void_p), ('d_episode_fail_count', C.c_void_p),
        ('ld_rollouts', C.c_int64),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not args and 'struct_size' not in kw:
            self.struct_size = args_tag(C.sizeof(RolloutArgs))


ROLLOUT_ARGS_MIN_SIZE = 120


class StepArgs(C.Structure):
    """bpg_step_args; struct_size is filled in with this mirror's size (BPG_STEP_ARGS_INIT) unless given."""
    _fields_ = [
        ('struct_size', C.c_uint32), ('reserved_', C.c_uint32), ('d_masks', C.c_void_p), ('d_reward', C.c_void_p),
        ('d_flags', C.c_void_p), ('d_success_f32', C.c_void_p), ('d_ag', C.c_void_p), ('ld_ag', C.c_int64),
        ('d_u_in', C.c_void_p), ('d_u_out', C.c_void_p), ('d_success_count', C.c_void_p), ('d_return_acc', C.c_void_p),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not args and 'struct_size' not in kw:
            self.struct_size = args_tag(C.sizeof(StepArgs))


STEP_ARGS_MIN_SIZE = 88


class PhysState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ('d_grip_pos', 'd_grip_velp', 'd_gripper_state', 'd_gripper_vel',
                 'd_obj_pos', 'd_obj_rot', 'd_obj_velp', 'd_obj_velr')] + [('dt', C.c_double), ('d_nan_flag', C.c_void_p)]


class PlacementArgs(C.Structure):
    _fields_ = [('grip_x', C.c_double), ('grip_y', C.c_double), ('obj_range', C.c_double),
                ('max_obj_range', C.c_double), ('wrong_obj_range', C.c_double),
                ('test', C.c_int32), ('challenge', C.c_int32),
                ('d_eu', C.c_void_p), ('n_eu', C.c_int32), ('d_gz', C.c_void_p), ('n_gz', C.c_int32),
                ('d_gu', C.c_void_p), ('n_gu', C.c_int32), ('d_env_mask', C.c_void_p)]


class Curriculum(C.Structure):
    """bpg_curriculum"""
    _fields_ = [('difficulty', C.c_int32), ('challenge', C.c_int32), ('obj_range', C.c_double), ('obj_range_step', C.c_double),
                ('max_obj_range', C.c_double), ('wrong_obj_range', C.c_double), ('wrong_obj_range_step', C.c_double),
                ('has_step', C.c_int32), ('reserved_', C.c_int32)]


# name -> (restype, argtypes); every symbol include/bpgym.h declares
SIGNATURES = {
    'bpg_last_error': (C.c_char_p, []),
    'bpg_version': (C.c_int, []),
    'bpg_variant_from_id': (C.c_int, [C.c_char_p]),
    'bpg_get_variant_info': (C.c_int, [C.c_int, C.POINTER(VariantInfo)]),
    'bpg_geom2objid': (C.c_int, [C.c_char_p]),
    'bpg_pair_index': (C.c_int, [C.c_int, C.c_int]),
    'bpg_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_int,
                             C.c_uint64, C.c_int64]),
    'bpg_destroy': (C.c_int, [C.c_void_p]),
    'bpg_create_view': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int64]),
    'bpg_advance_generator': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    'bpg_num_envs': (C.c_int64, [C.c_void_p]),
    'bpg_seed': (C.c_int, [C.c_void_p, C.c_uint64]),
    'bpg_set_id2obj': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int8), C.c_int]),
    'bpg_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_reset_obs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_fill_static_obs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_step_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_step_contacts': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    'bpg_contacts_to_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'bpg_contacts_to_masks_tb': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p]),
    'bpg_contacts_csr_to_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'bpg_step_record_v2': (C.c_int, [C.c_void_p, C.POINTER(StepArgs), C.c_void_p]),
    'bpg_step_record': (C.c_int, []),
    'bpg_step_synth': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_synth_masks': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'bpg_synth_masks_ld': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_row_stride': (C.c_int64, [C.c_int64]),
    'bpg_get_achieved_goal_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_get_achieved_goal_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_get_desired_goal_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_assemble_obs_f64': (C.c_int, [C.c_void_p, C.POINTER(PhysState), C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_assemble_obs_f32': (C.c_int, [C.c_void_p, C.POINTER(PhysState), C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_set_action_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_mat2euler_f64': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_randomize_objects': (C.c_int, [C.c_void_p, C.POINTER(PlacementArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_state_ptrs': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'bpg_counter_ptr': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'bpg_compute_reward_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_int, C.c_void_p]),
    'bpg_compute_reward_f32_goal': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_compute_reward_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_compute_reward_f64_goal': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_compute_reward_packed': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_rollout_v2': (C.c_int, [C.c_void_p, C.POINTER(RolloutArgs), C.c_void_p]),
    'bpg_rollout': (C.c_int, []),
    'bpg_rollout_train': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(RolloutArgs), C.c_int64, C.c_int, C.c_void_p]),
    'bpg_reserve_return_scratch': (C.c_int, [C.c_void_p]),
    'bpg_episode_return_from_fail_count': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'bpg_success_count': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'bpg_discounted_return': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p,
                                        C.c_int, C.c_void_p]),
    'bpg_discounted_return_ld': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_void_p,
                                           C.c_int, C.c_void_p]),
    'bpg_discounted_return_scan': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_void_p,
                                             C.c_int, C.c_void_p]),
    'bpg_nan_flag_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    'bpg_get_curriculum': (C.c_int, [C.c_void_p, C.POINTER(Curriculum)]),
    'bpg_set_curriculum': (C.c_int, [C.c_void_p, C.POINTER(Curriculum)]),
    'bpg_get_difficulty': (C.c_int, [C.c_void_p]),
    'bpg_increase_difficulty': (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    'bpg_placement_from_curriculum': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PlacementArgs)]),
    'bpg_comm_unique_id': (C.c_int, [C.c_void_p]),
    'bpg_comm_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int]),
    'bpg_comm_adopt': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'bpg_comm_destroy': (C.c_int, [C.c_void_p]),
    'bpg_comm_rank': (C.c_int, [C.c_void_p]),
    'bpg_comm_world_size': (C.c_int, [C.c_void_p]),
    'bpg_allgather_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_allgather_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_allreduce_sum_u64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'bpg_trim_expert_inputs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
}


class BpgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('bpgym error %d: %s' % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libbpgym.so (raises if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'HIP extension %s is missing: run `python -c "import __graft_entry__ as g; '
                'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise BpgError(rc, lib().bpg_last_error().decode())
    return rc


def variant_info(variant):
    info = VariantInfo()
    check(lib().bpg_get_variant_info(int(variant), C.byref(info)))
    return info
