/*
 * bpgym.h -- C ABI of the MI355X-native batched gym_blocks step()/reset() path.
 *
 * The reference (matthew9671/BlockPuzzle-gym) has no FFI: its env boundary is
 * the Python gym.Env duck type.  This header is what a ctypes
 * binding of that boundary calls; every entry point cites the reference
 * interface it replaces (paths relative to the reference root).  Plain C types
 * only, `int` return codes (0 = BPG_OK, negative = error, text via
 * bpg_last_error()), no exceptions, no hidden host syncs in step/reset/rollout
 * (they only enqueue work on `stream`), one caller thread per handle, and the launches of one handle
 * ordered among themselves (one stream, or streams the caller orders with events): besides the packed state
 * a handle owns scratch words of the success count and the step-ok words of bpg_rollout_args.d_G.
 *
 * The boundary is "contacts-in": the closed-source physics the reference calls
 * (mujoco_py MjSim.step, robot_env.py:60) is outside; each step takes the set
 * of object pairs in contact, which is exactly what the reference extracts from
 * sim.data.contact (fetch_env.py:158-167).
 *
 * All `d_` pointers are DEVICE pointers (hipMalloc / torch tensors on the
 * handle's device).  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).
 *
 * Packed per-env state (8 bytes, two SoA uint32 arrays):
 *   now  : bits 0..20  pair (i<=j) touching this step   (achieved_goal == 1)
 *          bits 21..30 elapsed steps since reset, saturating (gym TimeLimit)
 *   ever : bits 0..20  pair has ever touched            (achieved_goal != -1)
 *          bits 21..23 num_objs of the episode (Variation: 4..6)
 *          bit  31     has_succeeded latch (fetch_env.py:80, :275-281)
 *   pair index of (i,j), i<=j<6:  i*6 - i*(i-1)/2 + (j-i)   (bpg_pair_index)
 * flags byte written per env-step: bit0 = info['is_success'], bit1 = done,
 * bit2 = the step's reward is -0.0f (success condition holds this step).
 *
 * What is NOT the reference's own code behind these entry points (third-party behaviour, absent from the reference
 * tree and not installable here: restated, and "parity unpinned" -- pinned only to the test harness's stand-ins):
 *   `done` (flags bit1, the elapsed counter): gym's TimeLimit wrapper, which the reference gets from
 *       register(max_episode_steps=50) (gym_blocks/__init__.py:6-53) -- restated as done = elapsed_steps >= max_episode_steps,
 *       counter reset by reset(), saturating at 1023;
 *   bpg_mat2euler_f64: gym.envs.robotics.rotations.mat2euler;
 *   the random streams behind bpg_seed / generator-drawn num_grey and placements: gym's seeding (SHA-512 -> MT19937) is
 *       replaced by a counter-based generator; only the seed ARITHMETIC (seed + 1000 * idx, rollout.py:209-210) is mirrored.
 * Outside SURVEY.md section 8 (imitation-learning glue of the pg worker, kept from round 3, frozen): bpg_trim_expert_inputs.
 */
#ifndef BPGYM_H
#define BPGYM_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPG_OK               0
#define BPG_ERR_ARG         -1
#define BPG_ERR_HIP         -2
#define BPG_ERR_UNSUPPORTED -3   /* mirrors the reference's NotImplementedError  */
#define BPG_ERR_ATTRIBUTE   -4   /* mirrors the reference's AttributeError (Q6)  */

#define BPG_MAX_OBJS   6
#define BPG_NUM_PAIRS  21
#define BPG_MAX_GEOMS  64
#define BPG_MAX_MODELS 3
#define BPG_FLAG_SUCCESS 1
#define BPG_FLAG_DONE    2
#define BPG_FLAG_STEP_OK 4   /* this step's reward is -0.0f (else -1.0f): flags alone encode the reward */

/* env ids registered by gym_blocks/__init__.py:6-53, in registration order */
enum {
    BPG_GRIPPER_TOUCH = 0,
    BPG_BLOCKS_TOUCH = 1,
    BPG_TOPPLE_TOWER = 2,
    BPG_BLOCKS_TOUCH_CURRICULUM = 3,
    BPG_BLOCKS_TOUCH_CHOOSE = 4,
    BPG_BLOCKS_TOUCH_CHOOSE_CURRICULUM = 5,
    BPG_BLOCKS_TOUCH_VARIATION = 6,
    BPG_NUM_VARIANTS = 7
};

typedef struct bpg_batch bpg_batch;     /* opaque: one batch of envs on one GPU */

/* Versioned argument structs (bpg_step_args, bpg_rollout_args): the first word is BPG_ARGS_TAG(sizeof(the struct as the
 * caller compiled it)) -- the size in the low 16 bits, a magic number in the high 16, so that a struct of another layout
 * (a caller built against a header that had no struct_size word there: T or a pointer sits in its place) can never
 * validate.  Fields are only ever appended; a library that knows more fields than the caller passed treats the missing ones
 * as zero, a library that knows fewer accepts the call only if the bytes it does not know are all zero
 * (BPG_ERR_UNSUPPORTED otherwise).  The entry points that take them carry a _v2 suffix: the round-3 names bpg_step_record
 * (ten positional arguments) and bpg_rollout (a struct without the size word) still exist and return BPG_ERR_UNSUPPORTED
 * without touching their arguments, so a stale caller gets an error code, not a crash.
 * `bpg_rollout_args a = BPG_ROLLOUT_ARGS_INIT;` for automatic variables (C and C++, no warnings under -Wall -Wextra
 * -pedantic); `static bpg_rollout_args a = BPG_ROLLOUT_ARGS_INIT_STATIC;` is a constant initialiser (C statics / globals). */
#define BPG_ARGS_MAGIC 0xB5A90000u
#define BPG_ARGS_TAG(size) ((uint32_t)(size) | BPG_ARGS_MAGIC)

/* Static facts about a variant (tasks.py presets + fetch_env.py class consts). */
typedef struct {
    int32_t variant;
    int32_t num_objs;        /* fetch_env.py:75 (Variation: max, 6)            */
    int32_t side;            /* achieved_goal matrix side (Variation always 6) */
    int32_t dimg;            /* side*side                                      */
    int32_t dimo;            /* 10 + 15*(num_objs-2); Variation 87 (:567-621)  */
    int32_t dimu;            /* 4 (robot_env.py:39)                            */
    int32_t block_gripper;   /* tasks.py presets                               */
    int32_t is_variation;
    int32_t max_episode_steps;        /* 50 for every id (__init__.py)         */
    int32_t set_test_supported;       /* fetch_env.py:100-101 vs :365,:443,:641 */
    int32_t increase_difficulty_status; /* BPG_OK / _UNSUPPORTED / _ATTRIBUTE (Q6) */
    int32_t desired_goal_is_float;    /* Q9: Variation float64, others int64   */
    int32_t obj_colors[BPG_MAX_OBJS]; /* _sample_colors, constant per variant  */
    int32_t goal[BPG_MAX_OBJS * BPG_MAX_OBJS]; /* _sample_goal :260-273/:682-695 */
    uint32_t want_touch;     /* packed goal planes: pairs with goal +1 / -1    */
    uint32_t want_never;
    uint32_t valid_pairs;    /* all pairs i<=j<side                            */
    char env_id[48];
} bpg_variant_info;

const char *bpg_last_error(void);
int bpg_version(void);

/* gym.make(id) name lookup (gym_blocks/__init__.py:6-53). -1 if not registered. */
int bpg_variant_from_id(const char *env_id);
int bpg_get_variant_info(int variant, bpg_variant_info *out);

/* BlocksEnv._geom2objid (fetch_env.py:106-117): geom name -> object id, -1 = None (NULL or unnamed geom).
 * Names the reference's `int(name[6:])` would raise ValueError on -- "object" with nothing or a non-digit behind the
 * first six characters, e.g. "object", "myobject3", "objectA" -- cannot occur in the shipped XML models; here the leading
 * decimal digits behind position 6 are read as C's strtol does and a missing number counts as 0 (-> object id 2)
 * instead of raising. */
int bpg_geom2objid(const char *name);
/* pair index used by every packed mask. */
int bpg_pair_index(int i, int j);

/* rollout.py:33 `[make_env() for _ in range(B)]` -> one handle for n_envs envs
 * on `device_id`.  env_id_base offsets the global env index of env 0 (multi-GPU
 * shards, mirrors seed + 1000*idx of rollout.py:209-210).  State starts as the
 * constructor leaves it (fetch_env.py:78-80): matrix all -1, not succeeded. */
int bpg_create(bpg_batch **out, int variant, int64_t n_envs, int device_id,
               int max_episode_steps, uint64_t seed, int64_t env_id_base);
int bpg_destroy(bpg_batch *b);
int64_t bpg_num_envs(const bpg_batch *b);
/* A handle over envs first_env .. first_env + n_envs - 1 of `parent` -- the same partition of the env index range that spreads a
 * job over GPUs, applied inside one GPU: launches on views of disjoint ranges are independent and may run side by
 * side on different streams (while one launch ramps up or drains the others stream).  A view SHARES the parent's
 * packed state (its launches read and write the parent's arrays in place: whatever the views did is what the parent sees), its
 * seed, id2obj tables, curriculum and generator stream indices, and OWNS the scratch a launch needs (success-count words,
 * step-ok words).  Env e of the view is env first_env + e of the parent in every respect (generator keys included).
 * first_env % 4 == 0.  bpg_seed / bpg_set_id2obj / the curriculum calls go to the parent; a generator-drawn reset through a view
 * does not advance the generator's stream index -- call bpg_advance_generator on the parent once after the views' launches.
 * Destroy the views before the parent. */
int bpg_create_view(bpg_batch **out, bpg_batch *parent, int64_t first_env, int64_t n_envs);
/* Advances the handle's generator stream indices (device words, stream-ordered): `resets` generator-drawn resets and
 * `placements` generator-drawn placements have happened.  Only needed after launches through views. */
int bpg_advance_generator(bpg_batch *b, uint32_t resets, uint32_t placements, void *stream);

/* RobotEnv.seed (robot_env.py:53-55) / RolloutStudent.seed (rollout.py:206-210):
 * re-keys the counter-based generator (synthetic contacts, num_grey draw) and restarts its
 * stream indices.  SYNCHRONOUS (device-wide): work already enqueued on any stream keeps the old key. */
int bpg_seed(bpg_batch *b, uint64_t seed);

/* id2obj table of one model (fetch_env.py:284, Variation :651): host array of
 * `ngeom` object ids (-1 = None).  model_idx 0 except Variation (0..2 =
 * 2/3/4-block models, chosen per episode by num_objs-4).  SYNCHRONOUS (device-wide), like
 * bpg_seed: a setup call, not part of the step path. */
int bpg_set_id2obj(bpg_batch *b, int model_idx, const int8_t *id2obj, int ngeom);

/* RobotEnv.reset (robot_env.py:71-82) + _reset_sim (fetch_env.py:247-255;
 * Variation :646-679) + TimeLimit.reset for every env with d_env_mask[b] != 0
 * (NULL = all).  Base variants keep the touch matrix (quirk Q1); Variation
 * clears it and takes num_objs = 4 + num_grey, with num_grey from
 * d_num_grey[b] (explicit variate for np_random.randint(3), :647) or, when
 * NULL, from the handle's counter-based generator. */
int bpg_reset(bpg_batch *b, const int32_t *d_num_grey, const uint8_t *d_env_mask, void *stream);
/* The same reset, and in the SAME launch the observation it leaves behind, for the envs being reset:
 * d_obs (n_envs rows of dimo float32, row stride ld) gets what the reference's _get_obs fills without
 * physics (Variation, fetch_env.py:579-607: obs[0] = num_blocks, colour one-hot of every present
 * block, zeros in the slots of absent blocks) and ZEROS in every physics-owned entry -- there is no
 * physics state behind the contacts-in boundary; a backend writes its own values afterwards
 * (bpg_assemble_obs_*, from its on_reset).  Rows of envs outside d_env_mask are not touched. */
int bpg_reset_obs(bpg_batch *b, const int32_t *d_num_grey, const uint8_t *d_env_mask, float *d_obs,
                  int64_t ld, void *stream);
/* Only the physics-free entries, from the state as it stands (after construction or a restored
 * checkpoint), for every env; the physics-owned entries of d_obs are left as they are. */
int bpg_fill_static_obs(bpg_batch *b, float *d_obs, int64_t ld, void *stream);

/* RobotEnv.step (robot_env.py:57-69) after the physics: _step_callback
 * (fetch_env.py:154-167) + _is_success latch (:275-281) + compute_reward
 * (:135-143) + TimeLimit done, fused in one kernel.  The step's contacts are
 * given as packed pair masks d_masks[n_envs].  d_reward (float32, -1.0f /
 * -0.0f) and d_flags (uint8) may each be NULL. */
int bpg_step_masks(bpg_batch *b, const uint32_t *d_masks, float *d_reward,
                   uint8_t *d_flags, void *stream);

/* Same step with the contacts as the reference reads them: per env a list of
 * geom-id pairs (sim.data.contact[i].geom1/geom2, fetch_env.py:159-162),
 * d_contacts[n_envs][maxcon][2] int16, d_ncon[n_envs] valid entries each;
 * geoms are mapped through the id2obj table of the env's current model. */
int bpg_step_contacts(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon,
                      int maxcon, float *d_reward, uint8_t *d_flags, void *stream);

/* The id2obj half of _step_callback on its own (fetch_env.py:158-166): geom-id pair lists ->
 * packed pair masks d_masks[n_envs], so a backend that emits contact lists can drive the
 * mask-based kernels (bpg_step_record, bpg_rollout). */
int bpg_contacts_to_masks(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                          uint32_t *d_masks, void *stream);

/* The lists of a whole recorded rollout at once -- d_contacts (T, n_envs, maxcon, 2) int16 and d_ncon (T, n_envs), time-major,
 * as a physics trace stores sim.data.contact of every step -- into the (T, ld_masks) mask stream bpg_rollout reads
 * (ld_masks = 0: contiguous rows).  d_num_grey (n_envs, optional): Variation picks the sim, and with it the id2obj table,
 * of an episode by num_objs = 4 + num_grey (fetch_env.py:647-651); pass the draws of the episode the trace belongs to when
 * the conversion runs BEFORE that episode's reset (bpg_rollout with reset_first), NULL = the model of the current state. */
int bpg_contacts_to_masks_tb(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int T, int maxcon,
                             const int32_t *d_num_grey, uint32_t *d_masks, int64_t ld_masks, void *stream);

/* The same from RAGGED contact lists (CSR): env e owns contacts d_offsets[e] .. d_offsets[e+1]-1
 * (int64, n_envs+1 entries, non-decreasing) of one packed d_pairs[total][2] int16 geom-id array --
 * sim.data.contact of every env compacted instead of padded to the longest list.  Offsets are
 * clamped to 0..total, so a malformed table cannot index outside d_pairs. */
int bpg_contacts_csr_to_masks(bpg_batch *b, const int16_t *d_pairs, const int64_t *d_offsets, int64_t total,
                              uint32_t *d_masks, void *stream);

/* One env.step exactly as the rollout worker consumes it (rollout.py:121-131, :144-150), in ONE
 * launch: the step of bpg_step_masks plus whatever of the following is asked for (every pointer is optional, NULL = not
 * produced; d_u_in / d_u_out go together).  A versioned struct (see BPG_ARGS_TAG): BPG_STEP_ARGS_INIT sets struct_size. */
typedef struct {
    uint32_t struct_size;      /* BPG_ARGS_TAG(sizeof(bpg_step_args)) */
    uint32_t reserved_;        /* 0 */
    const uint32_t *d_masks;   /* (n_envs) packed pair masks of the step; NULL = the synthetic source */
    float *d_reward;           /* (n_envs) float32 step reward, -1.0f / -0.0f */
    uint8_t *d_flags;          /* (n_envs) flags byte */
    float *d_success_f32;      /* (n_envs) info['is_success'] as float32 (rollout.py:127-131) */
    float *d_ag;               /* obs['achieved_goal'] as float32 rows at d_ag + e * ld_ag (fetch_env.py:219-228): a dense
                                  (n_envs, dimg) array or the time slot of an episode buffer */
    int64_t ld_ag;             /* row stride of d_ag in elements (>= dimg) */
    const float *d_u_in;       /* (n_envs, 4) float32 actions of the step ... */
    float *d_u_out;            /* ... stored into the episode's action slot (`acts.append(u.copy())`, rollout.py:147) */
    unsigned long long *d_success_count; /* += the number of envs whose has_succeeded latch is set after this step --
                                  `successful = successes[-1]` (rollout.py:163-167) inside the LAST step's launch of a
                                  rollout instead of a bpg_success_count launch of its own.  8-byte aligned */
    float *d_return_acc;       /* (n_envs) float32, d_return_acc[e] += the step reward: the per-env undiscounted episode
                                  return of a rollout that runs one launch per step (a policy between the steps), the
                                  per-step counterpart of bpg_rollout_args.d_episode_return.  Initialise it to -0.0f
                                  before the first step for the same r[0] + r[1] + ... convention (-0.0f when every
                                  step succeeded) */
} bpg_step_args;
#define BPG_STEP_ARGS_MIN_SIZE 88u
static inline bpg_step_args bpgh_step_args_init(void)
{
    bpg_step_args a;
    memset(&a, 0, sizeof(a));
    a.struct_size = BPG_ARGS_TAG(sizeof(a));
    return a;
}
#define BPG_STEP_ARGS_INIT bpgh_step_args_init()
#define BPG_STEP_ARGS_INIT_STATIC { BPG_ARGS_TAG(sizeof(bpg_step_args)), 0u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }
int bpg_step_record_v2(bpg_batch *b, const bpg_step_args *args, void *stream);
/* the round-3 entry point of that name took ten positional arguments: kept as a stub that returns BPG_ERR_UNSUPPORTED */
int bpg_step_record(void);

/* One step whose contacts come from the build-defined counter-based synthetic
 * source keyed (seed, global env id, elapsed step) -- physics stand-in for
 * throughput runs, no reference counterpart. */
int bpg_step_synth(bpg_batch *b, float *d_reward, uint8_t *d_flags, void *stream);

/* Fills d_masks[T][n_envs] (time-major) with that synthetic source for steps
 * t0 .. t0+T-1 of the current episode. */
int bpg_synth_masks(bpg_batch *b, int t0, int T, uint32_t *d_masks, void *stream);
/* Same with a row stride: row t of the stream starts ld elements after row t-1 (ld >= n_envs). */
int bpg_synth_masks_ld(bpg_batch *b, int t0, int T, uint32_t *d_masks, int64_t ld, void *stream);

/* Row stride (in elements) this build recommends for the time-major (T, ld) streams of an
 * n-env batch -- masks, rewards, flags (bpg_rollout_args.ld_masks / ld_out).  Build-defined
 * layout choice, no reference counterpart (the reference has no time-major arrays; its episode
 * arrays are batch-major, rollout.py:155-172): rows a power of two apart land on the same HBM
 * channels, rows shifted by ~1 KB each do not.  Pure host function. */
int64_t bpg_row_stride(int64_t n_envs);

/* obs['achieved_goal'] / obs['desired_goal'] of _get_obs (fetch_env.py:219-228,
 * Variation :609-621) for the whole batch: env e's dimg values are written at
 * d_out + e*ld (ld = dimg for a dense (n_envs, dimg) array; a larger ld writes
 * straight into the time slot of a batch-major episode buffer, rollout.py:149-150). */
int bpg_get_achieved_goal_f32(bpg_batch *b, float *d_out, int64_t ld, void *stream);
int bpg_get_achieved_goal_f64(bpg_batch *b, double *d_out, int64_t ld, void *stream);
int bpg_get_desired_goal_f32(bpg_batch *b, float *d_out, int64_t ld, void *stream);

/* obs['observation'] of _get_obs (fetch_env.py:187-228; Variation :567-621) assembled
 * from what the reference reads off the simulator -- all float64 device arrays, as
 * mujoco_py returns them, nb = num_objs-2 block slots per env (Variation: 4):
 *   d_grip_pos (n,3)  get_site_xpos('robot0:grip')     d_grip_velp (n,3) get_site_xvelp, unscaled
 *   d_gripper_state (n,2) robot_qpos[-2:]              d_gripper_vel (n,2) robot_qvel[-2:], unscaled
 *   d_obj_pos / d_obj_rot / d_obj_velp / d_obj_velr (n,nb,3): site pos, mat2euler(site xmat),
 *   site linear / angular velocity (unscaled);  dt = nsubsteps * timestep (fetch_env.py:190).
 * Layout, dt scaling, relative quantities, Variation's num_blocks / colour one-hots /
 * zero padding are produced here; the _f64 form is bit-identical to the reference. */
typedef struct {
    const double *d_grip_pos, *d_grip_velp, *d_gripper_state, *d_gripper_vel;
    const double *d_obj_pos, *d_obj_rot, *d_obj_velp, *d_obj_velr;
    double dt;
    uint32_t *d_nan_flag;   /* optional: *d_nan_flag |= 1 when any value written is a NaN -- the rollout
                               worker's np.isnan(o_new).any() (rollout.py:139-142) inside the launch that
                               produces o_new, instead of a second pass over it (bpg_nan_flag_f32) */
} bpg_phys_state;
int bpg_assemble_obs_f64(bpg_batch *b, const bpg_phys_state *ph, double *d_obs, int64_t ld, void *stream);
int bpg_assemble_obs_f32(bpg_batch *b, const bpg_phys_state *ph, float *d_obs, int64_t ld, void *stream);

/* robot_env.py:58 (clip to the action Box) + BlocksEnv._set_action (fetch_env.py:170-185):
 * d_actions (n,4) float32 -> d_out9 (n,9) float64 = [pos_ctrl*0.05 (float32 product), the fixed
 * quaternion 1,0,1,0, gripper target twice (zeros when block_gripper)], i.e. the vector the
 * reference passes to gym's ctrl_set_action / mocap_set_action -- the physics backend's input. */
int bpg_set_action_f64(bpg_batch *b, const float *d_actions, double *d_out9, void *stream);

/* gym.envs.robotics.rotations.mat2euler (call sites fetch_env.py:205, :588): n row-major 3x3
 * site matrices -> n euler triples for bpg_phys_state.d_obj_rot.  THIRD-PARTY code absent
 * from the reference tree, restated from memory of gym 0.10 -- parity UNPINNED. */
int bpg_mat2euler_f64(const double *d_mat, int64_t n, double *d_euler, int device_id, void *stream);

/* _randomize_objects of the five env classes (fetch_env.py:328-336, 370-399, 448-517,
 * 697-764, 777-787): rejection-sampled cube positions for every env, d_xy (n, nb, 2) float64
 * (nb = num_objs-2 block slots; Variation leaves absent blocks untouched).  The output feeds
 * the physics backend (object qpos[:2]).  Curriculum state is passed in (host attributes in
 * the reference): obj_range / max_obj_range / wrong_obj_range, test (set_test / test rollouts
 * use the maximum ranges), challenge; grip_x/grip_y = initial_gripper_xpos[:2] (physics).
 * Random draws: either explicit STANDARD variates in the reference's consumption order --
 * d_eu (n, n_eu) random_sample() values of the per-env np_random stream, d_gz (n, n_gz)
 * standard normals and d_gu (n, n_gu) random_sample() values of the global np.random
 * stream -- or, with all three NULL, the handle's counter-based generator.
 * d_status (n) int32, optional: 0 ok, 1 = variates exhausted / iteration cap reached. */
typedef struct {
    double grip_x, grip_y;
    double obj_range, max_obj_range, wrong_obj_range;
    int32_t test, challenge;
    const double *d_eu; int32_t n_eu;
    const double *d_gz; int32_t n_gz;
    const double *d_gu; int32_t n_gu;
    const uint8_t *d_env_mask;   /* partial resets (reset_rollout(i), rollout.py:47-58): only envs with a
                                    non-zero byte are placed, the other rows of d_xy stay; NULL = all */
} bpg_placement_args;
int bpg_randomize_objects(bpg_batch *b, const bpg_placement_args *a, double *d_xy, int32_t *d_status,
                          void *stream);

/* Raw packed state, device pointers owned by the handle (zero-copy views). */
int bpg_state_ptrs(bpg_batch *b, uint32_t **d_now, uint32_t **d_ever);
/* The handle's two generator stream indices (device uint32[2]: generator-drawn resets,
 * generator-drawn placements so far); together with the packed state this is everything a
 * checkpoint of the batch needs (the reference never checkpoints env state). */
int bpg_counter_ptr(bpg_batch *b, uint32_t **d_counters);

/* BlocksEnv.compute_reward (fetch_env.py:135-143), stateless and vectorised as HER calls it
 * (config.py:110-111): n rows of `len` floats each, len in 1..256; reward = -(sum(ag * g) != count_nonzero(g)).
 * Every row the env or a replay buffer can hold is ternary ({-1, 0, 1}: any summation order is exact); for
 * arbitrary rows the sum is taken in NumPy's order for that precision -- products rounded on their own, then the
 * pairwise add-reduction (third-party NumPy; pinned by tests/golden/compute_reward_order.npz) -- so the result
 * is NumPy's for float32 arrays (_f32) and for float64 arrays (_f64), not that of a mixed call. */
int bpg_compute_reward_f32(const float *d_ag, const float *d_g, int64_t n, int len,
                           float *d_out, int device_id, void *stream);
/* Same against ONE goal row of `len` floats shared by every row of d_ag -- env.compute_reward(ag,
 * env.goal) with NumPy broadcasting (fetch_env.py:135-143; the goal is constant per variant, :260-273). */
int bpg_compute_reward_f32_goal(const float *d_ag, const float *d_goal_row, int64_t n, int len,
                                float *d_out, int device_id, void *stream);
/* float64 rows: what env.step itself passes (achieved_goal float64, goal int64 or float64 -> float64 products
 * and sum, fetch_env.py:141-143 / robot_env.py:66). */
int bpg_compute_reward_f64(const double *d_ag, const double *d_g, int64_t n, int len,
                           float *d_out, int device_id, void *stream);
int bpg_compute_reward_f64_goal(const double *d_ag, const double *d_goal_row, int64_t n, int len,
                                float *d_out, int device_id, void *stream);
/* Same on packed states against packed goal planes. */
int bpg_compute_reward_packed(const uint32_t *d_now, const uint32_t *d_ever,
                              uint32_t want_touch, uint32_t want_never, int64_t n,
                              float *d_out, int device_id, void *stream);

/* OUTSIDE SURVEY.md section 8 (section 2 #6 marks expert mixing / trim "imitation-learning glue"): built in round 3, pinned
 * by fixtures, frozen since.
 * RolloutStudent.trim of the policy-gradient worker (policy_gradient/rollout.py:105-171, batched branch): the inputs
 * of an expert trained on a smaller puzzle (`self.expert.get_actions(o_, ag_, g_, compute_raw=True)`, :196-200).
 *   d_o (n, dimo_in), d_g / d_ag (n, dimg_in) float32 rows, contiguous.
 *   d_g_out / d_ag_out (n, dimg_out) float32: entries (i, j), i, j < num_objs, of the sqrt(dimg_in)-sided matrices (:139-151);
 *     dimg_out must be their number (the reference's assert, :146).
 *   is_variation != 0 (:152-168): d_o_out (n, dimo_out) FLOAT64 = o[1:11] followed by the 15 base features of every
 *     block slot whose colour one-hot peaks (np.argmax) at GREEN or BLUE, slot order; dimo_out must be 10 + 15 *
 *     (num_objs - 2) (:171).  A row with another number of such slots makes NumPy raise in the reference (:166): bit 0 of
 *     *d_bad_flag is set (the caller clears it; read it where the rollout is synchronised anyway) and the row's
 *     missing picks repeat slot 0.
 *   is_variation == 0 (:169-170): d_o_out (n, dimo_out) FLOAT32 = o[:, :dimo_out]; d_bad_flag may be NULL.
 * Callers skip the call when dimo_in == dimo_out (the reference returns its inputs, :107-108).  dimg_in <= 64. */
int bpg_trim_expert_inputs(const float *d_o, const float *d_g, const float *d_ag, int64_t n, int dimo_in, int dimg_in,
                           int dimo_out, int dimg_out, int num_objs, int is_variation, void *d_o_out, float *d_g_out,
                           float *d_ag_out, uint32_t *d_bad_flag, int device_id, void *stream);

/* RolloutStudent.generate_rollouts inner loop (rollout.py:100-153) for T steps
 * in ONE launch: state stays in registers, every array below is optional
 * (NULL = not produced).  With reset_first the launch also performs the
 * reset_all_rollouts of rollout.py:79; otherwise call bpg_reset first.
 * A versioned struct (see BPG_ARGS_TAG): BPG_ROLLOUT_ARGS_INIT sets struct_size; a size below BPG_ROLLOUT_ARGS_MIN_SIZE or a
 * first word without the tag is BPG_ERR_ARG. */
typedef struct {
    uint32_t struct_size;      /* BPG_ARGS_TAG(sizeof(bpg_rollout_args)) as the caller compiled it */
    int32_t T;
    int32_t reset_first;       /* 1: reset every env (bpg_reset semantics) inside the launch, before step 0 */
    int32_t rollouts;          /* 0 / 1: one rollout.  K > 1 (stream path, reset_first = 1, no d_ep_*, no d_G): K consecutive
                                  rollouts in this ONE launch -- generate_rollouts called K times in a row with nothing
                                  between the calls (rollout.py:75-172) -- rollout r = reset + T steps on rows r*T .. r*T+T-1
                                  of d_masks / d_reward_tb / d_flags_tb (K*T rows each), its success count added to
                                  d_success_count[r] (K words), row r of d_episode_return, explicit d_num_grey (K, n_envs) or
                                  the generator's next K draws.  Same results as K calls; the fixed cost of a launch is paid
                                  once. */
    const int32_t *d_num_grey; /* reset_first on Variation: explicit num_grey draws, NULL = generator */
    const uint32_t *d_masks;   /* (T, n_envs) time-major contact masks; NULL = synthetic source */
    float *d_reward_tb;        /* (T, n_envs) float32 step rewards                */
    uint8_t *d_flags_tb;       /* (T, n_envs) flags bytes                         */
    float *d_ep_ag;            /* (n_envs, T+1, dimg) episode['ag'] batch-major   */
    float *d_ep_g;             /* (n_envs, T, dimg)   episode['g']                */
    float *d_ep_success;       /* (n_envs, T, 1)      episode['info_is_success']  */
    unsigned long long *d_success_count; /* += #envs with has_succeeded after step T (rollout.py:163-167) */
    unsigned long long *d_zero_count;    /* optional second counter the launch sets to 0 (ping-pong) */
    int64_t ld_masks;          /* row stride of d_masks in elements; 0 = n_envs (contiguous rows) */
    int64_t ld_out;            /* row stride of d_reward_tb and d_flags_tb in elements; 0 = n_envs */
    double *d_G;               /* (n_envs, T) float64 batch-major: the policy-gradient return of the rollout
                                  (policy_gradient/rollout.py:217-231, :255-258), bit-identical to the reference's
                                  float64 accumulation, WITHOUT a reward stream: for T <= 64 and 0 <= gamma with finite
                                  powers the rollout launch keeps each env's step rewards as one 64-bit step-ok word
                                  (8 bytes per env, handle-owned: bpg_reserve_return_scratch, or the first call that asks
                                  for d_G, which must then be outside a stream capture) and a second launch enqueued by the same call
                                  expands the words into the rows; 64 < T <= 319, or a gamma that is negative or whose
                                  powers overflow: computed from d_reward_tb, which is then required.  8-byte aligned
                                  (16 for the fast T = 50 / 10 forms).  NULL = not produced */
    double gamma;              /* discount of d_G (config.py:83: 1 - 1/T) */
    float *d_episode_return;   /* (n_envs) float32, or (rollouts, n_envs) with rollouts = K: every env's UNDISCOUNTED episode
                                  return r[0] + r[1] + ... + r[T-1] of the float32 step rewards, accumulated in step order
                                  starting from r[0] -- the reference's own accumulation of returns[0]
                                  (policy_gradient/rollout.py:255-258) at gamma = 1 -- = -(number of steps whose success
                                  condition did not hold), and -0.0f when it held at every step (a sum that starts from r[0]
                                  keeps the sign of zero; numerically equal to np.sum(rewards, axis=0)).  BASELINE.json's
                                  "aggregated rollout return": 4 bytes per env from a per-env counter in registers, no reward
                                  stream written or read for it.  4-byte aligned.  NULL = not produced */
    uint8_t *d_episode_fail_count; /* (n_envs) uint8, or (rollouts, n_envs): the same quantity as ONE BYTE per env -- the
                                  number of steps of the rollout whose success condition did not hold, so that
                                  d_episode_return[e] == -(float)d_episode_fail_count[e] bit for bit (-0.0f for 0).  Needs
                                  T <= 255.  The array a multi-GPU job exchanges (all-gather): a quarter of
                                  the float32 form's bytes on the wire, expanded on the receiving side by
                                  bpg_episode_return_from_fail_count.  NULL = not produced */
    int64_t ld_rollouts;       /* rollouts = K > 1: row stride in elements of the (K, n_envs) arrays d_num_grey, d_episode_return,
                                  d_episode_fail_count -- 0 = n_envs; larger when the handle is a view (bpg_create_view) of a batch
                                  whose (K, total) arrays it owns a column range of */
} bpg_rollout_args;
#define BPG_ROLLOUT_ARGS_MIN_SIZE 120u   /* everything up to and including gamma (low 16 bits of struct_size) */
/* `bpg_rollout_args a = BPG_ROLLOUT_ARGS_INIT;` -- struct_size set, every other field zero (C and C++, no warnings under
 * -Wall -Wextra -pedantic).  bpgh_* are header-only helpers, not symbols of the library. */
static inline bpg_rollout_args bpgh_rollout_args_init(void)
{
    bpg_rollout_args a;
    memset(&a, 0, sizeof(a));
    a.struct_size = BPG_ARGS_TAG(sizeof(a));
    return a;
}
#define BPG_ROLLOUT_ARGS_INIT bpgh_rollout_args_init()
#define BPG_ROLLOUT_ARGS_INIT_STATIC { BPG_ARGS_TAG(sizeof(bpg_rollout_args)), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0 }
int bpg_rollout_v2(bpg_batch *b, const bpg_rollout_args *args, void *stream);
/* the round-3 entry point of that name took a struct without the size word: kept as a stub that returns BPG_ERR_UNSUPPORTED */
int bpg_rollout(void);
/* A TRAIN of fused rollouts over stream shards in one call: launch l (l < n_launches) of shard s (s < n_shards) is
 * bpg_rollout_v2(handles[s], &args[l * n_shards + s], streams[s]) -- args is an array of n_launches * n_shards structs,
 * args_stride_bytes apart (the caller's sizeof(bpg_rollout_args)), launch-major.  The shards' streams are forked from
 * caller_stream in front of the train and joined into it behind it (one event and n_shards waits each way; the events are the
 * handles' own, created on first use), and nothing else is synchronised: the launches of different shards run side by side.  The
 * handles are views of one batch (bpg_create_view) or batches of their own, one stream each, all on one device.  This is the
 * evaluator's loop -- `for _ in range(n_test_rollouts): evaluator.generate_rollouts()` (train.py:63-64) -- for a batch split into
 * shards, without a host round trip per launch: the host cost of a launch is that of the kernel launch itself. */
int bpg_rollout_train(bpg_batch *const *handles, void *const *streams, int n_shards, const bpg_rollout_args *args,
                      int64_t args_stride_bytes, int n_launches, void *caller_stream);
/* The step-ok words behind bpg_rollout_args.d_G (8 bytes per env) are handle-owned scratch that the first call asking for d_G
 * would otherwise allocate with a device-synchronising hipMalloc in the middle of the stream (and could not at all inside a
 * stream capture): allocate them up front, once, from set-up code.  Idempotent. */
int bpg_reserve_return_scratch(bpg_batch *b);
/* d_out[e] = -(float)d_count[e] (-0.0f for 0): bpg_rollout_args.d_episode_fail_count back into d_episode_return's float32
 * form, e.g. after an all-gather of the byte form.  n elements on `device_id`. */
int bpg_episode_return_from_fail_count(const uint8_t *d_count, float *d_out, int64_t n, int device_id, void *stream);

/* rollout.py:163-167 `successful = successes[-1]`: number of envs whose
 * has_succeeded latch is set, added to *d_count (uint64, caller zeroes it). */
int bpg_success_count(bpg_batch *b, unsigned long long *d_count, void *stream);

/* policy_gradient/rollout.py:255-258: G[b][t_] = r[t_] + sum_{t>t_}
 * gamma**(t-t_) * r[t], float64, accumulated in the reference's order.
 * d_reward_tb (T, n) float32 time-major in, d_G (n, T) float64 batch-major out.
 * T in 1..319 (the 64-env tile of returns has to fit the 160 KB LDS); the gamma**k table travels
 * in the kernel arguments, so the call only enqueues a launch (hipGraph-capturable). */
int bpg_discounted_return(const float *d_reward_tb, int64_t n, int T, double gamma,
                          double *d_G, int device_id, void *stream);
/* Same for a reward stream with row stride ld >= n (the padded rows bpg_rollout writes).
 * T <= 64: one env per lane, the T returns in registers (T = 50 / 25 / 10 have instantiations of their own, every other
 * horizon runs the 32- or 64-accumulator form); a wave whose rewards are all -1.0f / -0.0f (the only values
 * compute_reward produces, fetch_env.py:141-143) replaces product + sum by a conditional subtraction of
 * gamma**k -- exact for those rewards, same bits --, any other wave runs the general product + sum.
 * 64 < T <= 319: the returns accumulate in an LDS tile. */
int bpg_discounted_return_ld(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma,
                             double *d_G, int device_id, void *stream);
/* the same return as an O(T) reverse scan G[t] = fma(gamma, G[t+1], r[t]) instead of the
 * reference's O(T^2) loop (policy_gradient/rollout.py:255-258).  NOT the reference's summation order, so
 * tolerance-based:  |G_scan[t] - G_reference[t]| <= 4 T 2^-53 * sum_{k>=0} gamma^k |r[t+k]|  (0 <= gamma <= 1).
 * bpg_discounted_return stays the default (bit-exact).  ld = row stride of d_reward_tb (0 = n). */
int bpg_discounted_return_scan(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma,
                               double *d_G, int device_id, void *stream);

/* `np.isnan(o_new).any()` of the rollout worker (rollout.py:139-142) as a device-side flag:
 * *d_flag |= 1 when any of the rows x cols float32 values at d_x (row stride ld) is a NaN.  The
 * worker reads the flag once per rollout instead of synchronising every step. */
int bpg_nan_flag_f32(const float *d_x, int64_t rows, int cols, int64_t ld, uint32_t *d_flag,
                     int device_id, void *stream);

/* ---- curriculum (fetch_env.py:340-358, :404-432, :561-563, :623-630; tasks.py presets) ----------------------------------
 * The ranges _randomize_objects samples from and the difficulty counter are per-env-object attributes in the reference;
 * one handle = one batch of identical envs (rollout.py:66-73 raises the difficulty of every env alike), so the handle holds
 * ONE copy.  bpg_create initialises it as the variant's constructor does. */
typedef struct {
    int32_t difficulty;              /* get_difficulty() (fetch_env.py:96-97) */
    int32_t challenge;               /* BlocksTouchChoose*: self.challenge (:417) */
    double obj_range, obj_range_step, max_obj_range;
    double wrong_obj_range, wrong_obj_range_step;
    int32_t has_step;                /* 0: the preset has no *_step attributes -> increase_difficulty raises AttributeError (Q6) */
    int32_t reserved_;
} bpg_curriculum;
int bpg_get_curriculum(const bpg_batch *b, bpg_curriculum *out);
/* overwrite the whole state (restoring a checkpoint; a test that starts from a given difficulty) */
int bpg_set_curriculum(bpg_batch *b, const bpg_curriculum *in);
int bpg_get_difficulty(const bpg_batch *b);
/* increase_difficulty() of the variant (fetch_env.py:351-358 BlocksTouch*, :419-432 BlocksTouchChoose*, :623-630 Variation):
 * *at_max = what the reference returns (1 = the maximum level was reached, nothing else changed).  BPG_ERR_UNSUPPORTED where
 * the reference raises NotImplementedError (:93-94), BPG_ERR_ATTRIBUTE where it raises AttributeError (quirk Q6). */
int bpg_increase_difficulty(bpg_batch *b, int *at_max);
/* The ranges of a placement as reset() (test = 0) or set_test() (test = 1, fetch_env.py:365-368, :443-446) uses them, from the
 * handle's curriculum state, into `out` (grip_x / grip_y, the variates and the env mask are the caller's to fill).
 * BPG_ERR_UNSUPPORTED with test = 1 where set_test raises NotImplementedError (:100-101). */
int bpg_placement_from_curriculum(const bpg_batch *b, int test, bpg_placement_args *out);

/* ---- the one exchange of a multi-GPU job (train.py:21-26 mpi_average, rollout.py:163-167) -----------------
 * RCCL collectives issued by the library itself on the caller's stream, so that a host that is not PyTorch can run a sharded
 * job and so that the collective sits on the SAME stream as the rollout launch that produced its input (no cross-stream
 * event).  librccl.so is loaded on first use: single-GPU hosts never need it (BPG_ERR_UNSUPPORTED if it cannot be loaded).
 * One process per GPU: rank 0 calls bpg_comm_unique_id, hands the 128 bytes to every rank by its own means (MPI, a socket,
 * torch.distributed's store), every rank calls bpg_comm_create with the same bytes.  bpg_comm_adopt wraps a communicator
 * the host already has (an ncclComm_t as void*); bpg_comm_destroy leaves an adopted one alone. */
typedef struct bpg_comm bpg_comm;
#define BPG_COMM_ID_BYTES 128
int bpg_comm_unique_id(void *id_bytes);
int bpg_comm_create(bpg_comm **out, int rank, int world_size, const void *id_bytes, int device_id);
int bpg_comm_adopt(bpg_comm **out, void *nccl_comm, int rank, int world_size, int device_id);
int bpg_comm_destroy(bpg_comm *c);
int bpg_comm_rank(const bpg_comm *c);
int bpg_comm_world_size(const bpg_comm *c);
/* d_out (world_size * n_local elements, global env order for equal contiguous shards) <- every rank's d_local (n_local
 * elements): the episode fail counts (uint8) or returns (float32) of a rollout, one ncclAllGather on `stream`. */
int bpg_allgather_u8(bpg_comm *c, const uint8_t *d_local, uint8_t *d_out, int64_t n_local, void *stream);
int bpg_allgather_f32(bpg_comm *c, const float *d_local, float *d_out, int64_t n_local, void *stream);
/* in-place sum over the ranks of `count` uint64 words (the [successes, envs] rows of rollout.py:163-167): one ncclAllReduce */
int bpg_allreduce_sum_u64(bpg_comm *c, unsigned long long *d_inout, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif
