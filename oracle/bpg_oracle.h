/*
 * bpg_oracle.h -- CPU ORACLE for the gym_blocks step()/reset() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, scalar restatement of the
 * reference's algorithm (matthew9671/BlockPuzzle-gym, paths relative to the
 * reference root) in the SAME FORM the reference computes it: a float64
 * N x N ternary "touch matrix" per env, Python-style loops, no bit packing.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (blockpuzzle-gym_amd/) never links, imports or calls
 * anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against fixtures recorded from the reference's own code
 * (tests/golden/make_golden.py; generated in the build container by importing
 * /root/reference through inert stand-ins for gym / mujoco_py).  Not pinned,
 * because the code lives in third-party packages absent from the reference
 * tree (mujoco_py>=1.50 / MuJoCo 1.50, gym>=0.10 -- no version is pinned by
 * setup.py:6): physics, observation floats, gym's seeding hash, TimeLimit
 * (restated from SURVEY.md A.4, marked below).
 */
#ifndef BPG_ORACLE_H
#define BPG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_OBJS 6
#define ORC_MAX_G    36

/* colours, gym_blocks/envs/fetch_env.py:11-15 */
enum { ORC_GREY = 0, ORC_RED = 1, ORC_GREEN = 2, ORC_BLUE = 3, ORC_NUM_COLORS = 4 };

/* one per class registered in gym_blocks/__init__.py:6-53 */
enum {
    ORC_GRIPPER_TOUCH = 0,            /* GripperTouch-v0                 */
    ORC_BLOCKS_TOUCH = 1,             /* BlocksTouch-v0                  */
    ORC_TOPPLE_TOWER = 2,             /* ToppleTower-v0                  */
    ORC_BLOCKS_TOUCH_CURRICULUM = 3,  /* BlocksTouchCurriculum-v0        */
    ORC_BLOCKS_TOUCH_CHOOSE = 4,      /* BlocksTouchChoose-v0            */
    ORC_BLOCKS_TOUCH_CHOOSE_CURRICULUM = 5,
    ORC_BLOCKS_TOUCH_VARIATION = 6,   /* BlocksTouchVariation-v0         */
    ORC_NUM_VARIANTS = 7
};

typedef struct {
    int variant;
    int num_objs;                 /* fetch_env.py:75; Variation per episode :649 */
    int side;                     /* matrix side: num_objs, Variation always 6 (:663) */
    int is_variation;
    double achieved_goal[ORC_MAX_G];   /* fetch_env.py:78, row-major side x side */
    long long goal[ORC_MAX_G];         /* fetch_env.py:260-273 / :682-695        */
    int obj_colors[ORC_MAX_OBJS];
    int has_succeeded;            /* fetch_env.py:80  */
    int elapsed_steps;            /* gym TimeLimit (third-party) */
    int max_episode_steps;        /* gym_blocks/__init__.py: 50 everywhere   */
} orc_env;

/* fetch_env.py:106-117: geom name -> object id, -1 for None. */
int orc_geom2objid(const char *name);

/* fetch_env.py:323-326,360-363,434-441,632-639,772-775: constant colourings.
 * Returns the number of objects written. */
int orc_sample_colors(int variant, int *colors);

/* fetch_env.py:260-273 (Variation :682-695): colour table -> goal matrix. */
void orc_sample_goal(const int *colors, int side, long long *goal);

/* fetch_env.py:39-86 + tasks.py presets: fresh env, matrix all -1. */
void orc_env_init(orc_env *e, int variant, int max_episode_steps);

/* robot_env.py:71-82 + fetch_env.py:247-255 (Variation :646-679) + TimeLimit.
 * num_grey is the explicit variate for np_random.randint(3) (:647); ignored
 * for the other variants. Base classes do NOT clear achieved_goal (quirk Q1). */
void orc_env_reset(orc_env *e, int num_grey);

/* fetch_env.py:154-167 over a contact list of geom ids (contacts[2*i],
 * contacts[2*i+1]); id2obj[g] == -1 means None. */
void orc_step_callback(orc_env *e, const int *id2obj, const int *contacts, int ncon);

/* Same update with the step's contact set given as a bit mask over the
 * 21 unordered pairs (i<=j) of 6 objects, pair index orc_pair_index(i,j). */
void orc_step_callback_mask(orc_env *e, uint32_t mask);

/* fetch_env.py:119-124 */
int orc_check_goal(const orc_env *e);

/* fetch_env.py:135-143 for one row: -(sum(ag*g) != count_nonzero(g)) as float32. */
float orc_compute_reward(const double *ag, const double *g, int len);
/* vectorised over n rows of `len` (config.py:110-111 call shape). */
void orc_compute_reward_batch(const double *ag, const double *g, int n, int len, float *out);
/* float32 arrays: float32 products and NumPy's float32 pairwise sum; dsum (optional) receives the sums */
void orc_compute_reward_f32_batch(const float *ag, const float *g, int n, int len, float *out, float *dsum);
/* NumPy-order row sums (np.sum(a, axis=-1)) */
void orc_rowsum_f64(const double *a, int n, int len, double *out);
void orc_rowsum_f32(const float *a, int n, int len, float *out);

/* fetch_env.py:275-281: recompute from own state, latch. */
int orc_is_success(orc_env *e);

/* robot_env.py:57-69 (+ TimeLimit done).  After the physics (not modelled) has
 * produced the step's contact list.  Writes reward/done/is_success. */
void orc_env_step(orc_env *e, const int *id2obj, const int *contacts, int ncon,
                  float *reward, int *done, int *is_success);
void orc_env_step_mask(orc_env *e, uint32_t mask, float *reward, int *done, int *is_success);

/* achieved_goal as _get_obs returns it (fetch_env.py:220 / :610-613). */
void orc_get_achieved_goal(const orc_env *e, double *out);

/* robot_env.py:58 clip + fetch_env.py:170-185 _set_action: float32 action (4) -> the float64
 * 9-vector [pos_ctrl*0.05 (float32 product), 1,0,1,0, grip, grip] handed to the robotics
 * utils (gripper entries zeroed when block_gripper). */
void orc_set_action(const float *action, int block_gripper, double *out9);

/* What _get_obs reads from the simulator (all float64, as mujoco_py returns them):
 * site pos / linear velocity of robot0:grip, the last two robot joint positions and
 * velocities, and per block the site pos, euler angles (rotations.mat2euler of the site
 * matrix -- third-party, passed in), linear and angular site velocities. */
typedef struct {
    const double *grip_pos, *grip_velp, *gripper_state, *gripper_vel;   /* 3,3,2,2 */
    const double *obj_pos, *obj_rot, *obj_velp, *obj_velr;              /* [nblocks][3] */
} orc_phys;
/* fetch_env.py:187-228 (Variation :567-621): obs['observation'].  Returns its length. */
int orc_get_obs(const orc_env *e, const orc_phys *p, double dt, double *obs);

/* pair index of unordered pair (i,j), i,j < 6: row-major upper triangle
 * including the diagonal (build-defined packing, 21 pairs). */
int orc_pair_index(int i, int j);
/* contact list -> pair mask (build-defined helper). */
uint32_t orc_contacts_to_mask(const int *id2obj, const int *contacts, int ncon);

/* policy_gradient/rollout.py:255-258: returns[t_] += gamma**(t-t_) * r_new,
 * evaluated in the reference's order, float64.  r: (T) rewards of one env. */
void orc_discounted_return(const double *r, int T, double gamma, double *G);

/* fetch_env.py:30-32 */
int orc_out_of_table(double x, double y);

/* _randomize_objects of the five env classes (fetch_env.py:328-336, 370-399, 448-517,
 * 697-764, 777-787) with the random draws passed in as STANDARD variates in the order the
 * reference consumes them: eu = random_sample() values of the per-env
 * np_random stream, gz / gu = standard normals / uniforms of the global np.random stream.
 * uniform(lo,hi) is restated as lo + (hi-lo)*u and normal() as z (NumPy legacy RandomState).
 * np.linalg.norm of a 2-vector is sqrt(fma(y,y,x*x)) -- what NumPy 2.2.6 + its BLAS ddot
 * compute in the container the fixtures were recorded in (BLAS dependent in general).
 * xy: (num_objs-2, 2) block positions written for the blocks the reference moves.
 * Returns 0, or 1 when a variate array ran out before the rejection loops finished. */
typedef struct {
    double grip_x, grip_y;                         /* initial_gripper_xpos[:2], fetch_env.py:300-301 */
    double obj_range, max_obj_range, wrong_obj_range;
    int test, challenge;
    const double *eu; int n_eu;
    const double *gz; int n_gz;
    const double *gu; int n_gu;
} orc_placement;
int orc_randomize_objects(const orc_env *e, const orc_placement *p, double *xy, int *used /*[3]*/);

/* ---- build-defined synthetic contact source (NO reference counterpart) ----
 * Stand-in for the physics in throughput runs.  Counter-based:
 * a pure function of (seed, env_id, t).  Table-block pairs (1,k) forced on,
 * every other pair of the variant's objects Bernoulli(5/32).               */
uint32_t orc_synth_mask(uint64_t seed, uint64_t env_id, uint32_t t, int num_objs);
uint32_t orc_hash32(uint32_t x);
/* per-env generator key: the reference's seed + 1000 * idx (rollout.py:206-210) on the global env index */
uint32_t orc_env_key(uint64_t seed, uint64_t env_id, uint32_t salt);
int orc_num_grey_draw(uint64_t seed, uint64_t env_id, uint32_t counter);

/* rollout.py:75-172 for B envs and T steps with mask contacts (time-major
 * masks[t*B+b]).  Outputs float32 episode arrays in batch-major layout:
 * ag (B,T+1,dimg), g (B,T,dimg), info_is_success (B,T,1); optional per-step
 * reward (T,B) float32.  Returns mean(successes[-1]). envs are reset first
 * (num_grey[b] for Variation, may be NULL -> 0). */
double orc_generate_rollouts(orc_env *envs, int B, int T, const uint32_t *masks,
                             const int *num_grey,
                             float *ag, float *g, float *info_is_success, float *reward_tb);

/* cpu_baseline leg: B envs x T steps of the matrix-form step on synthetic
 * masks, no episode buffers.  Returns a checksum so the work is not elided. */
uint64_t orc_bench_rollout(int variant, int B, int T, uint64_t seed);
/* Same workload as the GPU headline: the (T, B) masks are generated beforehand
 * (orc_fill_masks, untimed) and READ from memory by the timed loop, which also
 * writes the per-step reward (float32) and flags byte like the kernel does. */
void orc_fill_masks(uint64_t seed, uint64_t env_base, int B, int T, int num_objs, uint32_t *masks);
uint64_t orc_bench_rollout_masks(int variant, int B, int T, int num_grey, const uint32_t *masks,
                                 float *reward_tb, uint8_t *flags_tb);

#ifdef __cplusplus
}
#endif
#endif
