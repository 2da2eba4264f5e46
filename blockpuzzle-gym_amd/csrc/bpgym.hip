// bpgym.hip -- hand-written HIP (gfx950 / CDNA4, wave64) implementation of the
// batched gym_blocks step()/reset() hot path behind include/bpgym.h.
//
// Reference semantics implemented here (paths relative to the reference root):
//   _step_callback   gym_blocks/envs/fetch_env.py:148-167
//   compute_reward   fetch_env.py:135-143   (-1.0f / -0.0f)
//   _is_success      fetch_env.py:275-281   (latched)
//   _sample_goal     fetch_env.py:260-273, Variation :682-695
//   _reset_sim       fetch_env.py:247-255,  Variation :646-679
//   RobotEnv.step    gym_blocks/envs/robot_env.py:57-69 (+ gym TimeLimit done)
//   generate_rollouts gym_blocks/rollout.py:75-172 (episode arrays, batch-major)
//
// Design: the symmetric ternary N x N touch matrix is stored as two
// bit planes over the 21 unordered pairs of <=6 objects, 8 bytes per env in two
// SoA uint32 arrays.  Everything is integer/bitmask streaming work: kernels are
// HBM-bound, use 16-byte-per-lane coalesced accesses, keep state in registers
// across fused steps, stage per-wave history tiles in LDS so that the float32
// episode buffers are written as contiguous 1 KiB wave stores, and reduce
// success counts with ballot/popcount.  No MFMA (nothing here is a contraction).
//
// This file never touches oracle/ and has no CPU fallback: without a GPU every
// launch fails with BPG_ERR_HIP.

#include <hip/hip_runtime.h>

// Bit-exact float64/float32 parity with the reference's NumPy arithmetic needs
// every product and sum rounded separately: HIP's __dmul_rn/__dadd_rn are plain
// operators, and hipcc contracts a*b+c into an fma by default.  Nothing in this
// file is FLOP-bound, so contraction is simply off (also -ffp-contract=off in build.py).
#pragma STDC FP_CONTRACT OFF

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bpgym.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(BPG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

constexpr uint32_t PAIR_BITS = (1u << BPG_NUM_PAIRS) - 1u;  // bits 0..20
constexpr uint32_t ELAPSED_SHIFT = 21;
constexpr uint32_t ELAPSED_MAX = 1023;                      // 10 bits, saturating
constexpr uint32_t NOBJ_SHIFT = 21;                         // in `ever`
constexpr uint32_t SUCC_BIT = 1u << 31;                     // in `ever`
constexpr uint32_t RBITS_FAIL = 0xBF800000u;                // -1.0f
constexpr uint32_t RBITS_OK = 0x80000000u;                  // -0.0f (quirk Q2)

constexpr int host_pair_index(int i, int j)
{
    return i <= j ? i * BPG_MAX_OBJS - i * (i - 1) / 2 + (j - i)
                  : j * BPG_MAX_OBJS - j * (j - 1) / 2 + (i - j);
}

// ---------------------------------------------------------------------------
// variant table (host): tasks.py presets + constant colourings + goal planes
// ---------------------------------------------------------------------------
enum { GREY = 0, RED = 1, GREEN = 2, BLUE = 3 };   // fetch_env.py:11-14

struct VariantDef {
    const char *id;
    int num_objs;
    int colors[BPG_MAX_OBJS];
    int block_gripper;       // tasks.py
    int set_test;            // fetch_env.py:100-101 raises unless overridden
    int incr_status;         // quirk Q6
};

const VariantDef kVariants[BPG_NUM_VARIANTS] = {
    // fetch_env.py:323-326, tasks.py:4-20
    {"GripperTouch-v0", 3, {BLUE, GREY, GREEN, 0, 0, 0}, 0, 0, BPG_ERR_UNSUPPORTED},
    // fetch_env.py:360-363, tasks.py:22-40
    {"BlocksTouch-v0", 4, {GREY, GREY, GREEN, BLUE, 0, 0}, 0, 1, BPG_OK},
    // fetch_env.py:772-775, tasks.py:126-144
    {"ToppleTower-v0", 6, {RED, GREEN, GREY, GREY, GREY, BLUE}, 0, 0, BPG_ERR_UNSUPPORTED},
    // tasks.py:42-60 (curriculum=True, block_gripper=True)
    {"BlocksTouchCurriculum-v0", 4, {GREY, GREY, GREEN, BLUE, 0, 0}, 1, 1, BPG_OK},
    // fetch_env.py:434-441, tasks.py:62-81; curriculum=False lacks obj_range_step (:413-415)
    {"BlocksTouchChoose-v0", 5, {GREY, GREY, GREEN, BLUE, GREY, 0}, 1, 1, BPG_ERR_ATTRIBUTE},
    {"BlocksTouchChooseCurriculum-v0", 5, {GREY, GREY, GREEN, BLUE, GREY, 0}, 1, 1, BPG_OK},
    // fetch_env.py:632-639 (max_num_blocks = 4, :523), tasks.py:104-124
    {"BlocksTouchVariation-v0", 6, {GREY, GREY, GREEN, BLUE, GREY, GREY}, 1, 1, BPG_OK},
};

void fill_variant_info(int v, bpg_variant_info *o)
{
    const VariantDef &d = kVariants[v];
    std::memset(o, 0, sizeof(*o));
    o->variant = v;
    o->num_objs = d.num_objs;
    o->side = d.num_objs;
    o->dimg = o->side * o->side;
    o->is_variation = (v == BPG_BLOCKS_TOUCH_VARIATION);
    // fetch_env.py:187-228: 10 env features + 15 per block; Variation :567-621:
    // 1 + 10 + 4 * (15 + 4)
    o->dimo = o->is_variation ? 87 : 10 + 15 * (d.num_objs - 2);
    o->dimu = 4;
    o->block_gripper = d.block_gripper;
    o->max_episode_steps = 50;
    o->set_test_supported = d.set_test;
    o->increase_difficulty_status = d.incr_status;
    o->desired_goal_is_float = o->is_variation;
    std::snprintf(o->env_id, sizeof(o->env_id), "%s", d.id);
    for (int i = 0; i < BPG_MAX_OBJS; i++)
        o->obj_colors[i] = d.colors[i];
    // _sample_goal, fetch_env.py:260-273
    for (int i = 0; i < o->side; i++) {
        for (int j = 0; j < o->side; j++) {
            int ci = d.colors[i], cj = d.colors[j], g = 0;
            if ((ci == RED && cj == BLUE) || (ci == BLUE && cj == RED))
                g = -1;
            else if ((ci == GREEN && cj == BLUE) || (ci == BLUE && cj == GREEN))
                g = 1;
            o->goal[i * o->side + j] = g;
            uint32_t bit = 1u << host_pair_index(i, j);
            if (g == 1) o->want_touch |= bit;
            if (g == -1) o->want_never |= bit;
            o->valid_pairs |= bit;
        }
    }
}

}  // namespace

struct bpg_batch {
    bpg_variant_info info;
    int64_t n;
    int device;
    int max_steps;
    uint64_t seed;
    int64_t env_base;
    uint32_t *d_reset_counter;   // two device words: [0] generator-drawn resets so far, [1] generator-drawn
                                 // placements so far (stream indices); on the device so that hipGraph
                                 // replays advance them too
    uint32_t *now;
    uint32_t *ever;
    int8_t *d_id2obj;   // [BPG_MAX_MODELS][BPG_MAX_GEOMS]
    int ngeom[BPG_MAX_MODELS];
    unsigned long long *d_count_shards;   // COUNT_SHARDS words on 128-byte lines of their own (block_count_add)
    uint64_t *d_okbits;                   // n words: the step-ok bits of the last rollout that asked for G; allocated by
                                          // the first bpg_rollout call with d_G (NULL until then)
};

namespace bpgk {   // kernels + launch helpers (named so profiles read bpgk::k_*)

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------
// device code
// ---------------------------------------------------------------------------
struct StepConst {
    uint32_t valid;        // pairs of the variant's matrix
    uint32_t want_touch;   // goal == +1 pairs
    uint32_t want_never;   // goal == -1 pairs
    uint32_t max_steps;    // TimeLimit
    // fetch_env.py:154-157 demotes 1 -> 0 only for i, j < num_objs.  Variation's matrix is always 6 x 6
    // while num_objs is 4..6 (:649), so an entry that involves an object >= num_objs -- no id2obj table
    // can produce one, but a raw mask can carry it -- is set by :166-167 and never demoted again: it
    // stays 1.  sticky_by_n[n] = pairs of the matrix with an index >= n (0 for n >= side); sticky_on = 0
    // for the variants whose num_objs is the matrix side (nothing to keep, the lookup is skipped).
    uint32_t sticky_on;
    uint32_t sticky_by_n[BPG_MAX_OBJS];
};

// per-env sticky pairs (see StepConst): compare-select over the table rows, which sit in SGPRs
__device__ __forceinline__ uint32_t sticky_of(const StepConst &c, uint32_t ever)
{
    uint32_t s = 0u;
    if (c.sticky_on) {                                   // uniform
        const uint32_t n = (ever >> NOBJ_SHIFT) & 7u;
#pragma unroll
        for (uint32_t k = 0; k < BPG_MAX_OBJS; k++) s = n == k ? c.sticky_by_n[k] : s;
    }
    return s;
}

struct SynthConst {
    uint64_t seed0;        // seed of env 0 of this handle: seed + 1000 * env_id_base (mod 2^64), see env_key()
    uint32_t valid_by_n[8];   // off-diagonal pairs among the first n objects
    uint32_t forced_by_n[8];  // table-block pairs (1,k), k = 2..n-1
};

// One env step on the packed state.  fetch_env.py:154-157 demotes every 1 to 0
// before the contacts are read (quirk Q4), so `now` is simply this step's
// contact set and `ever` accumulates it; reward / success follow
// fetch_env.py:141-143 and :275-281 on the two goal planes.
__device__ __forceinline__ void step_core(uint32_t &now, uint32_t &ever, uint32_t mask,
                                          const StepConst &c, uint32_t &rbits, uint32_t &flags)
{
    const uint32_t cm = (mask & c.valid) | (now & sticky_of(c, ever));
    uint32_t elapsed = (now >> ELAPSED_SHIFT) & ELAPSED_MAX;
    elapsed = elapsed < ELAPSED_MAX ? elapsed + 1 : ELAPSED_MAX;
    const uint32_t ev = ever | cm;
    const uint32_t bad = (c.want_touch & ~cm) | (c.want_never & ev);
    const uint32_t ok = (bad == 0) ? 1u : 0u;
    const uint32_t succ = (ev >> 31) | ok;
    now = cm | (elapsed << ELAPSED_SHIFT);
    ever = (ev & ~SUCC_BIT) | (succ << 31);
    rbits = ok ? RBITS_OK : RBITS_FAIL;
    flags = succ | ((elapsed >= c.max_steps) ? (uint32_t)BPG_FLAG_DONE : 0u) | (ok ? (uint32_t)BPG_FLAG_STEP_OK : 0u);
}

// build-defined counter-based contact source (DESIGN.md "synthetic contacts"), a pure function
// of (seed, global env id, step).  Per env two 32-bit keys from the lowbias32 finaliser (once
// per launch); per step two well-mixed words a, b: a Weyl step of the key, an xorshift, ONE
// 32x32->64 multiply each, high half folded onto the low half (v_mad_u64_u32 + v_xor).  On gfx950
// an integer multiply issues like any other VOP3 instruction (tools/valu_rates.hip: 3.3 vs 3.2
// cycles), so the cost of a generator is its instruction count: round 1's four 2-multiply hashes
// per step (about 37 instructions) made the in-kernel-contacts rollout VALU-bound at 0.38 of peak.
// The 21 Bernoulli(5/32) bits are U < 5 on a bit-sliced 5-bit uniform per pair,
//     A & B & (C | D),   A, B, C fair bits, D a 1/4 bit,
// with A = a[0..20], B = b[0..20], C = a[21..31] ++ b[21..30] -- 63 fresh bits per step -- and
// the low-order digit D_j = a[(j-7) mod 32] & b[(j-13) mod 32] taken from bits that serve as
// A / B / C of OTHER pairs: every pair is exactly Bernoulli(5/32), steps and envs are independent,
// and two pairs of the same step that share such a bit correlate with r <= 0.04 (measured 0.038,
// tools/synth_quality.py) -- the price of 64 instead of 105 fresh bits per step.
__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// Per-env key of every generator-drawn quantity.  The reference seeds env idx of a worker with
// seed + 1000 * idx (RolloutStudent.seed, rollout.py:206-210) -- that arithmetic is reference-owned and is
// mirrored here on the GLOBAL env index: env i of a batch seeded s draws exactly what env 0 of a batch seeded
// s + 1000 i draws (what gym then does with the number -- a SHA-512 hash into MT19937 -- is third-party and
// replaced by the counter-based generator below).  salt separates the streams (contacts, num_grey, placement).
__device__ __forceinline__ uint32_t env_key(uint64_t seed0, uint32_t local_env, uint32_t salt)
{
    const uint64_t es = seed0 + 1000ull * local_env;
    return hash32(((uint32_t)es ^ salt) ^ hash32((uint32_t)(es >> 32) + 0x9E3779B9U));
}

__device__ __forceinline__ uint32_t mulfold(uint32_t x, uint32_t k)
{
    const uint64_t p = (uint64_t)x * k;
    return (uint32_t)(p >> 32) ^ (uint32_t)p;
}

// per-env part of the source, computed once per launch and kept in registers
struct SynthLane {
    uint32_t ka, kb, valid, forced;
};

__device__ __forceinline__ SynthLane synth_lane(const SynthConst &s, uint32_t local_env, uint32_t nobj)
{
    SynthLane l;
    l.ka = env_key(s.seed0, local_env, 0u);
    l.kb = hash32(l.ka ^ 0x85EBCA6BU);
    // compare-select over the eight table rows (they sit in SGPRs): indexing the kernel-argument
    // array with a per-lane value would be a memory load that depends on the state load
    uint32_t valid = s.valid_by_n[0], forced = s.forced_by_n[0];
#pragma unroll
    for (uint32_t k = 1; k < 8; k++) {
        valid = nobj == k ? s.valid_by_n[k] : valid;
        forced = nobj == k ? s.forced_by_n[k] : forced;
    }
    l.valid = valid;
    l.forced = forced;
    return l;
}

__device__ __forceinline__ uint32_t synth_mask(const SynthLane &l, uint32_t t)
{
    const uint32_t u = l.ka + (t + 1u) * 0x9E3779B9U;
    uint32_t x = u, y = u ^ l.kb;
    x ^= x >> 15;
    y ^= y >> 15;
    const uint32_t a = mulfold(x, 0xD6E8FEB9U), b = mulfold(y, 0xCA5A8263U);
    const uint32_t C = (a >> 21) | ((b >> 21) << 11);
    const uint32_t D = __builtin_amdgcn_alignbit(a, a, 25) & __builtin_amdgcn_alignbit(b, b, 19);   // rotl 7, rotl 13
    return (a & b & (C | D) & l.valid) | l.forced;
}

__device__ __forceinline__ uint32_t nobj_of(uint32_t ever) { return (ever >> NOBJ_SHIFT) & 7u; }

// Write-once reward / flag streams of the fused rollout use nontemporal stores:
// they are never re-read by these kernels, and keeping them out of the L2 /
// Infinity Cache leaves it to the inputs (round 1's tools/tune_rollout.hip, profiles/r01_tune_rollout_variants.log: 86 vs 88 us
// with cold inputs, 75 vs 88 us when the masks are cache-resident).  The
// write-only episode streams stay plain stores: a pure float4 fill runs 5.1 TB/s
// plain vs 4.8 TB/s nontemporal on MI355X, and the episode writer follows it.
template <bool NT> __device__ __forceinline__ void st_u4(uint4 *p, uint4 v)
{
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}
template <bool NT> __device__ __forceinline__ void st_f4(float4 *p, float4 v)
{
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}
template <bool NT, typename V> __device__ __forceinline__ void st_1(V *p, V v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT> __device__ __forceinline__ void st_d2(double2 *p, double2 v)
{
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}
// Write-once outputs of the ONE-SHOT kernels (a step, an observation assembly, a float view, the G rows: every wave
// loads, computes, stores once and ends): nontemporal.  A plain store allocates in the L2 and is written back when the
// kernel ends, behind its last wave and before the next launch of the stream may start (MI355X_MICROARCH.md
// "boundary": + B / 6 TB/s for B dirty bytes); a nontemporal one streams out while the kernel runs.  Measured on the step
// launch (tools/tune_round4.hip, profiles/r04_tune_step_train.log) and per kernel with -DBPG_OUT_NT=0 against the
// default on the same box (tools/ab_output_stores.py, profiles/r04_ab_output_stores.log).  The multi-GB episode writer
// and the reset's observation rows are the exceptions (plain stores: profiles/r04_tune_episode_nontemporal.log; reset
// + rows 8.4 us plain against 9.6 us at 65 536 envs), and so is everything a later step of the same launch train re-reads
// (the packed state, contact masks).
#ifndef BPG_OUT_NT
#define BPG_OUT_NT 1
#endif
constexpr bool OUT_NT = (BPG_OUT_NT != 0);

// envs-per-lane vector types of the streaming kernels
template <int EPL> struct LaneVec;
template <> struct LaneVec<4> { using vec = uint4; using fl = uint32_t; };
template <> struct LaneVec<2> { using vec = uint2; using fl = uint16_t; };
template <> struct LaneVec<1> { using vec = uint32_t; using fl = uint8_t; };

template <int EPL> __device__ __forceinline__ void unpack(const typename LaneVec<EPL>::vec &v, uint32_t *o)
{
    if constexpr (EPL == 4) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else if constexpr (EPL == 2) { o[0] = v.x; o[1] = v.y; }
    else o[0] = v;
}
template <int EPL> __device__ __forceinline__ typename LaneVec<EPL>::vec pack(const uint32_t *o)
{
    if constexpr (EPL == 4) return make_uint4(o[0], o[1], o[2], o[3]);
    else if constexpr (EPL == 2) return make_uint2(o[0], o[1]);
    else return o[0];
}
// mask stream: read exactly once per launch -> nontemporal loads (profiles/r02_tune_geometry_nontemporal_loads.log)
template <bool NTL, int EPL> __device__ __forceinline__ typename LaneVec<EPL>::vec ld_vec(const typename LaneVec<EPL>::vec *p)
{
    if constexpr (!NTL) return *p;
    else if constexpr (EPL == 4)
        return make_uint4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y),
                          __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w));
    else if constexpr (EPL == 2) return make_uint2(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y));
    else return __builtin_nontemporal_load(p);
}

template <bool NT, int EPL> __device__ __forceinline__ void st_vec(typename LaneVec<EPL>::vec *p, const uint32_t *o)
{
    if constexpr (EPL == 4) st_u4<NT>(p, make_uint4(o[0], o[1], o[2], o[3]));
    else if constexpr (EPL == 2) {
        if (NT) { __builtin_nontemporal_store(o[0], &p->x); __builtin_nontemporal_store(o[1], &p->y); }
        else *p = make_uint2(o[0], o[1]);
    } else st_1<NT>(p, o[0]);
}


// ------------------------------ reset ---------------------------------------
// fetch_env.py:247-255: has_succeeded = False, matrix kept (Q1);
// Variation :646-679: matrix = -1, num_objs = 4 + num_grey.  TimeLimit: t = 0.
struct ResetConst {
    int32_t enabled;        // apply the reset before step 0 (fused rollouts)
    int32_t is_variation;
    uint32_t base_nobj;
    uint64_t seed0;                  // as SynthConst
    const uint32_t *reset_counter;   // device word, read (uniformly) by every lane
    const int32_t *num_grey;   // explicit variates for np_random.randint(3) (:647) or NULL
};

__device__ __forceinline__ void reset_core(uint32_t &now, uint32_t &ever, int64_t i, const ResetConst &rc,
                                           uint32_t stream_index_offset = 0u, int64_t variate_offset = 0)
{
    if (rc.is_variation) {
        uint32_t g;
        if (rc.num_grey) {
            const int32_t v = rc.num_grey[variate_offset + i];
            g = v < 0 ? 0u : (v > 2 ? 2u : (uint32_t)v);
        } else {
            // counter-based stand-in for np_random.randint(3)
            g = hash32(env_key(rc.seed0, (uint32_t)i, 0xA511E9B3U) + *rc.reset_counter + stream_index_offset) % 3u;
        }
        now = 0u;
        ever = (4u + g) << NOBJ_SHIFT;
    } else {
        now = now & PAIR_BITS;
        ever = (ever & PAIR_BITS) | (rc.base_nobj << NOBJ_SHIFT);
    }
}

// The observation a reset leaves behind, written by the reset launch itself into a float32 buffer
// (row stride ld).  There is no physics state behind the contacts-in boundary, so the row is what
// the reference's _get_obs fills WITHOUT physics -- Variation (fetch_env.py:579-607): obs[0] =
// num_blocks, the colour one-hot of every present block, zeros in the slots of absent blocks; every
// variant: nothing else -- and zeros in the physics-owned entries (a backend writes those in its
// on_reset).  full = 0 (bpg_fill_static_obs): only the physics-free entries are written, the rest of
// the row is left as it is.  obs == NULL: state only.
struct StaticObs {
    float *obs;
    int64_t ld;
    int32_t enabled, nb_max, dimo, is_variation, full, flat;   // flat: ld == dimo, no env mask, 16-B aligned
    int32_t colors[BPG_MAX_OBJS];
};

__device__ __forceinline__ float obs_entry(const StaticObs &so, int k, int nb, bool &owned)
{
    owned = true;                                   // entry is defined without physics
    if (!so.is_variation) { owned = false; return 0.0f; }
    if (k == 0) return (float)nb;                                           // :580
    const int j = k - 11;                                                   // 1 + ENV_FEATURES
    if (j < 0) { owned = false; return 0.0f; }                              // gripper entries: physics
    const int slot = j / 19, c = j - slot * 19;
    if (slot >= nb) return 0.0f;                                            // zero padding :606-607
    if (c >= 15) return so.colors[slot + 2] == c - 15 ? 1.0f : 0.0f;        // one_hot_color :599
    owned = false;                                                          // block pose / velocity: physics
    return 0.0f;
}

// RESET = false: only the observation entries, from the state as it is (constructor / restore).
template <bool RESET>
__global__ void __launch_bounds__(256)
k_reset(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, int64_t n, ResetConst rc,
        const uint8_t *__restrict__ env_mask, StaticObs so)
{
    __shared__ uint8_t s_nblocks[256];              // 0xFF = env not touched by this launch
    // whole-row form: the row is a function of the episode's block count alone -> one template row
    // per count (<= 5 x 87 floats), built once per block; the streaming loop below is then two LDS
    // reads per element instead of obs_entry()'s divisions
    __shared__ float s_tmpl[(BPG_MAX_OBJS - 1) * 88];
    if (so.enabled && so.flat) {
        const int nt = (so.nb_max + 1) * so.dimo;
        for (int i = threadIdx.x; i < nt; i += 256) {
            const int nb = i / so.dimo;
            bool owned;
            s_tmpl[i] = obs_entry(so, i - nb * so.dimo, nb, owned);
        }
    }
    const int64_t nblk = (n + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t base = blk * 256, i = base + threadIdx.x;
        uint8_t nbk = 0xFF;
        if (i < n && !(env_mask && env_mask[i] == 0)) {
            uint32_t e = ever[i];
            if (RESET) {
                uint32_t a = now[i];
                reset_core(a, e, i, rc);
                now[i] = a;
                ever[i] = e;
            }
            nbk = (uint8_t)(nobj_of(e) - 2u);
        }
        if (!so.enabled) continue;                   // uniform
        __syncthreads();                             // previous tile consumed
        s_nblocks[threadIdx.x] = nbk;
        __syncthreads();
        const int rows = (int)((n - base) < 256 ? (n - base) : 256);
        if (so.flat) {
            // the block's rows are one contiguous run of rows * dimo floats: float4 stores, every
            // element's (row, entry) from a division by the row length
            const int total = rows * so.dimo, nq = total >> 2;
            float *o = so.obs + base * so.dimo;
            for (int q = threadIdx.x; q < nq; q += 256) {
                float v[4];
                int e = (4 * q) / so.dimo, k = 4 * q - e * so.dimo;            // one division per float4
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    v[c] = s_tmpl[s_nblocks[e] * so.dimo + k];
                    if (++k == so.dimo) { k = 0; e++; }
                }
                reinterpret_cast<float4 *>(o)[q] = make_float4(v[0], v[1], v[2], v[3]);   // plain: nontemporal measured +14 % at 65 536 envs
            }
            for (int f = 4 * nq + threadIdx.x; f < total; f += 256) {
                const int e = f / so.dimo;
                o[f] = s_tmpl[s_nblocks[e] * so.dimo + (f - e * so.dimo)];
            }
            continue;
        }
        // general form (env mask, strided rows, or only the physics-free entries): each wave walks 64
        // of the block's rows, lane = entry of the row
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int k = lane; k < so.dimo; k += 64) {
            for (int e = wave * 64; e < rows && e < wave * 64 + 64; e++) {
                const int nb = s_nblocks[e];
                if (nb == 0xFF) continue;
                bool owned;
                const float v = obs_entry(so, k, nb, owned);
                if (owned || so.full) so.obs[(base + e) * so.ld + k] = v;
            }
        }
    }
}

// Enqueued right after every launch that reset the batch: stream order makes all lanes of the
// reset see the old value and the next reset the new one, also under hipGraph replay.
__global__ void k_bump_counter(uint32_t *counter, uint32_t by) { *counter += by; }

__global__ void __launch_bounds__(256)
k_init(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, int64_t n, uint32_t nobj)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        now[i] = 0u;                      // fetch_env.py:78 matrix all -1
        ever[i] = nobj << NOBJ_SHIFT;     // :80 has_succeeded False
    }
}

// ------------------------------ single step ---------------------------------
enum { SRC_MASKS = 0, SRC_SYNTH = 1 };

// the step's contact mask: read from memory, or the synthetic source at the env's elapsed step
template <int SRC>
__device__ __forceinline__ uint32_t fetch_mask(uint32_t m, uint32_t now, uint32_t ever, uint32_t local_env,
                                               const SynthConst &s)
{
    if (SRC == SRC_MASKS) return m;
    return synth_mask(synth_lane(s, local_env, nobj_of(ever)), (now >> ELAPSED_SHIFT) & ELAPSED_MAX);
}

// 4 envs per lane: 16-byte coalesced loads/stores of now / ever / masks / reward,
// one packed dword of 4 flag bytes.  Scalar tail for n % 4.
template <int SRC>
__global__ void __launch_bounds__(256)
k_step_vec4(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
            uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c, SynthConst s)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = gtid; v < nvec; v += stride) {
        uint4 a = reinterpret_cast<uint4 *>(now)[v];
        uint4 e = reinterpret_cast<uint4 *>(ever)[v];
        uint4 m = make_uint4(0, 0, 0, 0);
        if (SRC == SRC_MASKS) m = ld_vec<true, 4>(reinterpret_cast<const uint4 *>(masks) + v);   // read once
        uint32_t an[4] = {a.x, a.y, a.z, a.w}, ee[4] = {e.x, e.y, e.z, e.w}, mm[4] = {m.x, m.y, m.z, m.w};
        uint32_t rb[4], fl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mk = fetch_mask<SRC>(mm[k], an[k], ee[k], (uint32_t)(4 * v + k), s);
            step_core(an[k], ee[k], mk, c, rb[k], fl[k]);
        }
        // (the state stays a plain store: nontemporal state stores change nothing here -- 5.1-5.4 us per 1 M-env launch
        // with the state cache-resident, 5.9-6.1 us HBM-cold, either way: the launch is its fixed cost, a load and a
        // store round trip between two launch boundaries; profiles/r04_ab_output_stores.log)
        reinterpret_cast<uint4 *>(now)[v] = make_uint4(an[0], an[1], an[2], an[3]);
        reinterpret_cast<uint4 *>(ever)[v] = make_uint4(ee[0], ee[1], ee[2], ee[3]);
        // write-once streams: nontemporal, so that they do not push the state (re-read by the next
        // step) out of the caches
        if (reward) st_u4<true>(reinterpret_cast<uint4 *>(reward) + v, make_uint4(rb[0], rb[1], rb[2], rb[3]));
        if (flags)
            st_1<true>(reinterpret_cast<uint32_t *>(flags) + v, fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24));
    }
    const int64_t tail0 = nvec << 2;
    if (gtid < n - tail0) {
        const int64_t i = tail0 + gtid;
        uint32_t a = now[i], e = ever[i], rb, fl;
        const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
        step_core(a, e, mk, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

// Unaligned-pointer fallback: one env per lane, 4-byte accesses.
template <int SRC>
__global__ void __launch_bounds__(256)
k_step_scalar(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
              uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c, SynthConst s)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t a = now[i], e = ever[i], rb, fl;
        const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
        step_core(a, e, mk, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

// Contacts as the reference reads them (fetch_env.py:158-167): per env a list
// of geom-id pairs.  One env per lane; the id2obj tables live in LDS.
__device__ __forceinline__ uint32_t pair_bit_of(const int8_t *tab, uint32_t pr)
{
    const uint32_t g1 = pr & 0xFFFFu, g2 = pr >> 16;          // geom1 | geom2 << 16
    if (g1 < BPG_MAX_GEOMS && g2 < BPG_MAX_GEOMS) {
        const int o1 = tab[g1], o2 = tab[g2];
        if (o1 >= 0 && o2 >= 0) {                              // both not None (:165)
            const int lo = o1 < o2 ? o1 : o2, hi = o1 < o2 ? o2 : o1;
            return 1u << (lo * BPG_MAX_OBJS - lo * (lo - 1) / 2 + (hi - lo));
        }
    }
    return 0u;
}

// The pair mask of one padded list row.  VW = 4: the row is walked in 16-byte steps -- every lane strides through a row
// of its own, so a wave's load touches 64 cache lines whatever its width, and it is the number of such loads, not the
// bytes, that bounds the walk (dword steps: 0.40 of the HBM peak at maxcon = 16; 16-byte steps:
// profiles/r04_contacts_to_masks.log).  Needs maxcon % 4 == 0 and a 16-byte aligned list.
template <int VW>
__device__ __forceinline__ uint32_t row_pair_mask(const int8_t *tab, const uint32_t *row, int nc)
{
    uint32_t m = 0;
    if (VW == 4) {
        const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
        for (int k = 0; k < nc; k += 4) {
            const uint4 v = row4[k >> 2];
            m |= pair_bit_of(tab, v.x);
            if (k + 1 < nc) m |= pair_bit_of(tab, v.y);
            if (k + 2 < nc) m |= pair_bit_of(tab, v.z);
            if (k + 3 < nc) m |= pair_bit_of(tab, v.w);
        }
    } else {
        for (int k = 0; k < nc; k++) m |= pair_bit_of(tab, row[k]);
    }
    return m;
}

// One env per lane; the id2obj tables live in LDS.  (Round 2 staged the block's 256 rows through LDS with coalesced
// 16-byte loads; the 16-byte row walk of row_pair_mask<4> is as fast or faster at every list length -- 1 M envs,
// maxcon 8 / 16 / 32: 14.6 / 19.6 / 28.4 us against 16.2 / 19.9 / 33.9 staged and 20.1 / 29.1 / 50.0 with dword steps,
// profiles/r04_contacts_to_masks.log -- and needs no tile.)
template <int VW>
__global__ void __launch_bounds__(256)
k_step_contacts(uint32_t *__restrict__ now, uint32_t *__restrict__ ever,
                const int16_t *__restrict__ contacts, const int32_t *__restrict__ ncon, int maxcon,
                const int8_t *__restrict__ id2obj, int is_variation,
                uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t a = now[i], e = ever[i], rb, fl;
        int model = 0;
        if (is_variation) {
            model = (int)nobj_of(e) - 4;                   // fetch_env.py:650-651
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        int nc = ncon[i];
        nc = nc < 0 ? 0 : (nc > maxcon ? maxcon : nc);
        const uint32_t m = row_pair_mask<VW>(s_tab + model * BPG_MAX_GEOMS, reinterpret_cast<const uint32_t *>(contacts) + i * maxcon, nc);
        step_core(a, e, m, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

__global__ void __launch_bounds__(256)
k_synth_masks(const uint32_t *__restrict__ ever, int64_t n, int t0, int T, uint32_t *__restrict__ out,
              int64_t ld, SynthConst s)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const SynthLane l = synth_lane(s, (uint32_t)i, nobj_of(ever[i]));
        for (int t = 0; t < T; t++) {
            uint32_t tt = (uint32_t)(t0 + t);
            tt = tt > ELAPSED_MAX ? ELAPSED_MAX : tt;
            out[(int64_t)t * ld + i] = synth_mask(l, tt);
        }
    }
}

// ------------------------------ fused rollout -------------------------------
// rollout.py:163-167 inside a kernel: per-wave ballot + popcount, LDS sum over the block's waves, then
// ONE returning atomic per block on one of COUNT_SHARDS shard words (each on a 128-byte line of its
// own, owned by the handle, all zero between launches).  Device-scope atomics execute at the memory
// side and same-address ones queue up at 11-13 ns each: 256 blocks that finish together on ONE word
// cost 2-4 us at the end of the launch (the whole T-independent gap between a 65 536-env rollout with
// and without the count, profiles/r02_tune_geometry.log), 1 024 blocks 12 us.  Sharded, the chains are
// COUNT_SHARDS times shorter and run in parallel.  A shard word carries the sum (low 40 bits) and the
// number of arrivals (high 24 bits); the block whose add comes last on its shard -- told by the value
// the add returned, so no block ever waits -- moves the shard's sum to the caller's counter (at most
// COUNT_SHARDS adds on that word) and clears the shard for the next launch.  Blocks b with
// b % COUNT_SHARDS == s share shard s; which CU or XCD runs them does not matter.
constexpr int COUNT_SHARDS = 32;
constexpr int COUNT_SHARD_STRIDE = 16;              // uint64 words between shards (128 bytes)
constexpr unsigned long long COUNT_SUM_MASK = (1ull << 40) - 1ull;

struct CountArgs {
    unsigned long long *count;       // caller's counter (+= successes), or NULL
    unsigned long long *zero_next;   // optional second counter the launch sets to 0
    unsigned long long *shards;      // COUNT_SHARDS * COUNT_SHARD_STRIDE words; NULL = one plain atomic per block
};

// one thread per block
__device__ __forceinline__ void shard_count_add(unsigned long long tot, const CountArgs &ca)
{
    if (!ca.shards) {
        if (tot) atomicAdd(ca.count, tot);
        return;
    }
    const uint32_t s = blockIdx.x % COUNT_SHARDS;
    const uint32_t in_shard = (gridDim.x - s + COUNT_SHARDS - 1) / COUNT_SHARDS;      // blocks on this shard (>= 1)
    unsigned long long *w = ca.shards + (size_t)s * COUNT_SHARD_STRIDE;
    const unsigned long long old = atomicAdd(w, (1ull << 40) | tot);
    if ((uint32_t)(old >> 40) + 1u == in_shard) {                                    // the shard's last arrival
        const unsigned long long sum = (old & COUNT_SUM_MASK) + tot;
        (void)__hip_atomic_exchange(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sum) atomicAdd(ca.count, sum);
    }
}

__device__ __forceinline__ void block_count_add(uint32_t lane_count, const CountArgs &ca)
{
    __shared__ uint32_t s_part[16];
    uint32_t c = lane_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < nw; w++) tot += s_part[w];
        shard_count_add(tot, ca);
    }
}

// T steps in one launch, EPL envs per lane, state in registers.  Masks / rewards are
// time-major (T, n) so every access is an (EPL*4)-byte-per-lane coalesced stream.
// Optionally resets first (rollout.py:79) and counts the final successes.
// EPL, the block size and the loop form are picked from n by rollout_geometry().
// OUT: which per-step streams are written (bit 0 reward, bit 1 flags) -- a compile-time choice,
// because a store behind a run-time `if (pointer)` makes the compiler assume "maybe not issued"
// when it counts outstanding memory operations, and every wait then also drains the stores.
enum { OUT_REWARD = 1, OUT_FLAGS = 2 };

// ---- policy-gradient return from the step-ok words of a rollout (policy_gradient/rollout.py:255-258) ----
// G[t_] starts as r[t_] and receives gamma**(t - t_) * r[t] for t = t_+1 .. T-1 in increasing t, in
// float64, every product and sum rounded on its own (NumPy).  Inside the rollout the step rewards are
// not arbitrary floats: r[t] is -0.0 when the step's success condition holds and -1.0 otherwise
// (fetch_env.py:141-143), i.e. ONE bit per step, and T <= 64 of them are a 64-bit mask the lane already
// holds.  With r in {-1.0, -0.0} the product gamma**k * r is exact (-gamma**k or -0.0), and adding
// -0.0 leaves every float64 value unchanged, so the reference's arithmetic is, bit for bit,
//     if (!ok[t]) G[t_] = G[t_] - pw[t - t_]          for all t_ < t, t ascending,
// one float64 subtraction per (t_, t) pair under the wave's execution mask instead of a product and a sum
// (round 2's k_discounted_return_reg: 2 450 float64 operations per env, here at most 1 225), no reward stream
// written and read back (21 -> 13 bytes per env-step for the pg path), and no per-lane reward operand:
// pw[k] sits in SGPRs.  Lane = env, TT accumulators in registers.  The T x T triangle is walked in two
// passes over the lag k = t - t_ (first k <= TT/2, then the rest): per accumulator the additions still
// come in increasing t, and each pass needs only half of the gamma**k table in SGPRs at a time.
constexpr int DR_TMAX_REG = 64;
struct PowTable { double v[DR_TMAX_REG]; };

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

// The gamma**k table reaches the kernels as their FIRST by-value argument, i.e. at offset 0 of the kernarg
// segment, and is read from there with scalar loads where it is used.  Left to itself the compiler loads
// every kernel argument in the entry block: 100 SGPRs of table alive through the whole launch, spilled to
// VGPR lanes and read back lane by lane.  kernarg_pow() hands out the segment pointer through an opaque
// asm, once per pass, so the loads stay where the pass is.
using KernargDouble = const __attribute__((address_space(4))) double;
__device__ __forceinline__ KernargDouble *kernarg_pow()
{
    KernargDouble *p = (KernargDouble *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// an opaque asm keeps a loaded constant in its SGPR pair: scalar loads from the kernarg segment can be
// rematerialised, and under SGPR pressure the register allocator re-loaded the constants inside every
// conditional block below (129 s_load_dwordx16 + a full lgkmcnt wait per step) instead of keeping them
__device__ __forceinline__ void pin_sgpr(double &x) { asm volatile("" : "+s"(x)); }

template <int TT>
__device__ __forceinline__ void returns_from_okbits(uint64_t ok, double (&acc)[TT])
{
    constexpr int KA = TT / 2;
    static_for<0, TT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        acc[t] = ((ok >> t) & 1ull) ? -0.0 : -1.0;                    // returns.append(r_new.copy())
    });
    {
        KernargDouble *pw = kernarg_pow();
        double ck[KA + 1];
        static_for<1, KA + 1>([&](auto kc) { ck[decltype(kc)::value] = pw[decltype(kc)::value]; });
        static_for<1, KA + 1>([&](auto kc) { pin_sgpr(ck[decltype(kc)::value]); });
        static_for<1, TT>([&](auto tc) {                                // lags 1 .. KA
            constexpr int t = decltype(tc)::value;
            if (!((ok >> t) & 1ull)) {
                static_for<(t > KA ? t - KA : 0), t>([&](auto uc) {
                    constexpr int t_ = decltype(uc)::value;
                    acc[t_] = acc[t_] - ck[t - t_];
                });
            }
        });
    }
    if constexpr (KA + 1 < TT) {
        KernargDouble *pw = kernarg_pow();
        double ck[TT];
        static_for<KA + 1, TT>([&](auto kc) { ck[decltype(kc)::value] = pw[decltype(kc)::value]; });
        static_for<KA + 1, TT>([&](auto kc) { pin_sgpr(ck[decltype(kc)::value]); });
        static_for<KA + 1, TT>([&](auto tc) {                           // lags KA+1 .. TT-1
            constexpr int t = decltype(tc)::value;
            if (!((ok >> t) & 1ull)) {
                static_for<0, t - KA>([&](auto uc) {
                    constexpr int t_ = decltype(uc)::value;
                    acc[t_] = acc[t_] - ck[t - t_];
                });
            }
        });
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The G rows of 64 consecutive envs (lane = env, first env e0, envs >= n are not written) leave through a
// wave-private LDS tile of 16 rows -- the tile is the output image itself, so every quarter of the wave's
// rows streams out as one contiguous 16 * TT * 8-byte piece with 16-byte stores.
constexpr int G_TILE_ROWS = 16;
template <int TT, int TR = G_TILE_ROWS>
__device__ __forceinline__ void wave_store_returns(const double (&acc)[TT], double *tile, double *__restrict__ G,
                                                   int64_t e0, int64_t n, int lane)
{
    constexpr bool PAIRS = TT % 2 == 0;          // rows are whole 16-byte pieces: 16-byte LDS and global accesses
#pragma unroll
    for (int c = 0; c < 64 / TR; c++) {
        if (c) wave_lds_sync();                                          // previous quarter streamed out
        if ((lane / TR) == c) {
            double *row = tile + (lane % TR) * TT;
            if constexpr (PAIRS) {
                static_for<0, TT / 2>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    reinterpret_cast<double2 *>(row)[i] = make_double2(acc[2 * i], acc[2 * i + 1]);
                });
            } else {
                static_for<0, TT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    row[i] = acc[i];
                });
            }
        }
        wave_lds_sync();
        const int64_t first = e0 + c * TR;
        const int64_t left = n - first;
        const int rows = left >= TR ? TR : (left > 0 ? (int)left : 0);
        double *o = G + first * TT;
        if constexpr (PAIRS) {
            const double2 *t2 = reinterpret_cast<const double2 *>(tile);
            double2 *o2 = reinterpret_cast<double2 *>(o);                // first * TT * 8 bytes: 16-byte aligned
            for (int f = lane; f < rows * (TT / 2); f += 64) st_d2<OUT_NT>(&o2[f], t2[f]);
        } else {
            for (int f = lane; f < rows * TT; f += 64) st_1<OUT_NT>(&o[f], tile[f]);
        }
    }
}

// The same for rows of which only the first T <= TT entries exist (k_returns_from_okbits_lane_rt): the tile keeps rows of
// TT doubles, the output rows are T doubles long, 16-byte stores when T is even.
template <int TT, int TR = G_TILE_ROWS>
__device__ __forceinline__ void wave_store_returns_rt(const double (&acc)[TT], double *tile, double *__restrict__ G,
                                                      int64_t e0, int64_t n, int lane, int T)
{
    static_assert(TT % 2 == 0, "rows are whole 16-byte pieces");
    const bool pairs = (T % 2 == 0) && ((reinterpret_cast<uintptr_t>(G) & 15u) == 0);
#pragma unroll
    for (int c = 0; c < 64 / TR; c++) {
        if (c) wave_lds_sync();
        if ((lane / TR) == c) {
            double *row = tile + (lane % TR) * TT;
            static_for<0, TT / 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (2 * i < T) reinterpret_cast<double2 *>(row)[i] = make_double2(acc[2 * i], acc[2 * i + 1]);
            });
        }
        wave_lds_sync();
        const int64_t first = e0 + c * TR;
        const int64_t left = n - first;
        const int rows = left >= TR ? TR : (left > 0 ? (int)left : 0);
        double *o = G + first * T;
        if (pairs) {                                   // first is a multiple of 16: the piece starts 16-byte aligned
            const int hp = T / 2;
            int r = 0, col = lane;
            while (col >= hp) { col -= hp; r++; }
            double2 *o2 = reinterpret_cast<double2 *>(o);
            for (int f = lane; f < rows * hp; f += 64) {
                st_d2<OUT_NT>(&o2[f], reinterpret_cast<const double2 *>(tile + r * TT)[col]);
                col += 64;
                while (col >= hp) { col -= hp; r++; }
            }
        } else {
            int r = 0, col = lane;
            while (col >= T) { col -= T; r++; }
            for (int f = lane; f < rows * T; f += 64) {
                st_1<OUT_NT>(&o[f], tile[r * TT + col]);
                col += 64;
                while (col >= T) { col -= T; r++; }
            }
        }
    }
}


// PIPE: software-pipelined mask requests (see the loop); used for the latency-bound batch sizes.
// The plain loop (PIPE = false: UNROLL requests, UNROLL steps, repeat) is what the bandwidth-bound
// fat shape runs: with >= 16 waves per CU other waves cover the round trips.
// RESET: the launch starts with reset_all_rollouts (rollout.py:79).  Compile-time because the
// elapsed counter is then the same for every lane (step t of the launch), so the time-limit
// test, the saturation and the generator's Weyl term run on the scalar unit.
// LB: the largest block the instantiation is ever launched with (the register budget follows from it:
// 128 VGPRs at 1024 lanes, 512 / waves-per-SIMD below).
// OKB: the launch also stores every env's step-ok word (bit t = the success condition held at step t, i.e. the
// step reward was -0.0f; T <= 64) into okbits[n] -- 8 bytes per env instead of a 4 T-byte reward stream -- from
// which k_returns_from_okbits computes the policy-gradient return.  (Computing G inside this launch, right
// after a wave's steps, was built and measured: 141-150 us against 52 + 83 us for the two launches at 1 M envs,
// profiles/r03_tune_pg_path.log -- the memory-bound step phase and the VALU-bound return phase of the same
// waves do not overlap -- and removed.)
// MULTI: `rollouts` consecutive rollouts in this ONE launch (each: reset, T steps, success count into its own counter
// word), rollout r on rows r*T .. r*T+T-1 of the streams -- exactly what `rollouts` launches in a row would do, without
// their fixed cost (5-6 us per launch, which is all a 65 536-env rollout has left to lose).  Envs are independent and a
// lane only ever touches its own, so nothing has to be synchronised between the rollouts of a launch; the counts go out
// as plain atomics that drain behind the next rollout's steps.
// Diagnostics for tools/tune_round3.hip (mode T) only: with BPG_TIMELINE defined, lane 0 of every block records the
// constant-rate clock at a few points of the launch into a buffer the tool sets.  Never defined in the library build.
#ifdef BPG_TIMELINE
__device__ unsigned long long *bpg_timeline = nullptr;
#define BPG_TL(slot) do { if (threadIdx.x == 0 && bpg_timeline) bpg_timeline[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define BPG_TL(slot) do { } while (0)
#endif
#ifndef BPG_ROLLOUT_MINB
#define BPG_ROLLOUT_MINB 1      // blocks per CU the register budget is sized for (an experiment of tools/tune_round3 only: see DESIGN.md section 4)
#endif
template <int SRC, bool NT, int EPL, int OUT, bool RESET, int UNROLL, bool PIPE, bool NTL, int LB, bool OKB = false,
          bool MULTI = false, bool RET = false>
__global__ void __launch_bounds__(LB, BPG_ROLLOUT_MINB)
k_rollout(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks_all,
          uint32_t *__restrict__ reward_all, uint8_t *__restrict__ flags_all, int64_t n, int T,
          int64_t ld_masks, int64_t ld_out,
          StepConst c, SynthConst s, ResetConst rc, CountArgs ca, uint64_t *__restrict__ okbits, int rollouts,
          uint32_t *__restrict__ ep_return_all)
{
    static_assert(!MULTI || (RESET && !OKB), "several rollouts per launch: each starts with its reset; no step-ok words");
    using Vec = typename LaneVec<EPL>::vec;
    using Fl = typename LaneVec<EPL>::fl;
    const int64_t nvec = n / EPL;               // the host guarantees n, ld_masks, ld_out % EPL == 0
    const int64_t ldm = ld_masks / EPL, ldo = ld_out / EPL;      // row strides of the (T, ld) streams
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (ca.zero_next && blockIdx.x == 0 && threadIdx.x == 0) *ca.zero_next = 0ull;
    BPG_TL(0);
    int tl_pass = 0;
    (void)tl_pass;
    for (int ro = 0; ro < (MULTI ? rollouts : 1); ro++) {
    const uint32_t *__restrict__ masks = masks_all + (MULTI ? (int64_t)ro * T * ld_masks : 0);
    uint32_t *__restrict__ reward_tb = reward_all + (MULTI ? (int64_t)ro * T * ld_out : 0);
    uint8_t *__restrict__ flags_tb = flags_all + (MULTI ? (int64_t)ro * T * ld_out : 0);
    uint32_t *__restrict__ ep_return = ep_return_all ? ep_return_all + (MULTI ? (int64_t)ro * n : 0) : nullptr;
    uint32_t nsucc = 0;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint32_t an[EPL] = {}, ee[EPL] = {};
        // a Variation reset (fetch_env.py:646-679) overwrites the whole state: nothing to read, and
        // no wave starts its episode behind a memory round trip
        if (!(RESET && rc.is_variation)) {
            unpack<EPL>(reinterpret_cast<const Vec *>(now)[v], an);
            unpack<EPL>(reinterpret_cast<const Vec *>(ever)[v], ee);
        }
        SynthLane sl[EPL];
        // step_core() with the state unpacked for the duration of the launch: the touching set,
        // the ever set (+ num_objs bits, untouched), the elapsed counter and the success latch (as
        // an all-ones / zero mask) each in a register of their own, and the two comparisons of
        // step_core turned into sign-bit arithmetic -- 11 instead of ~20 instructions per env-step.
        uint32_t cm[EPL], ev[EPL], el[EPL], lat[EPL], stk[EPL];
        uint64_t okb[EPL];                          // OKB: bit t = the success condition held at step t
        // RET: steps of this launch whose success condition held, counted for every env (one subtraction per
        // env-step): the undiscounted episode return is -(T - nok), bpg_rollout_args.d_episode_return.  Compile-time,
        // because the contacts-in-kernel form lives on occupancy and loses a wave per SIMD to the two extra registers.
        // With OKB the count is the population of the step-ok word and costs nothing per step.
        constexpr bool COUNT = RET && !OKB;
        uint32_t nok[EPL] = {};
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            if (RESET) reset_core(an[k], ee[k], EPL * v + k, rc, MULTI ? (uint32_t)ro : 0u, MULTI ? (int64_t)ro * n : 0);
            if (SRC == SRC_SYNTH) sl[k] = synth_lane(s, (uint32_t)(EPL * v + k), nobj_of(ee[k]));
            cm[k] = an[k] & PAIR_BITS;
            el[k] = RESET ? 0u : (an[k] >> ELAPSED_SHIFT) & ELAPSED_MAX;
            ev[k] = ee[k] & ~SUCC_BIT;
            lat[k] = (uint32_t)((int32_t)ee[k] >> 31);
            // pairs that are never demoted (StepConst); the synthetic source only emits pairs of present objects
            stk[k] = (SRC == SRC_SYNTH) ? 0u : sticky_of(c, ee[k]);
            okb[k] = 0ull;
        }
#ifdef BPG_TILED_EXPERIMENT
        // tools/tune_round3 mode L only: block-tiled streams [n / tile][T][tile] with tile = ld_masks = ld_out envs -- element
        // (t, e) at (e / tile) * T * tile + t * tile + e % tile -- instead of time-major rows (a row stride >= n selects rows)
        const bool tiled = ld_masks < n;
        const int64_t vt = tiled ? v / ldm : 0, vi = tiled ? v % ldm : v;
        const Vec *mrow = reinterpret_cast<const Vec *>(masks) + vt * T * ldm + vi;
        Vec *rrow = reinterpret_cast<Vec *>(reward_tb) + vt * T * ldo + vi;
        Fl *frow = reinterpret_cast<Fl *>(flags_tb) + vt * T * ldo + vi;
#else
        const Vec *mrow = reinterpret_cast<const Vec *>(masks) + v;
        Vec *rrow = reinterpret_cast<Vec *>(reward_tb) + v;
        Fl *frow = reinterpret_cast<Fl *>(flags_tb) + v;
#endif
        // ubit = 1 << (step within its batch of UNROLL): the step-ok bits of a batch collect in a 32-bit word with
        // compile-time masks and join the 64-bit word once per batch (OKB)
        uint32_t okpart[EPL] = {};
        auto one_step = [&](int t, const uint32_t *mt, uint32_t ubit) {
            uint32_t rb[EPL], fl[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                if (RESET) el[k] = (uint32_t)t;            // uniform: elapsed == step of the launch (T <= 1023)
                if (SRC == SRC_SYNTH) cm[k] = synth_mask(sl[k], el[k]) & c.valid;
                else cm[k] = (mt[k] & c.valid) | (cm[k] & stk[k]);
                ev[k] |= cm[k];
                const uint32_t bad = (c.want_touch & ~cm[k]) | (c.want_never & ev[k]);    // < 2^21
                const uint32_t okm = (uint32_t)((int32_t)(bad - 1u) >> 31);               // ~0 iff bad == 0
                lat[k] |= okm;
                if (COUNT) nok[k] -= okm;                                                 // okm is 0 or -1
                if (OKB) okpart[k] |= okm & ubit;
                el[k] = min(el[k] + 1u, ELAPSED_MAX);
                const uint32_t dn = ((c.max_steps - 1u - el[k]) >> 30) & (uint32_t)BPG_FLAG_DONE;   // el >= max_steps
                rb[k] = RBITS_FAIL ^ (okm & (RBITS_FAIL ^ RBITS_OK));
                fl[k] = (lat[k] & (uint32_t)BPG_FLAG_SUCCESS) | (okm & (uint32_t)BPG_FLAG_STEP_OK) | dn;
            }
            if (OUT & OUT_REWARD) st_vec<NT, EPL>(&rrow[(int64_t)t * ldo], rb);
            if (OUT & OUT_FLAGS) {
                uint32_t f = fl[0];
#pragma unroll
                for (int k = 1; k < EPL; k++) f |= fl[k] << (8 * k);
                st_1<NT>(&frow[(int64_t)t * ldo], (Fl)f);
            }
            // The contacts-in-kernel form is arithmetic-co-limited and lives on occupancy.  Left alone, the
            // scheduler interleaves the generators of all UNROLL steps (47 VGPRs in one build, 107-140 in the next,
            // depending on which side of the SGPR limit a few unrelated kernel arguments push it); a scheduling
            // fence per step keeps every step's ~70 instructions together: 50 VGPRs, 7-8 waves per SIMD.
            if (SRC == SRC_SYNTH) __builtin_amdgcn_sched_barrier(0);
        };
        auto flush_ok = [&](int t0) {
            if (OKB) {
#pragma unroll
                for (int k = 0; k < EPL; k++) {
                    okb[k] |= (uint64_t)okpart[k] << t0;
                    okpart[k] = 0u;
                }
            }
        };
        // Steps run in batches of UNROLL.  The masks of batch b+1 are requested BEFORE the rewards /
        // flags of batch b are stored: vector-memory operations retire in issue order (one vmcnt
        // counter for loads and stores), so a load issued behind a store cannot be consumed until
        // that store has been acknowledged by the memory side -- with the loads in front, a batch
        // waits for one memory latency instead of a store round trip plus a load round trip.
        const int nb = T / UNROLL;
        if constexpr (PIPE && SRC == SRC_MASKS) {
            Vec bufA[UNROLL], bufB[UNROLL];
            if (nb > 0) {
#pragma unroll
                for (int u = 0; u < UNROLL; u++) bufA[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)u * ldm]);
            }
            auto batch = [&](int t0, const Vec *cur, Vec *nxt, bool prefetch) {
                if (prefetch) {
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) nxt[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)(t0 + UNROLL + u) * ldm]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the requests in front of this batch's stores
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    uint32_t mt[EPL];
                    unpack<EPL>(cur[u], mt);
                    one_step(t0 + u, mt, 1u << u);
                }
                flush_ok(t0);
                if (t0 == 0 && tl_pass < 2) BPG_TL(1 + 4 * tl_pass);
            };
            // Two batches per trip with ping-pong register buffers (a register copy cur = nxt would
            // have to wait for the prefetch it copies), and the tail peeled so that every batch in
            // the loop prefetches unconditionally (a conditional prefetch makes the compiler place
            // the wait for the merged worst case at the loop head).
            int b = 0;
            for (; b + 2 < nb; b += 2) {
                batch(b * UNROLL, bufA, bufB, true);
                batch((b + 1) * UNROLL, bufB, bufA, true);
            }
            if (nb - b == 2) {
                batch(b * UNROLL, bufA, bufB, true);
                batch((b + 1) * UNROLL, bufB, bufA, false);
            } else if (nb - b == 1) {
                batch(b * UNROLL, bufA, bufB, false);
            }
        } else {
            for (int b = 0; b < nb; b++) {
                Vec cur[UNROLL];
                if (SRC == SRC_MASKS) {
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) cur[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)(b * UNROLL + u) * ldm]);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    uint32_t mt[EPL] = {};
                    if (SRC == SRC_MASKS) unpack<EPL>(cur[u], mt);
                    one_step(b * UNROLL + u, mt, 1u << u);
                }
                flush_ok(b * UNROLL);
                if (b == 0 && tl_pass < 2) BPG_TL(1 + 4 * tl_pass);
            }
        }
        if (tl_pass < 2) BPG_TL(2 + 4 * tl_pass);
        for (int t = nb * UNROLL; t < T; t++) {          // T % UNROLL leftover steps
            uint32_t mt[EPL] = {};
            if (SRC == SRC_MASKS) unpack<EPL>(mrow[(int64_t)t * ldm], mt);
            one_step(t, mt, 1u << (t - nb * UNROLL));
        }
        flush_ok(nb * UNROLL);
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            an[k] = cm[k] | (el[k] << ELAPSED_SHIFT);
            ee[k] = ev[k] | (lat[k] & SUCC_BIT);
            nsucc += lat[k] & 1u;
        }
        // (plain stores: nontemporal state stores were measured 0.8 us SLOWER per 1 M-env launch, 84.8 against 84.0 us
        // on one box, profiles/r04_ab_state_stores.log -- the 8 MB a launch leaves dirty are not what the boundary costs)
        reinterpret_cast<Vec *>(now)[v] = pack<EPL>(an);
        reinterpret_cast<Vec *>(ever)[v] = pack<EPL>(ee);
        if constexpr (OKB) {
#pragma unroll
            for (int k = 0; k < EPL; k++) okbits[EPL * v + k] = okb[k];          // adjacent: one 8 * EPL-byte store
        }
        if constexpr (RET || OKB) {
            if (ep_return) {                        // uniform (an OKB launch may or may not be asked for the return)
                // r[0] + r[1] + ... of T float32 rewards in {-1.0f, -0.0f}: -(number of -1.0f), -0.0f if there is none
                uint32_t rt[EPL];
#pragma unroll
                for (int k = 0; k < EPL; k++) {
                    const uint32_t held = OKB ? (uint32_t)__popcll(okb[k]) : nok[k];
                    rt[k] = __float_as_uint(-(float)((uint32_t)T - held));
                }
                reinterpret_cast<Vec *>(ep_return)[v] = pack<EPL>(rt);
            }
        }
        if (tl_pass < 2) BPG_TL(3 + 4 * tl_pass);
        tl_pass++;
    }
    BPG_TL(11);
    if (ca.count) {
        if (MULTI) {
            block_count_add(nsucc, CountArgs{ca.count + ro, nullptr, nullptr});
            __syncthreads();                       // the partial sums are written again by the next rollout
        } else {
            block_count_add(nsucc, ca);
        }
    }
    BPG_TL(12);
    }
#ifdef BPG_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);                 // the tool wants to see the stores drain
    BPG_TL(13);
#endif
}

// ------------------------------ float views ---------------------------------
// flat index f of the SIDE x SIDE matrix -> pair index, packed 4 x 5 bits per
// float4 chunk (even SIDE) or one byte per float (odd SIDE).
struct PairTable {
    uint32_t quad[9];     // chunk q covers flat 4q .. 4q+3
    uint8_t flat[36];
    float goal[36];       // desired_goal row as float32
};

__device__ __forceinline__ float tern(uint32_t now, uint32_t ever, uint32_t p)
{
    // achieved_goal entry: 1 touching now, 0 touched before, -1 never (fetch_env.py:129-134)
    float v = ((ever >> p) & 1u) ? 0.0f : -1.0f;
    return ((now >> p) & 1u) ? 1.0f : v;
}

// Writes `rows` consecutive matrices per env for the `nvalid` envs of one wave.
// The wave's output region is contiguous, so lanes take consecutive float4
// chunks (1 KiB per wave store); (env, row) is tracked incrementally.
// hist[e * R + r] = (now, ever) of env e at row r, in LDS.
// `out` points at row `row0` of the first env; env e's rows start env_stride floats later
// (env_stride = rows * DIMG when the whole region is contiguous).
template <int SIDE, bool NT>
__device__ __forceinline__ void wave_write_matrices(float *__restrict__ out, int nvalid, int rows, int env_stride,
                                                    const uint2 *hist, int R, const PairTable *tab, int lane,
                                                    int nlanes = 64)
{
    constexpr int DIMG = SIDE * SIDE;
    if (SIDE % 2 == 0) {
        constexpr int QPR = DIMG / 4;
        const int per_env = rows * QPR;
        const int total = nvalid * per_env;
        const int es4 = env_stride >> 2;
        float4 *o4 = reinterpret_cast<float4 *>(out);
        // one chunk stream per lane: a 4-way interleaved variant (four LDS reads / stores in flight per
        // iteration) measured SLOWER (3.0 vs 2.45 ms for the full episode at 1 M envs)
        int e = 0, rem = lane;
        while (rem >= per_env) { rem -= per_env; e++; }
        for (int cidx = lane; cidx < total; cidx += nlanes) {
            const int r = rem / QPR, q = rem - r * QPR;
            const uint2 h = hist[e * R + r];
            const uint32_t pw = tab->quad[q];
            float4 v;
            v.x = tern(h.x, h.y, pw & 31u);
            v.y = tern(h.x, h.y, (pw >> 5) & 31u);
            v.z = tern(h.x, h.y, (pw >> 10) & 31u);
            v.w = tern(h.x, h.y, (pw >> 15) & 31u);
            st_f4<NT>(&o4[(int64_t)e * es4 + rem], v);
            rem += nlanes;
            while (rem >= per_env) { rem -= per_env; e++; }
        }
    } else {
        const int per_env = rows * DIMG;
        const int total = nvalid * per_env;
        int e = 0, rem = lane;
        while (rem >= per_env) { rem -= per_env; e++; }
        for (int f = lane; f < total; f += nlanes) {
            const int r = rem / DIMG, k = rem - r * DIMG;
            const uint2 h = hist[e * R + r];
            st_1<NT>(&out[(int64_t)e * env_stride + rem], tern(h.x, h.y, tab->flat[k]));
            rem += nlanes;
            while (rem >= per_env) { rem -= per_env; e++; }
        }
    }
}

// obs['achieved_goal'] for the whole batch, float32 or float64; env e's matrix
// goes to out + e * ld (ld == dimg: dense (n, dimg); ld > dimg: a row slot of a
// batch-major episode buffer).  VEC4: 16-byte stores, needs ld % 4 == 0.
template <int SIDE, typename OutT, bool VEC4>
__global__ void __launch_bounds__(64)
k_get_ag(const uint32_t *__restrict__ now, const uint32_t *__restrict__ ever, int64_t n,
         OutT *__restrict__ out, int64_t ld, PairTable tabv)
{
    constexpr int DIMG = SIDE * SIDE;
    __shared__ uint2 hist[64];
    __shared__ PairTable tab;
    const int lane = threadIdx.x;
    if (lane == 0) tab = tabv;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    if (lane < nvalid) hist[lane] = make_uint2(now[base + lane], ever[base + lane]);
    __syncthreads();
    OutT *o = out + base * ld;
    if (VEC4) {
        constexpr int QPR = DIMG / 4;
        for (int cidx = lane; cidx < nvalid * QPR; cidx += 64) {
            const int e = cidx / QPR, q = cidx - e * QPR;
            const uint2 h = hist[e];
            const uint32_t pw = tab.quad[q];
            float4 v;
            v.x = tern(h.x, h.y, pw & 31u);
            v.y = tern(h.x, h.y, (pw >> 5) & 31u);
            v.z = tern(h.x, h.y, (pw >> 10) & 31u);
            v.w = tern(h.x, h.y, (pw >> 15) & 31u);
            st_f4<OUT_NT>(reinterpret_cast<float4 *>(reinterpret_cast<float *>(o) + e * ld + 4 * q), v);
        }
    } else {
        for (int f = lane; f < nvalid * DIMG; f += 64) {
            const int e = f / DIMG, k = f - e * DIMG;
            const uint2 h = hist[e];
            st_1<OUT_NT>(&o[e * ld + k], (OutT)tern(h.x, h.y, tab.flat[k]));
        }
    }
}

// One env.step as the rollout worker consumes it (rollout.py:121-131): the step, the
// float32 obs['achieved_goal'] row (at out + e*ld: a dense (n, dimg) array or the time slot
// of an episode buffer) and info['is_success'] as float32, in ONE launch -- the packed state
// never makes a second trip through HBM and the host issues one launch instead of five.
// A wave owns tiles of 64 consecutive envs: lane = env for the step, then the wave streams the tile's
// contiguous 64 x dimg float region from a wave-private LDS copy of the packed states -- no block-wide
// barrier, so the WPB waves of a block come and go on their own.  count (optional): += the number of envs
// whose has_succeeded latch is set after this step -- `successful = successes[-1]` of rollout.py:163-167
// inside the LAST step's launch of a rollout instead of a launch of its own.
// NT: the write-once outputs (achieved_goal rows, is_success, reward, flags, the action copy) leave as nontemporal
// stores.  Plain stores allocate in the L2 and are written back when the kernel ends, behind the last wave: in a
// graph-replayed train of step launches at 65 536 envs that put 1.4 us of store drain inside every launch and 0.8 us
// of write-back behind it (5.9 -> 4.1 us per step; 35.0 -> 33.3 us at 1 M envs, 152 -> 115 us at 4 M;
// tools/tune_round4.hip, profiles/r04_tune_step_train.log).  The packed state, re-read by the next step, stays plain.
#ifdef BPG_TIMELINE
// tools/tune_round4.hip: entry / exit time of every block of every node of a replayed train of step launches; the node
// is told by the time slot the launch writes its achieved_goal rows into
__device__ unsigned long long *bpg_tl_step = nullptr;
__device__ const float *bpg_tl_ag_base = nullptr;
__device__ long long bpg_tl_ag_slot = 1;
#define BPG_TL_STEP(slot, agp) do { if (threadIdx.x == 0 && bpg_tl_step) { \
        if (slot) __builtin_amdgcn_s_waitcnt(0); \
        const long long node_ = ((agp) - bpg_tl_ag_base) / bpg_tl_ag_slot; \
        bpg_tl_step[((size_t)node_ * gridDim.x + blockIdx.x) * 2 + (slot)] = wall_clock64(); } } while (0)
#else
#define BPG_TL_STEP(slot, agp) do { } while (0)
#endif
template <int SIDE, int SRC, bool VEC4, int WPB = 1, bool NT = false>
__global__ void __launch_bounds__(64 * WPB)
k_step_record(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
              uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, float *__restrict__ succ_f32,
              float *__restrict__ ag, int64_t ld, int64_t n, StepConst c, SynthConst s, PairTable tabv,
              const float4 *__restrict__ u_in, float4 *__restrict__ u_out, CountArgs ca, float *__restrict__ ret_acc)
{
    constexpr int DIMG = SIDE * SIDE;
    __shared__ uint2 hist_all[WPB][64];
    __shared__ PairTable tab_all[WPB];
    BPG_TL_STEP(0, ag);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint2 *hist = hist_all[wave];
    PairTable *tab = &tab_all[wave];
    if (lane == 0) *tab = tabv;
    uint32_t nsucc = 0;
    const int64_t ntiles = (n + 63) >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < ntiles; tile += (int64_t)gridDim.x * WPB) {
        const int64_t base = tile * 64;
        const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
        if (lane < nvalid) {
            const int64_t i = base + lane;
            uint32_t a = now[i], e = ever[i], rb, fl;
            const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
            step_core(a, e, mk, c, rb, fl);
            now[i] = a; ever[i] = e;
            hist[lane] = make_uint2(a, e);
            nsucc += fl & BPG_FLAG_SUCCESS;
            if (reward) st_1<NT>(&reward[i], rb);
            if (flags) st_1<NT>(&flags[i], (uint8_t)fl);
            if (succ_f32) st_1<NT>(&succ_f32[i], (fl & BPG_FLAG_SUCCESS) ? 1.0f : 0.0f);
            if (u_out) st_f4<NT>(&u_out[i], u_in[i]);   // acts.append(u.copy()) of rollout.py:147, same launch
            if (ret_acc) ret_acc[i] = ret_acc[i] + __uint_as_float(rb);          // bpg_step_args.d_return_acc (re-read next step: plain)
        }
        wave_lds_sync();
        if (ag) {
            float *o = ag + base * ld;
            if (VEC4) {
                constexpr int QPR = DIMG / 4;
                for (int cidx = lane; cidx < nvalid * QPR; cidx += 64) {
                    const int e = cidx / QPR, q = cidx - e * QPR;
                    const uint2 h = hist[e];
                    const uint32_t pw = tab->quad[q];
                    float4 v;
                    v.x = tern(h.x, h.y, pw & 31u);
                    v.y = tern(h.x, h.y, (pw >> 5) & 31u);
                    v.z = tern(h.x, h.y, (pw >> 10) & 31u);
                    v.w = tern(h.x, h.y, (pw >> 15) & 31u);
                    st_f4<NT>(reinterpret_cast<float4 *>(o + e * ld + 4 * q), v);
                }
            } else {
                for (int f = lane; f < nvalid * DIMG; f += 64) {
                    const int e = f / DIMG, k = f - e * DIMG;
                    const uint2 h = hist[e];
                    st_1<NT>(&o[e * ld + k], tern(h.x, h.y, tab->flat[k]));
                }
            }
        }
        wave_lds_sync();                                 // the tile is written again by the next trip
    }
    if (ca.count) block_count_add(nsucc, ca);            // uniform
    BPG_TL_STEP(1, ag);
}

// Geom-id contact lists -> packed pair masks (fetch_env.py:158-167 up to the matrix write), so
// a backend that produces contact lists can feed the mask-based fused kernels.
template <int VW>
__global__ void __launch_bounds__(256)
// T = 1, ld = n: one step's lists.  T > 1: the lists of a whole recorded rollout, time-major (T, n, maxcon, 2) / (T, n), into the
// (T, ld) mask stream of the fused rollout; num_grey (optional): the model of env e is the one of the episode that is ABOUT
// to be drawn (Variation picks its sim by num_objs = 4 + num_grey, fetch_env.py:647-651), not the one in the state.
k_contacts_to_masks(const uint32_t *__restrict__ ever, const int16_t *__restrict__ contacts,
                    const int32_t *__restrict__ ncon, int maxcon, const int8_t *__restrict__ id2obj,
                    int is_variation, uint32_t valid, uint32_t *__restrict__ out, int64_t n, int T, int64_t ld,
                    const int32_t *__restrict__ num_grey)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, total = n * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = T == 1 ? 0 : i / n, e = i - t * n;
        int model = 0;
        if (is_variation) {
            model = num_grey ? num_grey[e] : (int)nobj_of(ever[e]) - 4;
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        const int8_t *tab = s_tab + model * BPG_MAX_GEOMS;
        int nc = ncon[i];
        nc = nc < 0 ? 0 : (nc > maxcon ? maxcon : nc);
        const uint32_t *row = reinterpret_cast<const uint32_t *>(contacts) + i * maxcon;
        out[t * ld + e] = row_pair_mask<VW>(tab, row, nc) & valid;
    }
}

// The same for RAGGED lists (CSR): env i owns contacts offsets[i] .. offsets[i+1]-1 of one packed
// (total, 2) geom-id array -- what a physics backend produces when it compacts sim.data.contact
// of all envs instead of padding every env to the longest list.  Consecutive lanes own consecutive
// segments, so a wave's loads fall into one contiguous stretch of the array.
__global__ void __launch_bounds__(256)
k_contacts_csr_to_masks(const uint32_t *__restrict__ ever, const uint32_t *__restrict__ pairs,
                        const int64_t *__restrict__ offsets, int64_t total, const int8_t *__restrict__ id2obj,
                        int is_variation, uint32_t valid, uint32_t *__restrict__ out, int64_t n)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int model = 0;
        if (is_variation) {
            model = (int)nobj_of(ever[i]) - 4;
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        const int8_t *tab = s_tab + model * BPG_MAX_GEOMS;
        int64_t lo = offsets[i], hi = offsets[i + 1];
        lo = lo < 0 ? 0 : (lo > total ? total : lo);                 // never index outside the packed array
        hi = hi < lo ? lo : (hi > total ? total : hi);
        uint32_t m = 0;
        for (int64_t k = lo; k < hi; k++) m |= pair_bit_of(tab, pairs[k]);
        out[i] = m & valid;
    }
}

// ------------------------------ observation assembly ------------------------
// obs['observation'] of _get_obs (fetch_env.py:187-228, Variation :567-621) from the
// quantities the reference reads off the simulator; float64 products / differences are left
// unfused so the result is bit-identical to the reference's NumPy arithmetic.
// One wave = 64 envs.  Phase 1, lane = env: the env's 10 + 12 * nb inputs are read with
// per-lane strides of 16 / 24 bytes (consecutive lanes -> one contiguous stretch of each array),
// every entry of the row is computed once and parked in an LDS tile [64][dimo] (odd row length
// -> conflict-free column writes).  Phase 2: the wave streams the tile, which is the contiguous
// 64 * dimo stretch of the output, with consecutive lanes on consecutive addresses.  (Round 1
// computed one output element per lane of a grid-stride loop: two integer divisions, a state
// load and a dependent gather per element, one request in flight per lane -- 900 us for 1 M
// Variation envs, 0.12 of HBM peak.)
struct ObsConst {
    int32_t dimo, nb_max, is_variation, vec;    // vec: the arrays are 16-byte aligned
    int32_t colors[BPG_MAX_OBJS];
    double dt;
};

// An env's N consecutive doubles into registers: 16-byte loads when the row is an even number of
// doubles (and the array 16-byte aligned), so that every 32-byte sector is requested by two instructions
// instead of four.
template <int N>
__device__ __forceinline__ void load_row_f64(const double *__restrict__ p, bool vec, double *o)
{
    if (N % 2 == 0 && vec) {
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            const double2 v = reinterpret_cast<const double2 *>(p)[k];
            o[2 * k] = v.x; o[2 * k + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) o[k] = p[k];
    }
}

// NB = block slots of the variant (num_objs - 2): compile-time, so that the env's 10 + 12 * NB inputs
// are all requested before the first one is used and live in registers with constant indices.
template <typename OutT, int NB>
__global__ void __launch_bounds__(64)
k_assemble_obs(const uint32_t *__restrict__ ever, int64_t n, bpg_phys_state ph, ObsConst oc,
               OutT *__restrict__ out, int64_t ld)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    OutT *tile = reinterpret_cast<OutT *>(smem);                   // [64][dimo]
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    const int dimo = oc.dimo;
    bool bad = false;                          // any NaN among the values this lane writes (ph.d_nan_flag)
    auto put = [&](OutT *p, double v) { bad |= (v != v); *p = (OutT)v; };
    if (lane < nvalid) {
        const int64_t e = base + lane;
        OutT *row = tile + lane * dimo;
        double gp[3], gvp[3], gs[2], gvl[2], pos[NB * 3], rot[NB * 3], vp[NB * 3], vr[NB * 3];
        const uint32_t ev = ever[e];
        load_row_f64<3>(ph.d_grip_pos + e * 3, false, gp);
        load_row_f64<3>(ph.d_grip_velp + e * 3, false, gvp);
        load_row_f64<2>(ph.d_gripper_state + e * 2, oc.vec != 0, gs);
        load_row_f64<2>(ph.d_gripper_vel + e * 2, oc.vec != 0, gvl);
        load_row_f64<NB * 3>(ph.d_obj_pos + e * (NB * 3), oc.vec != 0, pos);
        load_row_f64<NB * 3>(ph.d_obj_rot + e * (NB * 3), oc.vec != 0, rot);
        load_row_f64<NB * 3>(ph.d_obj_velp + e * (NB * 3), oc.vec != 0, vp);
        load_row_f64<NB * 3>(ph.d_obj_velr + e * (NB * 3), oc.vec != 0, vr);
        const int nblocks = (int)nobj_of(ev) - 2;
        double gv[3];
#pragma unroll
        for (int c = 0; c < 3; c++) gv[c] = __dmul_rn(gvp[c], oc.dt);                           // :191 / :571
        int k = 0;
        if (oc.is_variation) row[k++] = (OutT)(double)nblocks;                                  // :580
        // [grip_pos, gripper_state, grip_velp * dt, gripper_vel * dt]  (:222 / :580)
#pragma unroll
        for (int c = 0; c < 3; c++) put(&row[k++], gp[c]);
#pragma unroll
        for (int c = 0; c < 2; c++) put(&row[k++], gs[c]);
#pragma unroll
        for (int c = 0; c < 3; c++) put(&row[k++], gv[c]);
#pragma unroll
        for (int c = 0; c < 2; c++) put(&row[k++], __dmul_rn(gvl[c], oc.dt));
        const int per = oc.is_variation ? 19 : 15;
#pragma unroll
        for (int blk = 0; blk < NB; blk++) {
            if (blk < nblocks) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    put(&row[k + c], pos[blk * 3 + c]);
                    put(&row[k + 3 + c], __dsub_rn(pos[blk * 3 + c], gp[c]));                                    // :210
                    put(&row[k + 6 + c], rot[blk * 3 + c]);
                    put(&row[k + 9 + c], __dsub_rn(__dmul_rn(vp[blk * 3 + c], oc.dt), gv[c]));                   // :207, :211
                    put(&row[k + 12 + c], __dmul_rn(vr[blk * 3 + c], oc.dt));                                    // :208
                }
                if (oc.is_variation) {
#pragma unroll
                    for (int c = 0; c < 4; c++) row[k + 15 + c] = (OutT)((oc.colors[blk + 2] == c) ? 1.0 : 0.0);  // one_hot_color :599
                }
            } else {
                for (int c = 0; c < per; c++) row[k + c] = (OutT)0.0;                                             // zero padding :606-607
            }
            k += per;
        }
    }
    if (ph.d_nan_flag && __ballot(bad) != 0ull && lane == 0) atomicOr(ph.d_nan_flag, 1u);
    __syncthreads();
    if (ld == dimo) {
        OutT *o = out + base * dimo;
        const int total = nvalid * dimo;
        for (int f = lane; f < total; f += 64) st_1<OUT_NT>(&o[f], tile[f]);
    } else {
        for (int e = 0; e < nvalid; e++)
            for (int k = lane; k < dimo; k += 64) st_1<OUT_NT>(&out[(base + e) * ld + k], tile[e * dimo + k]);
    }
}

// robot_env.py:58 clip + BlocksEnv._set_action (fetch_env.py:170-185): the 9-vector
// [pos_ctrl * 0.05 (float32 product), 1, 0, 1, 0, grip, grip] the reference hands to the
// physics (gripper targets zeroed when block_gripper), one env per lane.
// The rows are 72 bytes: a lane writing its own row would touch nine partial lines per store
// instruction, so the block parks its 256 rows in LDS and streams the contiguous 18 KB.
__global__ void __launch_bounds__(256)
k_set_action(const float *__restrict__ action, int64_t n, int block_gripper, double *__restrict__ out)
{
    __shared__ double tile[256 * 9];
    const int64_t nblk = (n + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t base = blk * 256;
        const int nvalid = (int)((n - base) < 256 ? (n - base) : 256);
        if ((int)threadIdx.x < nvalid) {
            const float4 a4 = reinterpret_cast<const float4 *>(action)[base + threadIdx.x];
            float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int k = 0; k < 4; k++) a[k] = a[k] < -1.0f ? -1.0f : (a[k] > 1.0f ? 1.0f : a[k]);
            double *o = tile + threadIdx.x * 9;
            o[0] = (double)(a[0] * 0.05f);
            o[1] = (double)(a[1] * 0.05f);
            o[2] = (double)(a[2] * 0.05f);
            o[3] = 1.0; o[4] = 0.0; o[5] = 1.0; o[6] = 0.0;
            o[7] = o[8] = block_gripper ? 0.0 : (double)a[3];
        }
        __syncthreads();
        double *o = out + base * 9;
        for (int f = threadIdx.x; f < nvalid * 9; f += 256) st_1<OUT_NT>(&o[f], tile[f]);
        __syncthreads();
    }
}

// rotations.mat2euler of gym.envs.robotics (call site fetch_env.py:205, :588).  THIRD-PARTY
// code that is not in the reference tree: restated from SURVEY.md A.4's recollection of gym
// 0.10 (cy = sqrt(m22^2 + m12^2), branch on cy > 4*eps, negated arctan2 forms) -- parity
// UNPINNED; a convenience for physics backends that hand over site rotation matrices.
__global__ void __launch_bounds__(256)
k_mat2euler(const double *__restrict__ mat, int64_t n, double *__restrict__ euler)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double *m = mat + i * 9;
        const double cy = sqrt(m[8] * m[8] + m[5] * m[5]);
        const bool cond = cy > 4.0 * 2.220446049250313e-16;
        euler[i * 3 + 2] = cond ? -atan2(m[1], m[0]) : -atan2(-m[3], m[4]);
        euler[i * 3 + 1] = -atan2(-m[2], cy);
        euler[i * 3 + 0] = cond ? -atan2(m[5], m[8]) : 0.0;
    }
}

// ------------------------------ reset placement -----------------------------
// _randomize_objects of the five env classes (fetch_env.py:328-336, 370-399, 448-517,
// 697-764, 777-787): rejection-sampled xy positions of the cubes, one env per lane.
// Draws arrive as standard variates in the reference's consumption order (per-env
// np_random stream `eu`, global np.random stream `gz`/`gu`), or are
// generated on the fly from the counter-based generator when the arrays are NULL.
struct PlaceConst {
    int32_t variant, is_variation, nb_max, test, challenge;
    double grip_x, grip_y, obj_range, max_obj_range, wrong_obj_range;
    const double *eu, *gz, *gu;
    int32_t n_eu, n_gz, n_gu;
    uint64_t seed0;
    const uint32_t *reset_counter;
    const uint8_t *env_mask;      // partial resets: only envs with a non-zero byte are placed
};

struct Draws {
    const double *eu, *gz, *gu;
    int n_eu, n_gz, n_gu, ie, iz, iu, fail;
    uint32_t key;
    bool synth;
    double z_spare;
    bool has_spare;

    __device__ __forceinline__ double u53(uint32_t stream, uint32_t ctr)
    {
        // counter-based uniform in [0,1) with 53 random bits (build-defined)
        const uint32_t a = hash32(key + stream + 2u * ctr * 0x9E3779B9U);
        const uint32_t b = hash32(key + stream + (2u * ctr + 1u) * 0x9E3779B9U);
        return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
    }
    __device__ __forceinline__ double next_eu()
    {
        if (synth) return u53(0x1000193u, (uint32_t)ie++);
        if (ie >= n_eu) { fail = 1; return 0.5; }
        return eu[ie++];
    }
    __device__ __forceinline__ double next_gu()
    {
        if (synth) return u53(0x2000193u, (uint32_t)iu++);
        if (iu >= n_gu) { fail = 1; return 0.5; }
        return gu[iu++];
    }
    __device__ __forceinline__ double next_gz()
    {
        if (synth) {
            if (has_spare) { has_spare = false; return z_spare; }
            const double u1 = u53(0x3000193u, (uint32_t)iz), u2 = u53(0x4000193u, (uint32_t)iz);
            iz++;
            const double r = sqrt(-2.0 * log(1.0 - u1));          // Box-Muller
            z_spare = r * sinpi(2.0 * u2);
            has_spare = true;
            return r * cospi(2.0 * u2);
        }
        if (iz >= n_gz) { fail = 1; return 1.0; }
        return gz[iz++];
    }
};

// RandomState.uniform(low, high) = low + (high - low) * random_sample()
__device__ __forceinline__ double unif(double lo, double hi, double u) { return lo + (hi - lo) * u; }
// np.linalg.norm of a 2-vector = sqrt(x.dot(x)); the 2-element ddot is fma(y, y, x*x) in the
// NumPy build the fixtures were recorded with (BLAS dependent in general)
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }
// fetch_env.py:19-32 constants in Python evaluation order
__device__ __forceinline__ bool out_of_table(double x, double y)
{
    const double TABLE_W = 0.25 - 0.05 / 2, TABLE_H = 0.35 - 0.05 / 2;
    const double TABLE_X = 1.05 + 0.25, TABLE_Y = 0.40 + 0.35;
    return fabs(x - TABLE_X) > TABLE_W || fabs(y - TABLE_Y) > TABLE_H;
}
// direction = normal(size=2) / norm; mag = uniform(lo, hi); centre + direction * mag
__device__ __forceinline__ void polar_about(Draws &d, double cx, double cy, double lo, double hi, double &x, double &y)
{
    double dx = d.next_gz(), dy = d.next_gz();
    const double n = norm2(dx, dy);
    dx = dx / n; dy = dy / n;
    const double mag = unif(lo, hi, d.next_gu());
    x = cx + dx * mag;
    y = cy + dy * mag;
}

constexpr int PLACE_MAX_ITERS = 100000;     // every rejection loop is bounded (exit condition for all waves)

__global__ void __launch_bounds__(256)
k_randomize_objects(const uint32_t *__restrict__ ever, int64_t n, PlaceConst pc, double *__restrict__ xy,
                    int32_t *__restrict__ status)
{
    const double MIN_BLOCK_DIST = 1.5 * 0.05;
    const double TABLE_W = 0.25 - 0.05 / 2, TABLE_H = 0.35 - 0.05 / 2;
    const double TABLE_X = 1.05 + 0.25, TABLE_Y = 0.40 + 0.35;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (pc.env_mask && pc.env_mask[i] == 0) continue;
        Draws d;
        d.synth = (pc.eu == nullptr);
        d.eu = pc.eu ? pc.eu + i * pc.n_eu : nullptr;
        d.gz = pc.gz ? pc.gz + i * pc.n_gz : nullptr;
        d.gu = pc.gu ? pc.gu + i * pc.n_gu : nullptr;
        d.n_eu = pc.n_eu; d.n_gz = pc.n_gz; d.n_gu = pc.n_gu;
        d.ie = d.iz = d.iu = d.fail = 0;
        d.has_spare = false; d.z_spare = 0.0;
        d.key = hash32(env_key(pc.seed0, (uint32_t)i, 0x51ED270BU) + *pc.reset_counter);
        double *o = xy + i * pc.nb_max * 2;
        const double gx = pc.grip_x, gy = pc.grip_y;
        double x, y;
        int it = 0;
        if (pc.variant == BPG_GRIPPER_TOUCH || pc.variant == BPG_TOPPLE_TOWER) {
            x = gx; y = gy;
            while (!d.fail && it++ < PLACE_MAX_ITERS && norm2(x - gx, y - gy) < 0.1) {
                x = gx + unif(-pc.obj_range, pc.obj_range, d.next_eu());
                y = gy + unif(-pc.obj_range, pc.obj_range, d.next_eu());
            }
            for (int k = 0; k < pc.nb_max; k++) { o[2 * k] = x; o[2 * k + 1] = y; }
        } else if (pc.variant == BPG_BLOCKS_TOUCH || pc.variant == BPG_BLOCKS_TOUCH_CURRICULUM) {
            const double r = pc.test ? pc.max_obj_range : pc.obj_range;
            const double x0 = gx + unif(-r / 2, r / 2, d.next_eu());
            const double y0 = gy + unif(-r / 2, r / 2, d.next_eu());
            do {
                polar_about(d, x0, y0, MIN_BLOCK_DIST, r, x, y);
            } while (!d.fail && it++ < PLACE_MAX_ITERS && out_of_table(x, y));
            o[0] = x0; o[1] = y0; o[2] = x; o[3] = y;
        } else if (pc.variant == BPG_BLOCKS_TOUCH_CHOOSE || pc.variant == BPG_BLOCKS_TOUCH_CHOOSE_CURRICULUM) {
            double r, wrong_r, max_wrong, min_r;
            if (pc.test || pc.challenge) { r = pc.max_obj_range; wrong_r = 0; }
            else { r = pc.obj_range; wrong_r = pc.wrong_obj_range; }
            if (pc.challenge) { max_wrong = 0.04; min_r = 0.15; }
            else { min_r = MIN_BLOCK_DIST; max_wrong = pc.max_obj_range; }
            double bx, by, qx, qy;
            do {
                bx = gx + unif(-r / 2, r / 2, d.next_eu());
                by = gy + unif(-r / 2, r / 2, d.next_eu());
            } while (!d.fail && it++ < PLACE_MAX_ITERS && out_of_table(bx, by));
            do {
                polar_about(d, bx, by, min_r, r, qx, qy);
            } while (!d.fail && it++ < PLACE_MAX_ITERS && out_of_table(qx, qy));
            const double cx = (bx + qx) / 2.0, cy = (by + qy) / 2.0;
            do {
                polar_about(d, cx, cy, wrong_r, max_wrong, x, y);
            } while (!d.fail && it++ < PLACE_MAX_ITERS &&
                     (out_of_table(x, y) || norm2(x - bx, y - by) < MIN_BLOCK_DIST ||
                      norm2(x - qx, y - qy) < MIN_BLOCK_DIST));
            o[2] = bx; o[3] = by;       // blue  = block 1
            o[0] = qx; o[1] = qy;       // green = block 0
            o[4] = x;  o[5] = y;        // grey  = block 2
        } else {                        // Variation
            const int num_blocks = (int)nobj_of(ever[i]) - 2;
            const double r = pc.test ? pc.max_obj_range : pc.obj_range;
            double px[4], py[4];
            double bx, by, qx, qy;
            do {
                bx = gx + unif(-r / 2, r / 2, d.next_eu());
                by = gy + unif(-r / 2, r / 2, d.next_eu());
            } while (!d.fail && it++ < PLACE_MAX_ITERS && out_of_table(bx, by));
            do {
                polar_about(d, bx, by, MIN_BLOCK_DIST, r, qx, qy);
            } while (!d.fail && it++ < PLACE_MAX_ITERS && out_of_table(qx, qy));
            o[2] = bx; o[3] = by; o[0] = qx; o[1] = qy;
            px[0] = bx; py[0] = by; px[1] = qx; py[1] = qy;
            int np_ = 2;
#pragma unroll
            for (int b = 2; b < 4; b++) {
                if (b < num_blocks) {
                    bool again;
                    do {
                        x = TABLE_X + unif(-TABLE_W, TABLE_W, d.next_eu());
                        y = TABLE_Y + unif(-TABLE_H, TABLE_H, d.next_eu());
                        bool hit = false;
                        for (int k = 0; k < 3; k++)
                            if (k < np_ && !hit && norm2(x - px[k], y - py[k]) < MIN_BLOCK_DIST) hit = true;
                        again = hit ? true : out_of_table(x, y);
                    } while (!d.fail && it++ < PLACE_MAX_ITERS && again);
                    o[2 * b] = x; o[2 * b + 1] = y;
                    px[b] = x; py[b] = y;
                    np_ = b + 1;
                }
            }
        }
        if (it >= PLACE_MAX_ITERS) d.fail = 1;
        if (status) status[i] = d.fail;
    }
}

// ------------------------------ episode writer ------------------------------
// rollout.py:100-172 for one wave = 64 consecutive envs per block.  Phase 1:
// lane = env, T fused steps, packed (now, ever) history of every step kept in
// LDS (row stride R odd -> conflict-free ds_write_b64).  Phase 2: the wave
// streams its contiguous slices of episode['ag'] (B,T+1,dimg), episode['g']
// (B,T,dimg) and episode['info_is_success'] (B,T,1) as float32.
template <int SIDE, int SRC, bool NT>
__global__ void __launch_bounds__(128)
k_episode(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
          uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb,
          float *__restrict__ ep_ag, float *__restrict__ ep_g, float *__restrict__ ep_succ,
          int64_t n, int T, int64_t ldm, int64_t ldo, int RC, int R, StepConst c, SynthConst s, PairTable tabv,
          ResetConst rc, CountArgs ca, uint64_t *__restrict__ okbits_out, float *__restrict__ ep_return)
{
    // The T+1 matrices of an episode are produced in chunks of RC rows (one chunk = the whole
    // episode for T <= 63): a short step phase (lane = env, state in registers, packed rows of
    // the chunk into LDS, [64][R] x 8 B with R = RC | 1 odd -> conflict-free ds_write_b64)
    // followed by the streaming phase for those rows.
    constexpr int DIMG = SIDE * SIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *hist = reinterpret_cast<uint2 *>(smem);                       // [64][R]
    PairTable *tab = reinterpret_cast<PairTable *>(smem + (size_t)64 * R * sizeof(uint2));
    const int lane = threadIdx.x, nlanes = blockDim.x;     // wave 0 (lane < 64) steps the envs, every wave streams
    if (lane == 0) *tab = tabv;
    if (ca.zero_next && blockIdx.x == 0 && lane == 0) *ca.zero_next = 0ull;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    const bool active = lane < nvalid;                        // nvalid <= 64: only wave 0 can be active
    const int64_t i = base + (active ? lane : 0);
    uint32_t a = 0, e = 0;
    uint64_t okb = 0ull;                 // bit t = the success condition held at step t (t < 64): the step rewards
    uint32_t nfail = 0;                  // steps whose success condition did not hold: the episode return is -nfail
    SynthLane sl = {};
    if (active) {
        a = now[i]; e = ever[i];
        if (rc.enabled) reset_core(a, e, i, rc);
        if (SRC == SRC_SYNTH) sl = synth_lane(s, (uint32_t)i, nobj_of(e));
    }
    uint2 *my = hist + lane * R;
    constexpr int EP_BATCH = 10;        // masks of 10 steps are requested together: one memory
                                        // latency per batch instead of one per step
    for (int r0 = 0; r0 <= T; r0 += RC) {
        const int r1 = (r0 + RC < T + 1) ? r0 + RC : T + 1;      // rows [r0, r1) of every env
        if (active) {
            if (r0 == 0) my[0] = make_uint2(a, e);               // row 0: initial_ag (rollout.py:82-83)
            const int tb = (r0 == 0) ? 0 : r0 - 1, te = r1 - 1;  // steps tb .. te-1 produce rows tb+1 .. te
            for (int t0 = tb; t0 < te; t0 += EP_BATCH) {
                uint32_t mb[EP_BATCH];
                if (SRC == SRC_MASKS) {
#pragma unroll
                    for (int u = 0; u < EP_BATCH; u++)
                        mb[u] = (t0 + u < te) ? masks[(int64_t)(t0 + u) * ldm + i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < EP_BATCH; u++) {
                    const int t = t0 + u;
                    if (t < te) {
                        uint32_t m, rb, fl;
                        if (SRC == SRC_MASKS) m = mb[u];
                        else m = synth_mask(sl, (a >> ELAPSED_SHIFT) & ELAPSED_MAX);
                        step_core(a, e, m, c, rb, fl);
                        my[t + 1 - r0] = make_uint2(a, e);
                        okb |= (uint64_t)((fl >> 2) & 1u) << (t & 63);
                        nfail += ((fl >> 2) & 1u) ^ 1u;
                        if (reward_tb) reward_tb[(int64_t)t * ldo + i] = rb;
                        if (flags_tb) flags_tb[(int64_t)t * ldo + i] = (uint8_t)fl;
                    }
                }
            }
        }
        __syncthreads();
        if (ep_ag)
            wave_write_matrices<SIDE, NT>(ep_ag + (base * (int64_t)(T + 1) + r0) * DIMG, nvalid, r1 - r0,
                                          (T + 1) * DIMG, hist, R, tab, lane, nlanes);
        if (ep_succ) {
            // info['is_success'] after step t = latch in row t+1 (rollout.py:127-128,131)
            const int rs = (r0 == 0) ? 1 : r0, cnt = r1 - rs;    // rows rs .. r1-1 -> steps rs-1 .. r1-2
            float *o = ep_succ + base * (int64_t)T + (rs - 1);
            const int total = nvalid * cnt;
            int ee = 0, rem = lane;
            while (cnt > 0 && rem >= cnt) { rem -= cnt; ee++; }
            for (int f = lane; f < total; f += nlanes) {
                st_1<NT>(&o[(int64_t)ee * T + rem], (hist[ee * R + (rs - r0) + rem].y >> 31) ? 1.0f : 0.0f);
                rem += nlanes;
                while (rem >= cnt) { rem -= cnt; ee++; }
            }
        }
        __syncthreads();                                         // the tile is reused by the next chunk
    }
    if (active) {
        now[i] = a; ever[i] = e;
        // the step rewards of the episode as one word per env: k_returns_from_okbits turns it into the env's
        // G row (policy_gradient/rollout.py:255-258)
        if (okbits_out) okbits_out[i] = okb;
        if (ep_return) ep_return[i] = -(float)nfail;              // bpg_rollout_args.d_episode_return
    }
    if (ca.count && lane < 64) {
        const unsigned long long nb = (unsigned long long)__popcll(__ballot(active && (e >> 31) != 0));
        if (lane == 0) shard_count_add(nb, ca);
    }
    if (ep_g) {
        // episode['g']: T copies of the constant goal row per env (rollout.py:151)
        float *o = ep_g + base * (int64_t)T * DIMG;
        const int total = nvalid * T * DIMG;
        if (SIDE % 2 == 0) {
            constexpr int QPR = DIMG / 4;
            float4 *o4 = reinterpret_cast<float4 *>(o);
            for (int cidx = lane; cidx < total / 4; cidx += nlanes) {
                const int q = cidx % QPR;
                st_f4<NT>(&o4[cidx], make_float4(tab->goal[4 * q], tab->goal[4 * q + 1], tab->goal[4 * q + 2],
                                                 tab->goal[4 * q + 3]));
            }
        } else {
            for (int f = lane; f < total; f += nlanes) st_1<NT>(&o[f], tab->goal[f % DIMG]);
        }
    }
}

// ------------------------------ expert input trimming ------------------------
// RolloutStudent.trim of the policy-gradient worker (policy_gradient/rollout.py:105-171, batched branch): what an
// expert trained on a smaller puzzle is handed when it is in charge of a step (:196-200).
//   g_, ag_: the entries (i, j) with i, j < num_objs of the max_num_objs x max_num_objs goal matrices (:141-151), float32;
//   o_ (Variation, :152-168): o[1 : 1 + ENV_FEATURES] followed by the BLOCK_BASE_FEATURES of every block slot whose
//       colour one-hot has its (first) maximum at GREEN or BLUE, in slot order, as float64 (the reference concatenates
//       into a float64 array); a row with another number of such slots than num_objs - 2 is the reference's
//       ValueError (`block_features[i] = temp`, :166): reported through bad_flag, the row's missing picks read slot 0;
//   o_ (every other env, :169-170): o[:, :dimo], float32.
// One block = 64 envs: the rows are staged in LDS with batched 16-byte loads, lane = env picks the coloured slots,
// then all lanes stream the contiguous output regions.
constexpr int TRIM_ENV_FEATURES = 10, TRIM_BLOCK_BASE = 15, TRIM_COLORS = 4, TRIM_BLOCK = TRIM_BLOCK_BASE + TRIM_COLORS;
constexpr int TRIM_GREEN = 2, TRIM_BLUE = 3, TRIM_MAX_SLOTS = 8, TRIM_ENVS = 64;
struct TrimConst {
    int dimo_in, dimg_in, dimo_out, dimg_out;
    int num_blocks, max_blocks;     // Variation: slots kept / slots present in o
    bool vec_o;                     // o starts on a 16-byte boundary (every block's 64 rows then do)
    uint8_t sel[64];                // source column of g_ / ag_ column k
};

// Stages `total` consecutive floats (16-byte aligned start) from global memory into LDS, unchanged layout: every lane
// requests TRIM_U float4 before the first one is consumed (one memory round trip per TRIM_U * 4 KB of the block).
constexpr int TRIM_U = 6;
__device__ __forceinline__ void trim_stage(float *__restrict__ dst, const float *__restrict__ src, int total, int tid, bool vec)
{
    const int nq = vec ? total >> 2 : 0;            // vec: src is 16-byte aligned
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int q0 = tid; q0 < nq; q0 += 256 * TRIM_U) {
        float4 v[TRIM_U];
#pragma unroll
        for (int u = 0; u < TRIM_U; u++) {
            const int q = q0 + u * 256;
            v[u] = s4[q < nq ? q : nq - 1];           // clamped: no lane-dependent definedness of v
        }
#pragma unroll
        for (int u = 0; u < TRIM_U; u++)
            if (q0 + u * 256 < nq) d4[q0 + u * 256] = v[u];
    }
    for (int f = 4 * nq + tid; f < total; f += 256) dst[f] = src[f];
}

template <bool VAR>
__global__ void __launch_bounds__(256)
k_trim(const float *__restrict__ o, const float *__restrict__ g, const float *__restrict__ ag, int64_t n, TrimConst tc,
       void *__restrict__ o_out, float *__restrict__ g_out, float *__restrict__ ag_out, uint32_t *__restrict__ bad_flag)
{
    // LDS (Variation only): the block's o rows exactly as they lie in global memory (64 rows are a multiple of 16
    // bytes whatever the row length), then the source column of every o_ column.  The goal entries are gathered
    // straight from global memory (a wave reads 16 of the 36 floats of 4 consecutive rows: the same few lines):
    // requested first, stored last, so their latency hides behind the observation part.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *t_o = reinterpret_cast<float *>(smem);                    // [64][dimo_in]
    __shared__ uint8_t pick[TRIM_ENVS][TRIM_MAX_SLOTS];
    __shared__ uint8_t col_slot[TRIM_ENV_FEATURES + TRIM_BLOCK_BASE * TRIM_MAX_SLOTS];     // 255: env feature
    __shared__ uint16_t col_off[TRIM_ENV_FEATURES + TRIM_BLOCK_BASE * TRIM_MAX_SLOTS];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * TRIM_ENVS;
    const int nvalid = (int)((n - base) < TRIM_ENVS ? (n - base) : TRIM_ENVS);
    constexpr int TRIM_GU = 4;                   // 64 x 16 goal entries = 4 per lane
    const int gtotal = nvalid * tc.dimg_out;
    const float *pg = g + base * tc.dimg_in, *pa = ag + base * tc.dimg_in;
    float vg[TRIM_GU], va[TRIM_GU];
#pragma unroll
    for (int u = 0; u < TRIM_GU; u++) {
        int f = tid + u * 256;
        f = f < gtotal ? f : gtotal - 1;
        const int e = f / tc.dimg_out, k = f - e * tc.dimg_out;
        const int src = e * tc.dimg_in + tc.sel[k];
        vg[u] = pg[src];
        va[u] = pa[src];
    }
    if (VAR) {
        trim_stage(t_o, o + base * tc.dimo_in, nvalid * tc.dimo_in, tid, tc.vec_o);
        if (tid < tc.dimo_out) {
            const int k = tid;
            if (k < TRIM_ENV_FEATURES) { col_slot[k] = 255; col_off[k] = (uint16_t)(1 + k); }
            else {
                const int b = (k - TRIM_ENV_FEATURES) / TRIM_BLOCK_BASE;
                col_slot[k] = (uint8_t)b;
                col_off[k] = (uint16_t)(1 + TRIM_ENV_FEATURES + (k - TRIM_ENV_FEATURES) - b * TRIM_BLOCK_BASE);
            }
        }
        __syncthreads();
        if (tid < TRIM_ENVS) {
            int cnt = 0;
            if (tid < nvalid) {
                for (int j = 0; j < tc.max_blocks; j++) {
                    const float *c = t_o + tid * tc.dimo_in + 1 + TRIM_ENV_FEATURES + j * TRIM_BLOCK + TRIM_BLOCK_BASE;
                    // np.argmax (get_color, :24-26): first index of the maximum, a NaN counts as the maximum
                    int am = 0;
                    float best = c[0];
#pragma unroll
                    for (int q = 1; q < TRIM_COLORS; q++)
                        if (best == best && (c[q] > best || c[q] != c[q])) { best = c[q]; am = q; }
                    if (am == TRIM_GREEN || am == TRIM_BLUE) {
                        if (cnt < tc.num_blocks) pick[tid][cnt] = (uint8_t)j;
                        cnt++;
                    }
                }
                if (cnt != tc.num_blocks) atomicOr(bad_flag, 1u);
            }
            for (int q = cnt; q < tc.num_blocks; q++) pick[tid][q] = 0;
        }
    }
    __syncthreads();
    if (VAR) {
        double *oo = reinterpret_cast<double *>(o_out) + base * tc.dimo_out;
        const int total = nvalid * tc.dimo_out;
        int e = tid / tc.dimo_out, k = tid - e * tc.dimo_out;
        const int e_inc = 256 / tc.dimo_out, k_inc = 256 - e_inc * tc.dimo_out;
#pragma unroll 4
        for (int f = tid; f < total; f += 256) {
            const int sl = col_slot[k];
            const int src = col_off[k] + (sl == 255 ? 0 : TRIM_BLOCK * pick[e][sl]);
            oo[f] = (double)t_o[e * tc.dimo_in + src];
            e += e_inc; k += k_inc;
            if (k >= tc.dimo_out) { k -= tc.dimo_out; e++; }
        }
    } else {
        float *oo = reinterpret_cast<float *>(o_out) + base * tc.dimo_out;
        const float *po = o + base * tc.dimo_in;
        const int total = nvalid * tc.dimo_out;
        int e = tid / tc.dimo_out, k = tid - e * tc.dimo_out;
        const int e_inc = 256 / tc.dimo_out, k_inc = 256 - e_inc * tc.dimo_out;
#pragma unroll 4
        for (int f = tid; f < total; f += 256) {
            oo[f] = po[e * tc.dimo_in + k];
            e += e_inc; k += k_inc;
            if (k >= tc.dimo_out) { k -= tc.dimo_out; e++; }
        }
    }
    {
        float *og = g_out + base * tc.dimg_out, *oa = ag_out + base * tc.dimg_out;
#pragma unroll
        for (int u = 0; u < TRIM_GU; u++) {
            const int f = tid + u * 256;
            if (f < gtotal) { og[f] = vg[u]; oa[f] = va[u]; }
        }
        for (int f = tid + TRIM_GU * 256; f < gtotal; f += 256) {     // larger goal matrices than 4 x 4
            const int e = f / tc.dimg_out, k = f - e * tc.dimg_out;
            const int src = e * tc.dimg_in + tc.sel[k];
            og[f] = pg[src];
            oa[f] = pa[src];
        }
    }
}

// ------------------------------ compute_reward ------------------------------
// fetch_env.py:141-143 on float rows, one row per lane; the block's rows are
// staged through LDS with coalesced loads (row stride padded odd).
// NumPy's float add-reduction over a contiguous axis is a PAIRWISE sum (numpy/_core/src/umath/loops_utils.h.src,
// third-party; pinned by tests/golden/compute_reward_order.npz recorded from NumPy): below 8 elements a plain
// loop; up to 128 elements eight strided accumulators combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) and the
// n % 8 tail added one by one; above 128 the block is split at n/2 rounded down to a multiple of 8.  fetch_env.py:141
// sums achieved_goal * goal that way, in the arrays' own precision, so for rows that are not ternary the ORDER
// decides whether d == c.  Products and sums are rounded separately (FP_CONTRACT off).
template <typename F>
__device__ __forceinline__ F cr_block_sum(const F *p, int n)          // n <= 128; p = the products, already rounded
{
    if (n < 8) {
        F res = (F)0;
        for (int i = 0; i < n; i++) res = res + p[i];
        return res;
    }
    F r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = p[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = r[j] + p[i + j];
    }
    F res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + p[i];
    return res;
}

template <typename F>
__device__ __forceinline__ F cr_half_sum(const F *p, int n)           // n <= 256: at most one more split
{
    if (n <= 128) return cr_block_sum(p, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return cr_block_sum(p, n2) + cr_block_sum(p + n2, n - n2);
}

template <typename F>
__device__ __forceinline__ F cr_row_sum(const F *p, int n)            // n <= 256
{
    if (n <= 128) return cr_block_sum(p, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return cr_half_sum(p, n2) + cr_half_sum(p + n2, n - n2);
}

constexpr int CR_ROWS_MAX = 128;
template <typename F> struct CrVec;                     // the 16-byte vector of F and the number of requests in flight
template <> struct CrVec<float> { using V = float4; static constexpr int N = 4, U = 9; };
template <> struct CrVec<double> { using V = double2; static constexpr int N = 2, U = 8; };
__device__ __forceinline__ float cr_elem(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ double cr_elem(const double2 &v, int j) { return j == 0 ? v.x : v.y; }

// One block = blockDim.x rows, one row per lane in the end.  The tile that goes through LDS is the row-wise PRODUCT
// achieved_goal * goal (rounded on its own, as NumPy's temporary is) -- not the two operand tiles: half the LDS bytes
// and writes, twice the resident blocks -- plus per row the count of non-zero goal entries (an LDS atomic per vector).
// VEC: 16-byte loads (float4 / double2; len a multiple of their length, 16-byte aligned arrays), U vectors of each
// operand requested per lane before the first is consumed.  BCAST: one goal row for every row of ag
// (env.compute_reward(ag, env.goal), fetch_env.py:135-143 with NumPy broadcasting), staged once per block.
template <typename F, bool VEC, bool BCAST>
__device__ __forceinline__ void compute_reward_body(const F *__restrict__ ag, const F *__restrict__ g, int64_t n, int len, int ldp,
                                                    uint32_t *__restrict__ out)
{
    using V = typename CrVec<F>::V;
    // one goal row for all: only one operand stream -- float64: twice the vectors of it in flight (the same bytes per lane;
    // 124-132 -> 116 us for 2 M rows of 36); float32 measured no better with 18 than with 9
    constexpr int N = CrVec<F>::N, U = (BCAST && sizeof(F) == 8) ? 2 * CrVec<F>::U : CrVec<F>::U;
    const int CR_ROWS = blockDim.x;               // 128 rows per block, 64 when the rows are long (LDS)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    F *s_p = reinterpret_cast<F *>(smem);
    F *s_g = s_p + (size_t)CR_ROWS * ldp;                                       // BCAST: the goal row
    int *s_cnt = reinterpret_cast<int *>(s_p + (size_t)CR_ROWS * ldp);          // !BCAST: non-zero goal entries per row
    const int64_t row0 = (int64_t)blockIdx.x * CR_ROWS;
    const int rows = (int)((n - row0) < CR_ROWS ? (n - row0) : CR_ROWS);
    const F *pa = ag + row0 * len, *pg = BCAST ? g : g + row0 * len;
    if (BCAST) {
        for (int k = threadIdx.x; k < len; k += CR_ROWS) s_g[k] = pg[k];
    } else {
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    if (VEC) {
        // a vector never straddles a row; (row, col) advance incrementally
        const int qpr = len / N, total = rows * qpr;
        int r = threadIdx.x / qpr, cq = threadIdx.x - r * qpr;
        const int r_inc = CR_ROWS / qpr, c_inc = CR_ROWS - r_inc * qpr;
        const V *a4 = reinterpret_cast<const V *>(pa), *g4 = reinterpret_cast<const V *>(pg);
        // with one pair of vectors per trip the tile took rows * len / (N * CR_ROWS) serial memory round trips
        for (int q0 = threadIdx.x; q0 < total; q0 += CR_ROWS * U) {
            V va[U], vg[BCAST ? 1 : U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * CR_ROWS;
                if (q < total) {
                    va[u] = a4[q];
                    if constexpr (!BCAST) vg[u] = g4[q];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (q0 + u * CR_ROWS < total) {
                    F *dp = s_p + r * ldp + N * cq;
                    int nz = 0;
                    // the goal row sits 16-byte aligned in LDS (its offset is a multiple of 64 elements): one vector read
                    const V gv = BCAST ? *reinterpret_cast<const V *>(s_g + N * cq) : vg[BCAST ? 0 : u];
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        const F gj = cr_elem(gv, j);
                        dp[j] = cr_elem(va[u], j) * gj;
                        nz += (gj != (F)0) ? 1 : 0;
                    }
                    if (!BCAST && nz) atomicAdd(&s_cnt[r], nz);
                }
                r += r_inc; cq += c_inc;
                if (cq >= qpr) { cq -= qpr; r++; }
            }
        }
    } else {
        const int total = rows * len;
        for (int f = threadIdx.x; f < total; f += CR_ROWS) {
            const int r = f / len, k = f - r * len;
            const F gk = BCAST ? s_g[k] : pg[f];
            s_p[r * ldp + k] = pa[f] * gk;
            if (!BCAST && gk != (F)0) atomicAdd(&s_cnt[r], 1);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < rows) {
        int cnt = 0;
        if (BCAST) {
            for (int k = 0; k < len; k++) cnt += (s_g[k] != (F)0) ? 1 : 0;
        } else {
            cnt = s_cnt[threadIdx.x];
        }
        const F d = cr_row_sum<F>(s_p + threadIdx.x * ldp, len);    // NumPy's summation order for this precision
        out[row0 + threadIdx.x] = (d != (F)cnt) ? RBITS_FAIL : RBITS_OK;
    }
}

template <bool VEC4, bool BCAST = false>
__global__ void __launch_bounds__(CR_ROWS_MAX)
k_compute_reward_f32(const float *__restrict__ ag, const float *__restrict__ g, int64_t n, int len, int ldp,
                     uint32_t *__restrict__ out)
{
    compute_reward_body<float, VEC4, BCAST>(ag, g, n, len, ldp, out);
}

// The same on float64 rows (env.step's own call: achieved_goal float64, goal int64 / float64 -> a float64 sum,
// fetch_env.py:141-143).
template <bool VEC2, bool BCAST = false>
__global__ void __launch_bounds__(CR_ROWS_MAX)
k_compute_reward_f64(const double *__restrict__ ag, const double *__restrict__ g, int64_t n, int len, int ldp,
                     uint32_t *__restrict__ out)
{
    compute_reward_body<double, VEC2, BCAST>(ag, g, n, len, ldp, out);
}

__global__ void __launch_bounds__(256)
k_compute_reward_packed(const uint32_t *__restrict__ now, const uint32_t *__restrict__ ever,
                        uint32_t want_touch, uint32_t want_never, int64_t n, uint32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t bad = (want_touch & ~now[i]) | (want_never & ever[i]);
        out[i] = bad ? RBITS_FAIL : RBITS_OK;
    }
}

// ------------------------------ success count -------------------------------
// rollout.py:163-167 as its own launch: <= 256 blocks of 1024 lanes, 16-byte
// loads, block reduction, one atomic per block.
__global__ void __launch_bounds__(1024)
k_success_count(const uint32_t *__restrict__ ever, int64_t n, CountArgs ca)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0;
    for (int64_t v = gtid; v < nvec; v += stride) {
        const uint4 e = reinterpret_cast<const uint4 *>(ever)[v];
        c += (e.x >> 31) + (e.y >> 31) + (e.z >> 31) + (e.w >> 31);
    }
    if (gtid < n - (nvec << 2)) c += ever[(nvec << 2) + gtid] >> 31;
    block_count_add(c, ca);
}

// ------------------------------ discounted return ---------------------------
// policy_gradient/rollout.py:255-258: G[t_] starts as r[t_] and adds gamma**(t-t_) * r[t] for
// t = t_+1.. in increasing t (float64, products and sums rounded separately, so the result matches
// the reference's NumPy bit for bit).  gamma**k (C pow, as Python computes it) comes in through
// the kernel arguments.  The loop nest is the reference's own -- OUTER over the step t that brings a
// new reward, INNER over the returns t_ < t it is added to -- so the additions of one outer step are
// independent chains (an f64 result is ~60 cycles away on gfx950, profiles/r02_valu_rates.log).
// T = 50 / 25 / 10 run in registers, one env per lane (k_discounted_return_lane below; rounds 1 and 2 had
// one and two lanes per env with all 2 450 operations -- git history); every other horizon here:

// any T up to DR_TMAX: returns accumulate in the LDS tile (row of the lane), four at a time so
// that independent chains overlap; the gamma**k table travels in the kernel arguments as well
// (no allocation, no copy, nothing to wait for: the call only enqueues a launch).
constexpr int DR_TMAX = 319;                  // 64 x (T|1) x 8 B <= 160 KB of LDS
struct PowTableBig { double v[DR_TMAX + 1]; };

__global__ void __launch_bounds__(64)
k_discounted_return(const float *__restrict__ reward_tb, int64_t n, int64_t ld, int T, int R, PowTableBig pw,
                    double *__restrict__ G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tile = reinterpret_cast<double *>(smem);   // [64][R]
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    if (lane < nvalid) {
        double *my = tile + lane * R;
        const float *p = reward_tb + base + lane;
        // all rewards into the lane's LDS row first, eight requests in flight at a time (one request
        // per trip of the accumulation loop would be T serial memory round trips)
        for (int t0 = 0; t0 < T; t0 += 8) {
            float r8[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (t0 + u < T) r8[u] = p[(int64_t)(t0 + u) * ld];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (t0 + u < T) my[t0 + u] = (double)r8[u];
        }
        for (int t = 0; t < T; t++) {
            const double rt = my[t];
            int t_ = 0;
            for (; t_ + 4 <= t; t_ += 4) {
                const double p0 = pw.v[t - t_] * rt, p1 = pw.v[t - t_ - 1] * rt, p2 = pw.v[t - t_ - 2] * rt,
                             p3 = pw.v[t - t_ - 3] * rt;
                const double m0 = my[t_], m1 = my[t_ + 1], m2 = my[t_ + 2], m3 = my[t_ + 3];
                __builtin_amdgcn_sched_barrier(0);
                my[t_] = m0 + p0; my[t_ + 1] = m1 + p1; my[t_ + 2] = m2 + p2; my[t_ + 3] = m3 + p3;
            }
            for (; t_ < t; t_++) my[t_] = my[t_] + pw.v[t - t_] * rt;
        }
    }
    __syncthreads();
    double *o = G + base * (int64_t)T;
    const int total = nvalid * T;
    int e = 0, rem = lane;
    while (rem >= T) { rem -= T; e++; }
    for (int f = lane; f < total; f += 64) {
        o[f] = tile[e * R + rem];
        rem += 64;
        while (rem >= T) { rem -= T; e++; }
    }
}

// ---- lane = env return kernels (T = 50 / 25 / 10) --------------------------------------------------------
// EXACT: the reference's float64 result bit for bit.  Every reward this env produces is -1.0f or -0.0f
// (fetch_env.py:141-143), and for those the reference's product-then-sum is exactly a conditional
// subtraction of gamma**k (returns_from_okbits above): a wave whose 64 x T rewards all lie in {-1.0f, -0.0f}
// takes that path -- half the float64 operations of the general one and no per-lane reward operand --, any
// other wave re-reads its rewards (they are L2-resident by then) and runs the general unfused product + sum,
// in the reference's order.  The choice is per wave and changes no result bit.
// SCAN: SURVEY 8(f3)'s O(T) reverse scan G[t] = fma(gamma, G[t+1], r[t]) -- NOT the reference's summation
// order: |G_scan - G_ref| <= 4 T 2^-53 sum_k gamma^k |r[t+k]| (include/bpgym.h states the bound).
enum { DRM_EXACT = 0, DRM_SCAN = 1 };
constexpr int DR_LANE_WPB = 4;              // waves per block (each wave: 64 envs, a private 16-row LDS tile)

// RT: any horizon T_rt <= TT through the TT-accumulator form -- rows t >= T_rt are neither read nor do they contribute
// (every guard below is wave-uniform and compiled out when !RT), only the first T_rt entries of a row leave.
template <int TT, int MODE, int WPB, int TR = G_TILE_ROWS, bool RT = false>
__global__ void __launch_bounds__(64 * WPB)
k_discounted_return_lane(PowTable pw, const float *__restrict__ reward_tb, int64_t n, int64_t ld, double gamma,
                         double *__restrict__ G, int allow_plain, int T_rt)
{
    const int T = RT ? T_rt : TT;
    auto row = [&](int t) { return !RT || t < T; };      // row t of the stream exists
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WPB][TR * TT * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * WPB + wave) * 64;
    if (base >= n) return;                    // whole wave out of range (no block-wide barrier below)
    const bool live = base + lane < n;
    // one 32-bit byte offset per lane (the host guarantees 4 * ld * T < 2^32 per row step: ld < 2^28)
    const int64_t e = live ? base + lane : n - 1;
    const uint32_t *rp = reinterpret_cast<const uint32_t *>(reward_tb) + e;
    uint32_t r[TT];
    static_for<0, TT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        r[t] = row(t) ? __builtin_nontemporal_load(rp + (int64_t)t * ld) : RBITS_OK;          // read once
    });
    double acc[TT];
    static_assert(!(RT && MODE == DRM_SCAN), "the scan keeps its LDS-tile form at other horizons");
    if constexpr (MODE == DRM_SCAN) {
        acc[TT - 1] = (double)__uint_as_float(r[TT - 1]);
        static_for<1, TT>([&](auto tc) {
            constexpr int t = TT - 1 - decltype(tc)::value;
            acc[t] = __fma_rn(gamma, acc[t + 1], (double)__uint_as_float(r[t]));
        });
    } else {
        // every reward of this env is -1.0f or -0.0f -- and every gamma**k is finite and >= 0 (allow_plain, see
        // gamma_is_subtractable()): only then is the conditional subtraction the reference's product + sum bit for bit
        bool plain = allow_plain != 0;
        uint64_t ok = 0ull;
        static_for<0, TT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const bool isok = r[t] == RBITS_OK;
            plain &= isok | (r[t] == RBITS_FAIL);
            ok |= (uint64_t)isok << t;
        });
        if (__all(plain)) {
            returns_from_okbits<TT>(ok, acc);                // reads gamma**k from kernarg offset 0 (= pw)
        } else {
            // general rewards: the reference's loop nest (OUTER over the step t that brings a reward, INNER over
            // the returns t_ < t), product and sum rounded separately.  Same two passes over the lag as above
            // (per accumulator the additions still come in increasing t); the rewards are read again, step by
            // step and a few steps ahead (they are L2-resident by now), so that neither T reward registers nor
            // the whole gamma**k table stay alive beside the T accumulators.
            constexpr int KA = TT / 2, AHEAD = 4, GW = 6;     // GW: products in flight (their temporaries are registers)
            {
                KernargDouble *pwk = kernarg_pow();
                double ck[KA + 1];
                static_for<1, KA + 1>([&](auto kc) { ck[decltype(kc)::value] = pwk[decltype(kc)::value]; });
                static_for<1, KA + 1>([&](auto kc) { pin_sgpr(ck[decltype(kc)::value]); });
                // a running pointer, advanced one row per read: with T independent row addresses the compiler
                // computes all of them up front (2 VGPRs each)
                const uint32_t *rq = rp;
                asm volatile("" : "+v"(rq));
                float ahead[AHEAD];
                static_for<0, AHEAD>([&](auto uc) {
                    if (row(decltype(uc)::value)) {
                        ahead[decltype(uc)::value] = __uint_as_float(*rq);
                        if constexpr (decltype(uc)::value + 1 < TT) { if (row(decltype(uc)::value + 1)) rq += ld; }
                    }
                    asm volatile("" : "+v"(rq));
                });
                static_for<0, TT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if (!row(t)) return;
                    const double rt = (double)ahead[t % AHEAD];
                    if constexpr (t + AHEAD < TT) {
                        if (row(t + AHEAD)) {
                            ahead[t % AHEAD] = __uint_as_float(*rq);
                            if constexpr (t + AHEAD + 1 < TT) { if (row(t + AHEAD + 1)) rq += ld; }
                        }
                        asm volatile("" : "+v"(rq));
                    }
                    acc[t] = rt;                                            // G[t] starts as r[t]
                    constexpr int lo = t > KA ? t - KA : 0;
                    static_for<0, (t - lo + GW - 1) / GW>([&](auto gc) {       // GW products, then their GW sums
                        constexpr int g0 = lo + decltype(gc)::value * GW;
                        static_for<g0, (g0 + GW < t ? g0 + GW : t)>([&](auto uc) {
                            constexpr int t_ = decltype(uc)::value;
                            acc[t_] = acc[t_] + ck[t - t_] * rt;
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
            if constexpr (KA + 1 < TT) {
                KernargDouble *pwk = kernarg_pow();
                double ck[TT];
                static_for<KA + 1, TT>([&](auto kc) { ck[decltype(kc)::value] = pwk[decltype(kc)::value]; });
                static_for<KA + 1, TT>([&](auto kc) { pin_sgpr(ck[decltype(kc)::value]); });
                const uint32_t *rq = rp + (int64_t)(KA + 1) * ld;
                asm volatile("" : "+v"(rq));
                float ahead[AHEAD];
                static_for<0, AHEAD>([&](auto uc) {
                    if (row(KA + 1 + decltype(uc)::value)) {
                        ahead[decltype(uc)::value] = __uint_as_float(*rq);
                        if constexpr (KA + 1 + decltype(uc)::value + 1 < TT) { if (row(KA + 1 + decltype(uc)::value + 1)) rq += ld; }
                    }
                    asm volatile("" : "+v"(rq));
                });
                static_for<KA + 1, TT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if (!row(t)) return;
                    const double rt = (double)ahead[(t - KA - 1) % AHEAD];
                    if constexpr (t + AHEAD < TT) {
                        if (row(t + AHEAD)) {
                            ahead[(t - KA - 1) % AHEAD] = __uint_as_float(*rq);
                            if constexpr (t + AHEAD + 1 < TT) { if (row(t + AHEAD + 1)) rq += ld; }
                        }
                        asm volatile("" : "+v"(rq));
                    }
                    static_for<0, (t - KA + GW - 1) / GW>([&](auto gc) {
                        constexpr int g0 = decltype(gc)::value * GW;
                        static_for<g0, (g0 + GW < t - KA ? g0 + GW : t - KA)>([&](auto uc) {
                            constexpr int t_ = decltype(uc)::value;
                            acc[t_] = acc[t_] + ck[t - t_] * rt;
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        }
    }
    if constexpr (RT) wave_store_returns_rt<TT, TR>(acc, reinterpret_cast<double *>(lds_raw[wave]), G, base, n, lane, T);
    else wave_store_returns<TT, TR>(acc, reinterpret_cast<double *>(lds_raw[wave]), G, base, n, lane);
}

// G rows from the step-ok words of a rollout launch (k_rollout<..., OKB>, k_episode): same arithmetic as above,
// 8 bytes read and 8 T bytes written per env.
template <int TT, int WPB, int TR = G_TILE_ROWS>
__global__ void __launch_bounds__(64 * WPB)
k_returns_from_okbits_lane(PowTable pw_at_kernarg_offset_0, const uint64_t *__restrict__ okbits, double *__restrict__ G,
                           int64_t n)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WPB][TR * TT * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // grid-stride over 64-env tiles (one tile per wave when the grid covers the batch)
    for (int64_t base = ((int64_t)blockIdx.x * WPB + wave) * 64; base < n; base += (int64_t)gridDim.x * WPB * 64) {
        const int64_t e = base + lane < n ? base + lane : n - 1;
        const uint64_t ok = okbits[e];
        double acc[TT];
        returns_from_okbits<TT>(ok, acc);
        wave_store_returns<TT, TR>(acc, reinterpret_cast<double *>(lds_raw[wave]), G, base, n, lane);
        wave_lds_sync();                               // the tile is written again by the next trip
    }
}

// Any horizon T <= TT through the register-resident form: steps t >= T count as "ok" (nothing is subtracted for them, so
// the first T accumulators are bit for bit those of a T-step rollout -- only steps t < T ever contribute to them), the
// lag passes of steps that no lane needs are skipped wave-uniformly, and only the first T entries of a row leave.
// (The reference's tasks all use T = 50, which has its own instantiation; this one takes the cliff out of every other
// horizon: T = 40 at 1 M envs took 346 us through the LDS-tile kernel this one replaces, 71 us now, profiles/r04_survey_secondary.log.)
template <int TT, int WPB, int TR = G_TILE_ROWS>
__global__ void __launch_bounds__(64 * WPB)
k_returns_from_okbits_lane_rt(PowTable pw_at_kernarg_offset_0, const uint64_t *__restrict__ okbits, double *__restrict__ G,
                              int64_t n, int T)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WPB][TR * TT * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t beyond = T < 64 ? ~0ull << T : 0ull;
    for (int64_t base = ((int64_t)blockIdx.x * WPB + wave) * 64; base < n; base += (int64_t)gridDim.x * WPB * 64) {
        const int64_t e = base + lane < n ? base + lane : n - 1;
        const uint64_t ok = okbits[e] | beyond;
        double acc[TT];
        returns_from_okbits<TT>(ok, acc);
        wave_store_returns_rt<TT, TR>(acc, reinterpret_cast<double *>(lds_raw[wave]), G, base, n, lane, T);
        wave_lds_sync();                               // the tile is written again by the next trip
    }
}

// any T <= DR_TMAX: O(T) reverse scan, rows in an LDS tile
__global__ void __launch_bounds__(64)
k_discounted_return_scan(const float *__restrict__ reward_tb, int64_t n, int64_t ld, int T, int R, double gamma,
                         double *__restrict__ G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tile = reinterpret_cast<double *>(smem);   // [64][R]
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    if (lane < nvalid) {
        double *my = tile + lane * R;
        const float *p = reward_tb + base + lane;
        double g = 0.0;
        for (int t1 = T; t1 > 0; t1 -= 8) {            // steps t1-8 .. t1-1, eight requests in flight
            float r8[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (t1 - 1 - u >= 0) r8[u] = p[(int64_t)(t1 - 1 - u) * ld];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t1 - 1 - u;
                if (t >= 0) {
                    g = (t == T - 1) ? (double)r8[u] : __fma_rn(gamma, g, (double)r8[u]);
                    my[t] = g;
                }
            }
        }
    }
    __syncthreads();
    double *o = G + base * (int64_t)T;
    const int total = nvalid * T;
    int e = 0, rem = lane;
    while (rem >= T) { rem -= T; e++; }
    for (int f = lane; f < total; f += 64) {
        o[f] = tile[e * R + rem];
        rem += 64;
        while (rem >= T) { rem -= T; e++; }
    }
}

// np.isnan(o_new).any() of the rollout worker (rollout.py:139-142) without a host round trip:
// ORs 1 into *flag when any of rows x cols float32 values (row stride ld) is a NaN.
// FLAT: the rows are one contiguous, 16-byte aligned run -> float4 loads, four in flight per lane.
template <bool FLAT>
__global__ void __launch_bounds__(256)
k_nan_flag(const float *__restrict__ x, int64_t rows, int cols, int64_t ld, uint32_t *__restrict__ flag)
{
    const int64_t total = rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (FLAT) {
        const int64_t nq = total >> 2;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int64_t q0 = gtid; q0 < nq; q0 += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (q0 + u * stride < nq) v[u] = x4[q0 + u * stride];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (q0 + u * stride < nq)
                    bad |= (v[u].x != v[u].x) | (v[u].y != v[u].y) | (v[u].z != v[u].z) | (v[u].w != v[u].w);
        }
        for (int64_t f = (nq << 2) + gtid; f < total; f += stride) bad |= (x[f] != x[f]);
    } else {
        for (int64_t f = gtid; f < total; f += stride) {
            const int64_t r = f / cols;
            const float v = x[r * ld + (f - r * cols)];
            bad |= (v != v);
        }
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ void __launch_bounds__(256)
k_fill_goal(float *__restrict__ out, int64_t total, int dimg, int64_t ld, PairTable tab)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += stride) {
        const int64_t e = f / dimg;
        const int k = (int)(f - e * dimg);
        st_1<OUT_NT>(&out[e * ld + k], tab.goal[k]);
    }
}

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
// cap = 2048: 256 CUs x 8 blocks, grid-stride the rest.  Pure read streams run faster as one short-lived
// block per tile (blocks beyond the resident set start in address order as earlier ones retire: the
// box's read ceiling is 6.2-6.7 TB/s that way against 5.2-5.5 for a grid-stride loop,
// profiles/r02_stream_ceiling.log) -- those pass GRID_UNCAPPED; kernels with scattered small stores
// (k_fill_goal: 2x slower uncapped) keep the cap.
constexpr int64_t GRID_UNCAPPED = (int64_t)1 << 22;
int grid_for(int64_t work_items, int block, int64_t cap = 2048)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

StepConst step_const(const bpg_batch *b)
{
    StepConst c;
    c.valid = b->info.valid_pairs;
    c.want_touch = b->info.want_touch;
    c.want_never = b->info.want_never;
    c.max_steps = (uint32_t)b->max_steps;
    // Variation: matrix side 6, num_objs 4..6 per episode (fetch_env.py:649) -> pairs with an index >= n are
    // never demoted (:154-157).  Every other variant: num_objs == side, nothing is sticky.
    c.sticky_on = b->info.is_variation ? 1u : 0u;
    for (int n = 0; n < BPG_MAX_OBJS; n++) {
        uint32_t demoted = 0;
        for (int i = 0; i < n && i < b->info.side; i++)
            for (int j = i; j < n && j < b->info.side; j++) demoted |= 1u << host_pair_index(i, j);
        c.sticky_by_n[n] = b->info.valid_pairs & ~demoted;
    }
    return c;
}

CountArgs count_args(const bpg_batch *b, unsigned long long *count, unsigned long long *zero_next)
{
    CountArgs ca;
    ca.count = count;
    ca.zero_next = zero_next;
    ca.shards = b->d_count_shards;
    return ca;
}

// gamma ** k as Python's float pow (C pow), k < DR_TMAX_REG
PowTable pow_table(double gamma, int T)
{
    PowTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (int k = 0; k < T && k < DR_TMAX_REG; k++) tab.v[k] = std::pow(gamma, (double)k);
    return tab;
}

SynthConst synth_const(const bpg_batch *b)
{
    SynthConst s;
    s.seed0 = b->seed + 1000ull * (uint64_t)b->env_base;        // rollout.py:209-210 on the global env index
    for (int n = 0; n < 8; n++) {
        uint32_t valid = 0, forced = 0;
        const int m = n > BPG_MAX_OBJS ? BPG_MAX_OBJS : n;
        for (int i = 0; i < m; i++)
            for (int j = i + 1; j < m; j++) valid |= 1u << host_pair_index(i, j);
        for (int k = 2; k < m; k++) forced |= 1u << host_pair_index(1, k);
        s.valid_by_n[n] = valid;
        s.forced_by_n[n] = forced;
    }
    return s;
}

ResetConst reset_const(const bpg_batch *b, int enabled, const int32_t *d_num_grey)
{
    ResetConst rc;
    const SynthConst s = synth_const(b);
    rc.enabled = enabled;
    rc.is_variation = b->info.is_variation;
    rc.base_nobj = (uint32_t)b->info.num_objs;
    rc.seed0 = s.seed0;
    rc.reset_counter = b->d_reset_counter;
    rc.num_grey = d_num_grey;
    return rc;
}

int fat_grid(int64_t work_items)
{
    int64_t g = (work_items + 1023) / 1024;
    if (g < 1) g = 1;
    if (g > 256) g = 256;        // one 1024-lane block per CU; <= 256 counter atomics per launch
    return (int)g;
}

// Launch shape of the fused rollout, chosen from the batch size (tools/tune_round3.hip mode A,
// profiles/r03_tune_mid_shapes.log; round 2: profiles/r02_tune_geometry.log; MI355X, T = 50, masks read cold
// from HBM).  The kernel is a pure stream with T independent requests per lane, so what matters is (a) enough
// resident waves on every CU to cover the HBM latency and (b), when there are not, that no wave waits for a
// store round trip before it can consume a load (the pipelined loop).  rocprofv3 means per launch
// (profiles/r03_pmc/kernel_means.json), round 1's plain 4/lane x 1024 shape in parentheses:
//     65 536      9.9 us  (49)   1/lane, 256-lane blocks, pipelined, 10 rows in flight
//     131 072    14.5 us  (45)   1/lane, 256, pipelined, 5 rows in flight
//     262 144    23.5 us  (52)   2/lane, 256, plain loop
//     524 288    43.8 us  (58)   2/lane, 1024, plain
//     1 048 576  82-91 us (89)   2/lane, 1024, plain (shifted rows, bpg_row_stride; box-dependent)
//     4 194 304  351 us   (372)  2/lane, 256, plain, when the rows are a power of two apart (1024 lanes: 468 us)
// 8-byte accesses (512 B per wave instruction) stream as fast as 16-byte ones here, so there is no
// 4-per-lane instantiation.  One returning atomic per block on a sharded counter (block_count_add).
constexpr int64_t ROLLOUT_UNROLL5_FROM = 81920, ROLLOUT_EPL2_FROM = 229376;
struct RolloutGeometry { int epl, block, grid, unroll; bool pipe; };

RolloutGeometry rollout_geometry(int64_t n, int64_t ld_masks, int64_t ld_out, const void *masks, const void *reward,
                                 const void *flags, const void *ep_return = nullptr)
{
    // 2 envs per lane need an even batch, even row strides and 8-byte aligned mask / reward streams
    // (2-byte flags)
    const bool ok2 = n % 2 == 0 && ld_masks % 2 == 0 && ld_out % 2 == 0 && ((uintptr_t)masks & 7u) == 0 &&
                     ((uintptr_t)reward & 7u) == 0 && ((uintptr_t)flags & 1u) == 0 && ((uintptr_t)ep_return & 7u) == 0;
    RolloutGeometry g;
    g.unroll = 5;
    if (ok2 && n >= 131072 && !masks) {
        // contacts generated in-kernel: write-only and co-limited by the generator's arithmetic.  One
        // short-lived 256-lane block per 512 envs (no grid cap: blocks start in address order as
        // earlier ones retire), 10 steps per trip: 51 vs 53-58 us at 1 M envs, 178 vs 190-220 us at
        // 4 M (profiles/r02_tune_geometry_synth.log)
        g.epl = 2;
        g.block = 256;
        g.pipe = false;
        g.unroll = 10;
        g.grid = (int)((n / 2 + 255) / 256);
        return g;
    }
    if (ok2 && n >= ROLLOUT_EPL2_FROM) {
        // 2 envs per lane, plain loop.  From 524 288 envs on in 1024-lane blocks (one per CU, 16 waves cover
        // the round trips); below that -- and from 4 M envs on when the rows are a power of two apart -- in
        // 256-lane blocks.  (Round 2 ran a software-pipelined 2-per-lane loop here whose ping-pong buffers
        // spilled to scratch under the 1024-lane register bound it was compiled with; with an honest bound it
        // needs 200 VGPRs and is no faster than the plain loop: profiles/r03_tune_mid_shapes.log.)
        g.epl = 2;
        const bool aliased = ld_masks % 4096 == 0 || ld_out % 4096 == 0;
        const bool fat = n >= 524288 && (n < 4194304 || !aliased);
        g.block = fat ? 1024 : 256;
        g.pipe = false;
    } else {
        // 1 env per lane, software-pipelined requests (latency-bound sizes): 10 rows in flight per lane below
        // ~100 K envs, 5 above (profiles/r03_tune_mid_shapes.log)
        g.epl = 1;
        g.block = 256;
        g.pipe = true;
        g.unroll = n < ROLLOUT_UNROLL5_FROM ? 10 : 5;
    }
    const int64_t lanes = n / g.epl;
    int64_t grid = (lanes + g.block - 1) / g.block;
    if (grid < 1) grid = 1;
    const int64_t cap = 256 * (1024 / g.block);           // at most 1024 lanes' worth of blocks per CU
    g.grid = (int)(grid > cap ? cap : grid);
    return g;
}

PairTable pair_table(const bpg_variant_info &info)
{
    PairTable t;
    std::memset(&t, 0, sizeof(t));
    const int side = info.side;
    for (int f = 0; f < side * side; f++) {
        t.flat[f] = (uint8_t)host_pair_index(f / side, f % side);
        t.goal[f] = (float)info.goal[f];
    }
    for (int q = 0; q < 9; q++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) {
            const int f = 4 * q + k;
            if (f < side * side) w |= (uint32_t)t.flat[f] << (5 * k);
        }
        t.quad[q] = w;
    }
    return t;
}

template <typename F>
int dispatch_side(int side, F &&f)
{
    switch (side) {
    case 3: return f(std::integral_constant<int, 3>());
    case 4: return f(std::integral_constant<int, 4>());
    case 5: return f(std::integral_constant<int, 5>());
    case 6: return f(std::integral_constant<int, 6>());
    }
    return fail(BPG_ERR_ARG, "unsupported matrix side");
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BPG_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return BPG_OK;
}

}  // namespace bpgk
using namespace bpgk;

// LDS of a block: the product tile + either the goal row (bcast) or the per-row counts
template <typename F>
static size_t compute_reward_lds(int rows, int ldp, bool bcast)
{
    return (size_t)rows * ldp * sizeof(F) + (bcast ? (size_t)ldp * sizeof(F) : (size_t)rows * sizeof(int));
}

template <typename F, typename K>
static int compute_reward_launch(K kern_vec_b, K kern_vec, K kern_b, K kern, const F *d_ag, const F *d_g, bool bcast, int64_t n,
                                 int len, float *d_out, int device_id, void *stream, const char *what)
{
    if (!d_ag || !d_g || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (n < 0 || len < 1 || len > 256) return fail(BPG_ERR_ARG, "len must be in 1..256");
    if (((uintptr_t)d_ag % sizeof(F)) || ((uintptr_t)d_g % sizeof(F))) return fail(BPG_ERR_ARG, "rows must be aligned to their element size");
    if (n == 0) return BPG_OK;
    DeviceGuard guard(device_id);
    const int ldp = len | 1;
    // 128 rows per block while two blocks fit a CU's 160 KB of LDS, 64 rows beyond (float64 rows longer than 79)
    const int CR_ROWS = compute_reward_lds<F>(CR_ROWS_MAX, ldp, bcast) <= 80 * 1024 ? CR_ROWS_MAX : 64;
    const size_t lds = compute_reward_lds<F>(CR_ROWS, ldp, bcast);
    constexpr int N = (int)(16 / sizeof(F));
    const bool vec = (len % N == 0) && (len / N <= CR_ROWS) && aligned16(d_ag) && (bcast || aligned16(d_g));
    K fn = vec ? (bcast ? kern_vec_b : kern_vec) : (bcast ? kern_b : kern);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((n + CR_ROWS - 1) / CR_ROWS)), block(CR_ROWS);
    hipLaunchKernelGGL(fn, grid, block, lds, (hipStream_t)stream, d_ag, d_g, n, len, ldp, reinterpret_cast<uint32_t *>(d_out));
    return check_launch(what);
}

// versioned argument structs (include/bpgym.h): copies what the caller passed into a zeroed struct of this library's
// layout; bytes this library does not know must be zero
template <typename A>
static int take_args(A &local, const A *args, size_t min_size, const char *what)
{
    const size_t caller_size = args->struct_size;
    if (caller_size < min_size || caller_size > 4096 || caller_size % 8 != 0)
        return fail(BPG_ERR_ARG, std::string(what) + ".struct_size is not set (use the _INIT macro; this library is bpg_version() 104)");
    std::memset(&local, 0, sizeof(local));
    std::memcpy(&local, args, caller_size < sizeof(local) ? caller_size : sizeof(local));
    for (size_t i = sizeof(local); i < caller_size; i++)
        if (reinterpret_cast<const unsigned char *>(args)[i] != 0)
            return fail(BPG_ERR_UNSUPPORTED, std::string(what) + " carries fields this library (bpg_version() 104) does not know");
    return BPG_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char *bpg_last_error(void) { return g_err.c_str(); }
int bpg_version(void) { return 104; }    // 100 + round

int bpg_variant_from_id(const char *env_id)
{
    if (!env_id) return -1;
    for (int v = 0; v < BPG_NUM_VARIANTS; v++)
        if (std::strcmp(env_id, kVariants[v].id) == 0) return v;
    return -1;
}

int bpg_get_variant_info(int variant, bpg_variant_info *out)
{
    if (variant < 0 || variant >= BPG_NUM_VARIANTS || !out) return fail(BPG_ERR_ARG, "bad variant");
    fill_variant_info(variant, out);
    return BPG_OK;
}

int bpg_geom2objid(const char *name)
{
    // fetch_env.py:106-117
    if (name) {
        if (std::strstr(name, "finger")) return 0;
        if (std::strcmp(name, "table") == 0) return 1;
        if (std::strstr(name, "object")) {
            const char *p = name + (std::strlen(name) < 6 ? std::strlen(name) : 6);
            return std::atoi(p) + 2;
        }
    }
    return -1;
}

int bpg_pair_index(int i, int j)
{
    if (i < 0 || j < 0 || i >= BPG_MAX_OBJS || j >= BPG_MAX_OBJS) return -1;
    return host_pair_index(i, j);
}

int bpg_create(bpg_batch **out, int variant, int64_t n_envs, int device_id, int max_episode_steps,
               uint64_t seed, int64_t env_id_base)
{
    if (!out) return fail(BPG_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (variant < 0 || variant >= BPG_NUM_VARIANTS) return fail(BPG_ERR_ARG, "unknown variant");
    if (n_envs < 1) return fail(BPG_ERR_ARG, "n_envs must be >= 1");
    if (max_episode_steps < 1 || max_episode_steps > (int)ELAPSED_MAX)
        return fail(BPG_ERR_ARG, "max_episode_steps must be in 1..1023");
    DeviceGuard guard(device_id);
    if (guard.err != hipSuccess)
        return fail(BPG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    bpg_batch *b = new (std::nothrow) bpg_batch();
    if (!b) return fail(BPG_ERR_ARG, "out of host memory");
    fill_variant_info(variant, &b->info);
    b->n = n_envs;
    b->device = device_id;
    b->max_steps = max_episode_steps;
    b->info.max_episode_steps = max_episode_steps;
    b->seed = seed;
    b->env_base = env_id_base;
    b->d_reset_counter = nullptr;
    b->now = b->ever = nullptr;
    b->d_id2obj = nullptr;
    b->d_count_shards = nullptr;
    b->d_okbits = nullptr;
    std::memset(b->ngeom, 0, sizeof(b->ngeom));
    const size_t bytes = (size_t)((n_envs + 3) & ~(int64_t)3) * sizeof(uint32_t);
    hipError_t e1 = hipMalloc(&b->now, bytes), e2 = hipMalloc(&b->ever, bytes);
    hipError_t e3 = hipMalloc(&b->d_id2obj, BPG_MAX_MODELS * BPG_MAX_GEOMS);
    if (e3 == hipSuccess) e3 = hipMalloc(&b->d_reset_counter, 2 * sizeof(uint32_t));
    const size_t shard_bytes = (size_t)COUNT_SHARDS * COUNT_SHARD_STRIDE * sizeof(unsigned long long);
    if (e3 == hipSuccess) e3 = hipMalloc(&b->d_count_shards, shard_bytes);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        bpg_destroy(b);
        return fail(BPG_ERR_HIP, "hipMalloc failed for the packed state");
    }
    hipError_t e4 = hipMemset(b->d_id2obj, 0xFF, BPG_MAX_MODELS * BPG_MAX_GEOMS);
    if (e4 == hipSuccess) e4 = hipMemset(b->d_reset_counter, 0, 2 * sizeof(uint32_t));
    if (e4 == hipSuccess) e4 = hipMemset(b->d_count_shards, 0, shard_bytes);
    hipLaunchKernelGGL(k_init, dim3(grid_for(n_envs, 256)), dim3(256), 0, 0, b->now, b->ever, n_envs,
                       (uint32_t)b->info.num_objs);
    hipError_t e5 = hipDeviceSynchronize();
    if (e4 != hipSuccess || e5 != hipSuccess) {
        bpg_destroy(b);
        return fail(BPG_ERR_HIP, "state initialisation failed");
    }
    *out = b;
    return BPG_OK;
}

int bpg_destroy(bpg_batch *b)
{
    if (!b) return BPG_OK;
    DeviceGuard guard(b->device);
    if (b->now) (void)hipFree(b->now);
    if (b->ever) (void)hipFree(b->ever);
    if (b->d_id2obj) (void)hipFree(b->d_id2obj);
    if (b->d_reset_counter) (void)hipFree(b->d_reset_counter);
    if (b->d_count_shards) (void)hipFree(b->d_count_shards);
    if (b->d_okbits) (void)hipFree(b->d_okbits);
    delete b;
    return BPG_OK;
}

int64_t bpg_num_envs(const bpg_batch *b) { return b ? b->n : 0; }

int bpg_seed(bpg_batch *b, uint64_t seed)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    DeviceGuard guard(b->device);
    // Synchronous by contract: launches in flight on ANY stream (torch side streams do not order
    // against the null stream) still read the old key and stream indices.
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(b->d_reset_counter, 0, 2 * sizeof(uint32_t)));
    HIP_TRY(hipDeviceSynchronize());
    b->seed = seed;
    return BPG_OK;
}

int bpg_set_id2obj(bpg_batch *b, int model_idx, const int8_t *id2obj, int ngeom)
{
    if (!b || !id2obj) return fail(BPG_ERR_ARG, "NULL argument");
    if (model_idx < 0 || model_idx >= BPG_MAX_MODELS) return fail(BPG_ERR_ARG, "model_idx out of range");
    if (ngeom < 1 || ngeom > BPG_MAX_GEOMS) return fail(BPG_ERR_ARG, "ngeom must be in 1..64");
    int8_t tab[BPG_MAX_GEOMS];
    std::memset(tab, 0xFF, sizeof(tab));
    for (int i = 0; i < ngeom; i++) {
        if (id2obj[i] >= b->info.num_objs) return fail(BPG_ERR_ARG, "object id >= num_objs in id2obj");
        tab[i] = id2obj[i] < 0 ? (int8_t)-1 : id2obj[i];
    }
    DeviceGuard guard(b->device);
    HIP_TRY(hipDeviceSynchronize());      // synchronous by contract: no launch may still be reading the table
    HIP_TRY(hipMemcpy(b->d_id2obj + model_idx * BPG_MAX_GEOMS, tab, BPG_MAX_GEOMS, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    b->ngeom[model_idx] = ngeom;
    return BPG_OK;
}

static StaticObs static_obs(const bpg_batch *b, float *d_obs, int64_t ld, bool full, const uint8_t *d_env_mask)
{
    StaticObs so;
    std::memset(&so, 0, sizeof(so));
    // rewriting whole rows concerns every variant; the physics-free entries alone only Variation
    so.enabled = (d_obs && (full || b->info.is_variation)) ? 1 : 0;
    so.obs = d_obs;
    so.ld = ld;
    so.nb_max = b->info.num_objs - 2;
    so.dimo = b->info.dimo;
    so.is_variation = b->info.is_variation;
    so.full = full ? 1 : 0;
    so.flat = (full && !d_env_mask && ld == b->info.dimo && aligned16(d_obs) && (256 * b->info.dimo) % 4 == 0) ? 1 : 0;
    for (int i = 0; i < BPG_MAX_OBJS; i++) so.colors[i] = b->info.obj_colors[i];
    return so;
}

int bpg_reset_obs(bpg_batch *b, const int32_t *d_num_grey, const uint8_t *d_env_mask, float *d_obs, int64_t ld,
                  void *stream)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (d_obs && ld < b->info.dimo) return fail(BPG_ERR_ARG, "ld must be >= dimo");
    if (!aligned4(d_obs)) return fail(BPG_ERR_ARG, "d_obs must be 4-byte aligned");
    DeviceGuard guard(b->device);
    const int64_t nblk = (b->n + 255) / 256;
    hipLaunchKernelGGL(k_reset<true>, dim3((unsigned)(nblk < 4096 ? nblk : 4096)), dim3(256), 0, (hipStream_t)stream,
                       b->now, b->ever, b->n, reset_const(b, 1, d_num_grey), d_env_mask,
                       static_obs(b, d_obs, ld, true, d_env_mask));
    if (b->info.is_variation && !d_num_grey)     // only generator-drawn resets consume a stream index
        hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, (hipStream_t)stream, b->d_reset_counter, 1u);
    return check_launch("k_reset");
}

int bpg_fill_static_obs(bpg_batch *b, float *d_obs, int64_t ld, void *stream)
{
    if (!b || !d_obs) return fail(BPG_ERR_ARG, "NULL argument");
    if (ld < b->info.dimo) return fail(BPG_ERR_ARG, "ld must be >= dimo");
    if (!aligned4(d_obs)) return fail(BPG_ERR_ARG, "d_obs must be 4-byte aligned");
    if (!b->info.is_variation) return BPG_OK;
    DeviceGuard guard(b->device);
    const int64_t nblk = (b->n + 255) / 256;
    hipLaunchKernelGGL(k_reset<false>, dim3((unsigned)(nblk < 4096 ? nblk : 4096)), dim3(256), 0, (hipStream_t)stream,
                       b->now, b->ever, b->n, reset_const(b, 0, nullptr), (const uint8_t *)nullptr,
                       static_obs(b, d_obs, ld, false, nullptr));
    return check_launch("k_reset");
}

int bpg_reset(bpg_batch *b, const int32_t *d_num_grey, const uint8_t *d_env_mask, void *stream)
{
    return bpg_reset_obs(b, d_num_grey, d_env_mask, nullptr, 0, stream);
}

static int launch_step(bpg_batch *b, int src, const uint32_t *d_masks, float *d_reward, uint8_t *d_flags,
                       void *stream)
{
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    uint32_t *rw = reinterpret_cast<uint32_t *>(d_reward);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(d_masks) && aligned16(d_reward) && aligned4(d_flags);
    if (vec) {
        const dim3 grid(grid_for((b->n + 3) / 4, 256)), block(256);
        if (src == SRC_MASKS)
            hipLaunchKernelGGL(k_step_vec4<SRC_MASKS>, grid, block, 0, st, b->now, b->ever, d_masks, rw, d_flags,
                               b->n, c, s);
        else
            hipLaunchKernelGGL(k_step_vec4<SRC_SYNTH>, grid, block, 0, st, b->now, b->ever, d_masks, rw, d_flags,
                               b->n, c, s);
    } else {
        const dim3 grid(grid_for(b->n, 256)), block(256);
        if (src == SRC_MASKS)
            hipLaunchKernelGGL(k_step_scalar<SRC_MASKS>, grid, block, 0, st, b->now, b->ever, d_masks, rw,
                               d_flags, b->n, c, s);
        else
            hipLaunchKernelGGL(k_step_scalar<SRC_SYNTH>, grid, block, 0, st, b->now, b->ever, d_masks, rw,
                               d_flags, b->n, c, s);
    }
    return check_launch("k_step");
}

int bpg_step_masks(bpg_batch *b, const uint32_t *d_masks, float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (!aligned4(d_masks) || !aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    return launch_step(b, SRC_MASKS, d_masks, d_reward, d_flags, stream);
}

int bpg_step_synth(bpg_batch *b, float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (!aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    return launch_step(b, SRC_SYNTH, nullptr, d_reward, d_flags, stream);
}

int bpg_step_contacts(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                      float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b || !d_contacts || !d_ncon) return fail(BPG_ERR_ARG, "NULL argument");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (b->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    const dim3 grid(grid_for(b->n, 256, 4096)), block(256);
    if (maxcon % 4 == 0 && aligned16(d_contacts))
        hipLaunchKernelGGL(k_step_contacts<4>, grid, block, 0, (hipStream_t)stream, b->now, b->ever, d_contacts, d_ncon, maxcon,
                           b->d_id2obj, b->info.is_variation, reinterpret_cast<uint32_t *>(d_reward), d_flags, b->n, step_const(b));
    else
        hipLaunchKernelGGL(k_step_contacts<1>, grid, block, 0, (hipStream_t)stream, b->now, b->ever, d_contacts, d_ncon, maxcon,
                           b->d_id2obj, b->info.is_variation, reinterpret_cast<uint32_t *>(d_reward), d_flags, b->n, step_const(b));
    return check_launch("k_step_contacts");
}

static int launch_contacts_to_masks(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                                    uint32_t *d_masks, int T, int64_t ld, const int32_t *d_num_grey, hipStream_t st)
{
    const dim3 grid(grid_for(b->n * T, 256)), block(256);
    if (maxcon % 4 == 0 && aligned16(d_contacts))
        hipLaunchKernelGGL(k_contacts_to_masks<4>, grid, block, 0, st, b->ever, d_contacts, d_ncon, maxcon, b->d_id2obj,
                           b->info.is_variation, b->info.valid_pairs, d_masks, b->n, T, ld, d_num_grey);
    else
        hipLaunchKernelGGL(k_contacts_to_masks<1>, grid, block, 0, st, b->ever, d_contacts, d_ncon, maxcon, b->d_id2obj,
                           b->info.is_variation, b->info.valid_pairs, d_masks, b->n, T, ld, d_num_grey);
    return check_launch("k_contacts_to_masks");
}

int bpg_contacts_to_masks(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                          uint32_t *d_masks, void *stream)
{
    if (!b || !d_contacts || !d_ncon || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_masks)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (b->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    return launch_contacts_to_masks(b, d_contacts, d_ncon, maxcon, d_masks, 1, b->n, nullptr, (hipStream_t)stream);
}

int bpg_contacts_to_masks_tb(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int T, int maxcon,
                             const int32_t *d_num_grey, uint32_t *d_masks, int64_t ld_masks, void *stream)
{
    if (!b || !d_contacts || !d_ncon || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (T < 1 || T > (int)ELAPSED_MAX) return fail(BPG_ERR_ARG, "T must be in 1..1023");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_masks)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int64_t ld = ld_masks ? ld_masks : b->n;
    if (ld < b->n) return fail(BPG_ERR_ARG, "ld_masks must be 0 or >= n_envs");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (b->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    return launch_contacts_to_masks(b, d_contacts, d_ncon, maxcon, d_masks, T, ld, d_num_grey, (hipStream_t)stream);
}

int bpg_contacts_csr_to_masks(bpg_batch *b, const int16_t *d_pairs, const int64_t *d_offsets, int64_t total,
                              uint32_t *d_masks, void *stream)
{
    if (!b || !d_offsets || !d_masks || (!d_pairs && total > 0)) return fail(BPG_ERR_ARG, "NULL argument");
    if (total < 0) return fail(BPG_ERR_ARG, "total < 0");
    if (!aligned4(d_pairs) || !aligned4(d_masks) || ((uintptr_t)d_offsets & 7u))
        return fail(BPG_ERR_ARG, "pairs / masks must be 4-byte, offsets 8-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (b->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_contacts_csr_to_masks, dim3(grid_for(b->n, 256)), dim3(256), 0, (hipStream_t)stream, b->ever,
                       reinterpret_cast<const uint32_t *>(d_pairs), d_offsets, total, b->d_id2obj, b->info.is_variation,
                       b->info.valid_pairs, d_masks, b->n);
    return check_launch("k_contacts_csr_to_masks");
}

int bpg_step_record(bpg_batch *b, const bpg_step_args *args, void *stream)
{
    if (!b || !args) return fail(BPG_ERR_ARG, "NULL argument");
    static_assert(sizeof(bpg_step_args) == BPG_STEP_ARGS_MIN_SIZE, "BPG_STEP_ARGS_MIN_SIZE");
    bpg_step_args a;
    if (int rc = take_args(a, args, BPG_STEP_ARGS_MIN_SIZE, "bpg_step_args")) return rc;
    if ((uintptr_t)a.d_success_count & 7u) return fail(BPG_ERR_ARG, "counters must be 8-byte aligned");
    if ((a.d_u_out != nullptr) != (a.d_u_in != nullptr)) return fail(BPG_ERR_ARG, "d_u_in and d_u_out go together");
    if (!aligned16(a.d_u_in) || !aligned16(a.d_u_out)) return fail(BPG_ERR_ARG, "action buffers must be 16-byte aligned");
    if (a.d_ag && a.ld_ag < b->info.dimg) return fail(BPG_ERR_ARG, "ld_ag must be >= dimg");
    if (!aligned4(a.d_masks) || !aligned4(a.d_reward) || !aligned4(a.d_success_f32) || !aligned4(a.d_ag) || !aligned4(a.d_return_acc))
        return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    const PairTable tab = pair_table(b->info);
    const CountArgs ca = count_args(b, a.d_success_count, nullptr);
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    const bool vec = (b->info.side % 2 == 0) && (a.ld_ag % 4 == 0) && aligned16(a.d_ag);
    const bool synth = (a.d_masks == nullptr);
    uint32_t *rw = reinterpret_cast<uint32_t *>(a.d_reward);
    hipStream_t st = (hipStream_t)stream;
    int rc = dispatch_side(b->info.side, [&](auto side) {
        constexpr int S = decltype(side)::value;
        constexpr bool EVEN = (S % 2 == 0);
#define BPG_LAUNCH_SR(SRC, V)                                                                                  \
        hipLaunchKernelGGL((k_step_record<S, SRC, V, 1, true>), grid, block, 0, st, b->now, b->ever, a.d_masks, rw, a.d_flags, \
                           a.d_success_f32, a.d_ag, a.ld_ag, b->n, c, s, tab, reinterpret_cast<const float4 *>(a.d_u_in), \
                           reinterpret_cast<float4 *>(a.d_u_out), ca, a.d_return_acc)
        if (synth) {
            if (vec) BPG_LAUNCH_SR(SRC_SYNTH, EVEN); else BPG_LAUNCH_SR(SRC_SYNTH, false);
        } else {
            if (vec) BPG_LAUNCH_SR(SRC_MASKS, EVEN); else BPG_LAUNCH_SR(SRC_MASKS, false);
        }
#undef BPG_LAUNCH_SR
        return BPG_OK;
    });
    return rc != BPG_OK ? rc : check_launch("k_step_record");
}

int bpg_synth_masks(bpg_batch *b, int t0, int T, uint32_t *d_masks, void *stream)
{
    return bpg_synth_masks_ld(b, t0, T, d_masks, b ? b->n : 0, stream);
}

int bpg_synth_masks_ld(bpg_batch *b, int t0, int T, uint32_t *d_masks, int64_t ld, void *stream)
{
    if (!b || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (t0 < 0 || T < 1) return fail(BPG_ERR_ARG, "bad t0/T");
    if (ld < b->n) return fail(BPG_ERR_ARG, "ld must be >= n_envs");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_synth_masks, dim3(grid_for(b->n, 256)), dim3(256), 0, (hipStream_t)stream, b->ever, b->n,
                       t0, T, d_masks, ld, synth_const(b));
    return check_launch("k_synth_masks");
}

// Row stride for the time-major (T, ld) streams of an n-env batch.  A wave keeps 5-10 rows of the
// same columns in flight, and rows a large power of two apart put all of them on the same HBM
// channels: measured at 2^20 .. 2^23 envs, 53-68 % of peak with ld = n against 67-75 % with every row
// shifted by an odd number of KB (profiles/r02_tune_row_stride.log; a 128-byte shift is worse than none, a
// 16 KB shift as bad as none, 2 KB as bad as none on most boxes).  Among the odd shifts (1, 3, 5, 9 KB) there is
// no winner: bench.py's row_stride_probe on a dozen box sessions (profiles/r03_row_stride_probes.log) has them
// within +-5 % of each other in either direction -- as far apart as two allocations of the SAME layout in one
// process.  Below 2^20 envs the shift does not pay (neutral up to 2^18, -2 % at 2^19 and 3 * 2^18).
// ld = n below 2^20; otherwise n rounded up to 256 elements, plus 256 when that leaves an even multiple.
int64_t bpg_row_stride(int64_t n)
{
    if (n < 1) return 0;
    if (n < ((int64_t)1 << 20)) return n;
    int64_t ld = (n + 255) / 256 * 256;
    if ((ld / 256) % 2 == 0) ld += 256;
    return ld;
}

int bpg_get_achieved_goal_f32(bpg_batch *b, float *d_out, int64_t ld, void *stream)
{
    if (!b || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (ld < b->info.dimg) return fail(BPG_ERR_ARG, "ld must be >= dimg");
    if (!aligned4(d_out)) return fail(BPG_ERR_ARG, "d_out must be 4-byte aligned");
    DeviceGuard guard(b->device);
    const PairTable tab = pair_table(b->info);
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    const bool vec = (b->info.side % 2 == 0) && (ld % 4 == 0) && aligned16(d_out);
    int rc = dispatch_side(b->info.side, [&](auto side) {
        constexpr int S = decltype(side)::value;
        if (vec)
            hipLaunchKernelGGL((k_get_ag<S, float, (S % 2 == 0)>), grid, block, 0, (hipStream_t)stream, b->now,
                               b->ever, b->n, d_out, ld, tab);
        else
            hipLaunchKernelGGL((k_get_ag<S, float, false>), grid, block, 0, (hipStream_t)stream, b->now, b->ever,
                               b->n, d_out, ld, tab);
        return BPG_OK;
    });
    return rc != BPG_OK ? rc : check_launch("k_get_ag");
}

int bpg_get_achieved_goal_f64(bpg_batch *b, double *d_out, int64_t ld, void *stream)
{
    if (!b || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (ld < b->info.dimg) return fail(BPG_ERR_ARG, "ld must be >= dimg");
    DeviceGuard guard(b->device);
    const PairTable tab = pair_table(b->info);
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    int rc = dispatch_side(b->info.side, [&](auto side) {
        hipLaunchKernelGGL((k_get_ag<decltype(side)::value, double, false>), grid, block, 0, (hipStream_t)stream,
                           b->now, b->ever, b->n, d_out, ld, tab);
        return BPG_OK;
    });
    return rc != BPG_OK ? rc : check_launch("k_get_ag");
}

int bpg_get_desired_goal_f32(bpg_batch *b, float *d_out, int64_t ld, void *stream)
{
    if (!b || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (ld < b->info.dimg) return fail(BPG_ERR_ARG, "ld must be >= dimg");
    DeviceGuard guard(b->device);
    const int64_t total = b->n * b->info.dimg;
    hipLaunchKernelGGL(k_fill_goal, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, d_out, total,
                       b->info.dimg, ld, pair_table(b->info));
    return check_launch("k_fill_goal");
}

static int assemble_obs(bpg_batch *b, const bpg_phys_state *ph, void *d_obs, int64_t ld, bool f64, void *stream)
{
    if (!b || !ph || !d_obs) return fail(BPG_ERR_ARG, "NULL argument");
    if (ld < b->info.dimo) return fail(BPG_ERR_ARG, "ld must be >= dimo");
    if (!ph->d_grip_pos || !ph->d_grip_velp || !ph->d_gripper_state || !ph->d_gripper_vel || !ph->d_obj_pos ||
        !ph->d_obj_rot || !ph->d_obj_velp || !ph->d_obj_velr)
        return fail(BPG_ERR_ARG, "every bpg_phys_state array is required");
    DeviceGuard guard(b->device);
    ObsConst oc;
    oc.dimo = b->info.dimo;
    oc.nb_max = b->info.num_objs - 2;
    oc.is_variation = b->info.is_variation;
    for (int i = 0; i < BPG_MAX_OBJS; i++) oc.colors[i] = b->info.obj_colors[i];
    oc.dt = ph->dt;
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    const size_t lds = (size_t)64 * b->info.dimo * (f64 ? sizeof(double) : sizeof(float));   // <= 44.5 KB
    oc.vec = (aligned16(ph->d_gripper_state) && aligned16(ph->d_gripper_vel) && aligned16(ph->d_obj_pos) &&
              aligned16(ph->d_obj_rot) && aligned16(ph->d_obj_velp) && aligned16(ph->d_obj_velr)) ? 1 : 0;
#define BPG_LAUNCH_ASSEMBLE(NB)                                                                                   \
    do {                                                                                                          \
        if (f64)                                                                                                  \
            hipLaunchKernelGGL((k_assemble_obs<double, NB>), grid, block, lds, (hipStream_t)stream, b->ever, b->n, *ph, \
                               oc, static_cast<double *>(d_obs), ld);                                             \
        else                                                                                                      \
            hipLaunchKernelGGL((k_assemble_obs<float, NB>), grid, block, lds, (hipStream_t)stream, b->ever, b->n, *ph,  \
                               oc, static_cast<float *>(d_obs), ld);                                              \
    } while (0)
    switch (oc.nb_max) {
    case 1: BPG_LAUNCH_ASSEMBLE(1); break;
    case 2: BPG_LAUNCH_ASSEMBLE(2); break;
    case 3: BPG_LAUNCH_ASSEMBLE(3); break;
    case 4: BPG_LAUNCH_ASSEMBLE(4); break;
    default: return fail(BPG_ERR_ARG, "unsupported number of block slots");
    }
#undef BPG_LAUNCH_ASSEMBLE
    return check_launch("k_assemble_obs");
}

int bpg_assemble_obs_f64(bpg_batch *b, const bpg_phys_state *ph, double *d_obs, int64_t ld, void *stream)
{
    return assemble_obs(b, ph, d_obs, ld, true, stream);
}

int bpg_assemble_obs_f32(bpg_batch *b, const bpg_phys_state *ph, float *d_obs, int64_t ld, void *stream)
{
    return assemble_obs(b, ph, d_obs, ld, false, stream);
}

int bpg_set_action_f64(bpg_batch *b, const float *d_actions, double *d_out9, void *stream)
{
    if (!b || !d_actions || !d_out9) return fail(BPG_ERR_ARG, "NULL argument");
    if (!aligned16(d_actions)) return fail(BPG_ERR_ARG, "d_actions must be 16-byte aligned");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_set_action, dim3(grid_for(b->n, 256, GRID_UNCAPPED)), dim3(256), 0, (hipStream_t)stream, d_actions, b->n,
                       b->info.block_gripper, d_out9);
    return check_launch("k_set_action");
}

int bpg_mat2euler_f64(const double *d_mat, int64_t n, double *d_euler, int device_id, void *stream)
{
    if (!d_mat || !d_euler) return fail(BPG_ERR_ARG, "NULL argument");
    if (n <= 0) return n == 0 ? BPG_OK : fail(BPG_ERR_ARG, "n < 0");
    DeviceGuard guard(device_id);
    hipLaunchKernelGGL(k_mat2euler, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_mat, n, d_euler);
    return check_launch("k_mat2euler");
}

int bpg_randomize_objects(bpg_batch *b, const bpg_placement_args *a, double *d_xy, int32_t *d_status, void *stream)
{
    if (!b || !a || !d_xy) return fail(BPG_ERR_ARG, "NULL argument");
    const bool explicit_draws = a->d_eu || a->d_gz || a->d_gu;
    if (explicit_draws) {
        if (!a->d_eu || a->n_eu < 1) return fail(BPG_ERR_ARG, "d_eu / n_eu required with explicit variates");
        const bool needs_g = b->info.variant != BPG_GRIPPER_TOUCH && b->info.variant != BPG_TOPPLE_TOWER;
        if (needs_g && (!a->d_gz || !a->d_gu || a->n_gz < 2 || a->n_gu < 1))
            return fail(BPG_ERR_ARG, "d_gz / d_gu required for this variant");
    }
    DeviceGuard guard(b->device);
    const SynthConst s = synth_const(b);
    PlaceConst pc;
    pc.variant = b->info.variant;
    pc.is_variation = b->info.is_variation;
    pc.nb_max = b->info.num_objs - 2;
    pc.test = a->test;
    pc.challenge = a->challenge;
    pc.grip_x = a->grip_x; pc.grip_y = a->grip_y;
    pc.obj_range = a->obj_range; pc.max_obj_range = a->max_obj_range; pc.wrong_obj_range = a->wrong_obj_range;
    pc.eu = explicit_draws ? a->d_eu : nullptr;
    pc.gz = a->d_gz; pc.gu = a->d_gu;
    pc.n_eu = a->n_eu; pc.n_gz = a->n_gz; pc.n_gu = a->n_gu;
    pc.seed0 = s.seed0; pc.reset_counter = b->d_reset_counter + 1;
    pc.env_mask = a->d_env_mask;
    hipLaunchKernelGGL(k_randomize_objects, dim3(grid_for(b->n, 256)), dim3(256), 0, (hipStream_t)stream, b->ever,
                       b->n, pc, d_xy, d_status);
    if (!explicit_draws)
        hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, (hipStream_t)stream, b->d_reset_counter + 1, 1u);
    return check_launch("k_randomize_objects");
}

int bpg_state_ptrs(bpg_batch *b, uint32_t **d_now, uint32_t **d_ever)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (d_now) *d_now = b->now;
    if (d_ever) *d_ever = b->ever;
    return BPG_OK;
}

int bpg_counter_ptr(bpg_batch *b, uint32_t **d_counters)
{
    if (!b || !d_counters) return fail(BPG_ERR_ARG, "NULL argument");
    *d_counters = b->d_reset_counter;
    return BPG_OK;
}

static int compute_reward_f32(const float *d_ag, const float *d_g, bool bcast, int64_t n, int len, float *d_out,
                              int device_id, void *stream);

int bpg_compute_reward_f32(const float *d_ag, const float *d_g, int64_t n, int len, float *d_out, int device_id,
                           void *stream)
{
    return compute_reward_f32(d_ag, d_g, false, n, len, d_out, device_id, stream);
}

int bpg_compute_reward_f32_goal(const float *d_ag, const float *d_goal_row, int64_t n, int len, float *d_out,
                                int device_id, void *stream)
{
    return compute_reward_f32(d_ag, d_goal_row, true, n, len, d_out, device_id, stream);
}

static int compute_reward_f32(const float *d_ag, const float *d_g, bool bcast, int64_t n, int len, float *d_out,
                              int device_id, void *stream)
{
    using K = void (*)(const float *, const float *, int64_t, int, int, uint32_t *);
    return compute_reward_launch<float, K>(k_compute_reward_f32<true, true>, k_compute_reward_f32<true, false>,
                                           k_compute_reward_f32<false, true>, k_compute_reward_f32<false, false>, d_ag, d_g, bcast, n,
                                           len, d_out, device_id, stream, "k_compute_reward_f32");
}

static int compute_reward_f64(const double *d_ag, const double *d_g, bool bcast, int64_t n, int len, float *d_out,
                              int device_id, void *stream)
{
    using K = void (*)(const double *, const double *, int64_t, int, int, uint32_t *);
    return compute_reward_launch<double, K>(k_compute_reward_f64<true, true>, k_compute_reward_f64<true, false>,
                                            k_compute_reward_f64<false, true>, k_compute_reward_f64<false, false>, d_ag, d_g, bcast, n,
                                            len, d_out, device_id, stream, "k_compute_reward_f64");
}

int bpg_compute_reward_f64(const double *d_ag, const double *d_g, int64_t n, int len, float *d_out, int device_id,
                           void *stream)
{
    return compute_reward_f64(d_ag, d_g, false, n, len, d_out, device_id, stream);
}

int bpg_compute_reward_f64_goal(const double *d_ag, const double *d_goal_row, int64_t n, int len, float *d_out,
                                int device_id, void *stream)
{
    return compute_reward_f64(d_ag, d_goal_row, true, n, len, d_out, device_id, stream);
}

int bpg_compute_reward_packed(const uint32_t *d_now, const uint32_t *d_ever, uint32_t want_touch,
                              uint32_t want_never, int64_t n, float *d_out, int device_id, void *stream)
{
    if (!d_now || !d_ever || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (n <= 0) return n == 0 ? BPG_OK : fail(BPG_ERR_ARG, "n < 0");
    DeviceGuard guard(device_id);
    hipLaunchKernelGGL(k_compute_reward_packed, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_now,
                       d_ever, want_touch & PAIR_BITS, want_never & PAIR_BITS, n,
                       reinterpret_cast<uint32_t *>(d_out));
    return check_launch("k_compute_reward_packed");
}

static int launch_returns_from_okbits(const uint64_t *d_okbits, double *d_G, int64_t n, int T, double gamma, hipStream_t st)
{
    const PowTable tab = pow_table(gamma, T);
    constexpr int WPB = DR_LANE_WPB;
    const dim3 pgrid((unsigned)((n + 64 * WPB - 1) / (64 * WPB))), pblock(64 * WPB);
    if (T == 50 && aligned16(d_G))
        hipLaunchKernelGGL((k_returns_from_okbits_lane<50, WPB>), pgrid, pblock, 0, st, tab, d_okbits, d_G, n);
    else if (T == 25)
        hipLaunchKernelGGL((k_returns_from_okbits_lane<25, WPB>), pgrid, pblock, 0, st, tab, d_okbits, d_G, n);
    else if (T == 10 && aligned16(d_G))
        hipLaunchKernelGGL((k_returns_from_okbits_lane<10, WPB>), pgrid, pblock, 0, st, tab, d_okbits, d_G, n);
    else if (T <= 32)                                  // every other horizon: the smallest register-resident form that holds it
        hipLaunchKernelGGL((k_returns_from_okbits_lane_rt<32, WPB>), pgrid, pblock, 0, st, tab, d_okbits, d_G, n, T);
    else
        hipLaunchKernelGGL((k_returns_from_okbits_lane_rt<64, WPB>), pgrid, pblock, 0, st, tab, d_okbits, d_G, n, T);
    return check_launch("k_returns_from_okbits");
}

// gamma for which the conditional-subtract form of the return (returns_from_okbits, the `plain` waves of
// k_discounted_return_lane) has the reference's bits: every gamma**k, k < T, finite and >= 0.  With gamma < 0 the
// reference's `+= gamma**k * r` turns a -0.0 accumulator into +0.0 (the product of an odd power and -0.0 is +0.0), and
// an overflowing power makes inf * -0.0 = NaN; both need the real product + sum.
static bool gamma_is_subtractable(double gamma, int T)
{
    if (!(gamma >= 0.0) || std::signbit(gamma)) return false;      // -0.0 too: (-0.0)**k * -0.0 is +0.0 for odd k
    return std::isfinite(std::pow(gamma, (double)(T > 1 ? T - 1 : 1)));
}

int bpg_rollout(bpg_batch *b, const bpg_rollout_args *args, void *stream)
{
    if (!b || !args) return fail(BPG_ERR_ARG, "NULL argument");
    // struct versioning (include/bpgym.h): fields the caller's header did not have are zero; bytes this library does
    // not know must be zero
    static_assert(offsetof(bpg_rollout_args, gamma) + sizeof(double) == BPG_ROLLOUT_ARGS_MIN_SIZE, "BPG_ROLLOUT_ARGS_MIN_SIZE");
    bpg_rollout_args local;
    if (int rc = take_args(local, args, BPG_ROLLOUT_ARGS_MIN_SIZE, "bpg_rollout_args")) return rc;
    const bpg_rollout_args *a = &local;
    if (a->T < 1 || a->T > (int)ELAPSED_MAX) return fail(BPG_ERR_ARG, "T must be in 1..1023");
    if ((a->d_success_count && ((uintptr_t)a->d_success_count & 7u)) ||
        (a->d_zero_count && ((uintptr_t)a->d_zero_count & 7u)))
        return fail(BPG_ERR_ARG, "counters must be 8-byte aligned");
    if (a->d_success_count && a->d_success_count == a->d_zero_count)
        return fail(BPG_ERR_ARG, "d_zero_count must differ from d_success_count");
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    const ResetConst rc = reset_const(b, a->reset_first ? 1 : 0, a->d_num_grey);
    hipStream_t st = (hipStream_t)stream;
    uint32_t *rw = reinterpret_cast<uint32_t *>(a->d_reward_tb);
    const CountArgs ca = count_args(b, a->d_success_count, a->d_zero_count);
    const bool episode = a->d_ep_ag || a->d_ep_g || a->d_ep_success;
    const int64_t ldm = a->ld_masks ? a->ld_masks : b->n, ldo = a->ld_out ? a->ld_out : b->n;
    if (ldm < b->n || ldo < b->n) return fail(BPG_ERR_ARG, "ld_masks / ld_out must be 0 or >= n_envs");
    // The policy-gradient return (policy_gradient/rollout.py:255-258).  Up to 64 steps the step rewards of an
    // env are one 64-bit word: the rollout launch stores it (8 bytes per env, handle-owned) and a second launch
    // on the same stream expands the words into the G rows -- no reward stream is written or read.  Longer
    // horizons go through the reward stream.
    enum { G_NONE, G_OKBITS, G_FROM_REWARD } gmode = G_NONE;
    if (a->d_G) {
        if (!(a->gamma == a->gamma)) return fail(BPG_ERR_ARG, "gamma is NaN");
        if (((uintptr_t)a->d_G & 7u) != 0) return fail(BPG_ERR_ARG, "d_G must be 8-byte aligned");
        if (a->T <= 64 && gamma_is_subtractable(a->gamma, a->T)) gmode = G_OKBITS;
        else if (a->T > DR_TMAX) return fail(BPG_ERR_ARG, "d_G needs T <= 319");
        else if (!a->d_reward_tb)
            return fail(BPG_ERR_ARG, "d_G with T > 64, a negative gamma or overflowing powers of gamma needs d_reward_tb "
                                     "(the return is then computed from the reward stream)");
        else gmode = G_FROM_REWARD;
    }
    if (a->d_episode_return && ((uintptr_t)a->d_episode_return & 3u) != 0)
        return fail(BPG_ERR_ARG, "d_episode_return must be 4-byte aligned");
    if (gmode == G_OKBITS && !b->d_okbits) {
        // the step-ok words are handle-owned scratch, allocated by the first call that asks for d_G: a hipMalloc is not
        // a stream operation, so that call must not sit inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone)
            return fail(BPG_ERR_ARG, "the first bpg_rollout call with d_G allocates the handle's step-ok words: make it once "
                                     "outside the stream capture (a warm-up call does)");
        if (hipMalloc(&b->d_okbits, (size_t)((b->n + 1) & ~(int64_t)1) * sizeof(uint64_t)) != hipSuccess) {
            b->d_okbits = nullptr;
            return fail(BPG_ERR_HIP, "hipMalloc failed for the step-ok words");
        }
    }
    uint64_t *okbits = gmode == G_OKBITS ? b->d_okbits : nullptr;
    // several rollouts in one launch (stream path only): rollout r = reset + T steps + count on rows r*T .. of the streams
    const int rollouts = a->rollouts > 1 ? a->rollouts : 1;
    if (a->rollouts < 0 || a->rollouts > 4096) return fail(BPG_ERR_ARG, "rollouts must be in 0..4096");
    if (rollouts > 1 && (!a->reset_first || episode || a->d_G))
        return fail(BPG_ERR_ARG, "rollouts > 1 needs reset_first and takes neither episode arrays nor d_G");
    int rcode = BPG_OK;
    if (!episode) {
        const int out = (rw ? OUT_REWARD : 0) | (a->d_flags_tb ? OUT_FLAGS : 0);
        const RolloutGeometry geo = rollout_geometry(b->n, ldm, ldo, a->d_masks, a->d_reward_tb, a->d_flags_tb,
                                                     a->d_episode_return);
        const dim3 grid(geo.grid), block(geo.block);
#define BPG_LAUNCH_ROLLOUT_4(SRC, EPL, OUT, RST, UNR, PIPE, LB, OKB, MULTI, RET)                               \
        hipLaunchKernelGGL((k_rollout<SRC, true, EPL, OUT, RST, UNR, PIPE, true, LB, OKB, MULTI, RET>), grid, block, 0, st, \
                           b->now, b->ever, a->d_masks, rw, a->d_flags_tb, b->n, a->T, ldm, ldo, c, s, rc, ca, okbits,  \
                           rollouts, reinterpret_cast<uint32_t *>(a->d_episode_return))
        // the in-loop counter only where the return is asked for and there is no step-ok word to take it from
#define BPG_LAUNCH_ROLLOUT_3(SRC, EPL, OUT, RST, UNR, PIPE, LB, OKB, MULTI)                                    \
        do {                                                                                                  \
            if (a->d_episode_return && !(OKB)) BPG_LAUNCH_ROLLOUT_4(SRC, EPL, OUT, RST, UNR, PIPE, LB, false, MULTI, true); \
            else BPG_LAUNCH_ROLLOUT_4(SRC, EPL, OUT, RST, UNR, PIPE, LB, OKB, MULTI, false);                   \
        } while (0)
#define BPG_LAUNCH_ROLLOUT_2(SRC, EPL, OUT, UNR, PIPE, LB, OKB)                                                \
        do {                                                                                                  \
            if (rollouts > 1) BPG_LAUNCH_ROLLOUT_3(SRC, EPL, OUT, true, UNR, PIPE, LB, false, true);           \
            else if (a->reset_first) BPG_LAUNCH_ROLLOUT_3(SRC, EPL, OUT, true, UNR, PIPE, LB, OKB, false);     \
            else BPG_LAUNCH_ROLLOUT_3(SRC, EPL, OUT, false, UNR, PIPE, LB, OKB, false);                        \
        } while (0)
#define BPG_LAUNCH_ROLLOUT_1(SRC, EPL, OUT, UNR, PIPE, LB)                                                     \
        do {                                                                                                  \
            if (okbits) BPG_LAUNCH_ROLLOUT_2(SRC, EPL, OUT, UNR, PIPE, LB, true);                              \
            else BPG_LAUNCH_ROLLOUT_2(SRC, EPL, OUT, UNR, PIPE, LB, false);                                    \
        } while (0)
#define BPG_LAUNCH_ROLLOUT(SRC, EPL, UNR, PIPE, LB)                                                            \
        do {                                                                                                  \
            switch (out) {                                                                                    \
            case 0: BPG_LAUNCH_ROLLOUT_1(SRC, EPL, 0, UNR, PIPE, LB); break;                                  \
            case 1: BPG_LAUNCH_ROLLOUT_1(SRC, EPL, 1, UNR, PIPE, LB); break;                                  \
            case 2: BPG_LAUNCH_ROLLOUT_1(SRC, EPL, 2, UNR, PIPE, LB); break;                                  \
            default: BPG_LAUNCH_ROLLOUT_1(SRC, EPL, 3, UNR, PIPE, LB); break;                                 \
            }                                                                                                 \
        } while (0)
        // the instantiations below are the only ones: LB is the block size the shape is launched with
        if (a->d_masks) {
            if (geo.epl == 2 && geo.block == 1024) BPG_LAUNCH_ROLLOUT(SRC_MASKS, 2, 5, false, 1024);
            else if (geo.epl == 2) BPG_LAUNCH_ROLLOUT(SRC_MASKS, 2, 5, false, 256);
            else if (geo.unroll == 5) BPG_LAUNCH_ROLLOUT(SRC_MASKS, 1, 5, true, 256);
            else BPG_LAUNCH_ROLLOUT(SRC_MASKS, 1, 10, true, 256);
        } else {                                   // nothing to request: the plain loop
            if (geo.epl == 2) BPG_LAUNCH_ROLLOUT(SRC_SYNTH, 2, 10, false, 256);
            else BPG_LAUNCH_ROLLOUT(SRC_SYNTH, 1, 5, false, 256);
        }
#undef BPG_LAUNCH_ROLLOUT
#undef BPG_LAUNCH_ROLLOUT_1
#undef BPG_LAUNCH_ROLLOUT_2
#undef BPG_LAUNCH_ROLLOUT_3
#undef BPG_LAUNCH_ROLLOUT_4
        rcode = check_launch("k_rollout");
    } else {
        if ((a->d_ep_ag && !aligned16(a->d_ep_ag)) || (a->d_ep_g && !aligned16(a->d_ep_g)) ||
            (a->d_ep_success && !aligned4(a->d_ep_success)))
            return fail(BPG_ERR_ARG, "episode buffers must be 16-byte aligned");
        // Rows per chunk.  One chunk whenever the whole episode fits a 64-row (33 KB) tile: the wave then
        // streams ONE contiguous 64 x (T+1) x dimg region.  Splitting T = 50 into two 26-row chunks was
        // measured 30 % SLOWER (4.1 vs 3.1 ms for the full episode, 2.4 vs 1.9 ms ag-only) although it
        // doubles the waves per CU: the per-env 3.7 KB segments with 3.7 KB holes lose the long contiguous
        // write streams.  Longer horizons are chunked because the tile would not fit.
        const int rows_total = a->T + 1;
        int nchunks = 1;
        while ((rows_total + nchunks - 1) / nchunks > 64) nchunks++;
        const int RC = (rows_total + nchunks - 1) / nchunks;
        const int R = RC | 1;
        const size_t lds = (size_t)64 * R * sizeof(uint2) + sizeof(PairTable);
        const PairTable tab = pair_table(b->info);
        // two waves per 64-env block: the first steps the envs, both stream the block's region (2.5-3 %
        // over one wave on three boxes; three or four waves add nothing)
        const dim3 grid((unsigned)((b->n + 63) / 64)), block(128);
        const bool synth = (a->d_masks == nullptr);
        rcode = dispatch_side(b->info.side, [&](auto side) {
            constexpr int S = decltype(side)::value;
            if (synth)
                hipLaunchKernelGGL((k_episode<S, SRC_SYNTH, false>), grid, block, lds, st, b->now, b->ever, a->d_masks, rw,
                                   a->d_flags_tb, a->d_ep_ag, a->d_ep_g, a->d_ep_success, b->n, a->T, ldm, ldo, RC, R, c, s,
                                   tab, rc, ca, okbits, a->d_episode_return);
            else
                hipLaunchKernelGGL((k_episode<S, SRC_MASKS, false>), grid, block, lds, st, b->now, b->ever, a->d_masks, rw,
                                   a->d_flags_tb, a->d_ep_ag, a->d_ep_g, a->d_ep_success, b->n, a->T, ldm, ldo, RC, R, c, s,
                                   tab, rc, ca, okbits, a->d_episode_return);
            return BPG_OK;
        });
        if (rcode == BPG_OK) rcode = check_launch("k_episode");
    }
    if (rcode == BPG_OK && gmode == G_OKBITS) rcode = launch_returns_from_okbits(b->d_okbits, a->d_G, b->n, a->T, a->gamma, st);
    if (rcode == BPG_OK && gmode == G_FROM_REWARD)
        rcode = bpg_discounted_return_ld(a->d_reward_tb, b->n, ldo, a->T, a->gamma, a->d_G, b->device, stream);
    if (rcode == BPG_OK && a->reset_first && b->info.is_variation && !a->d_num_grey) {
        hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, st, b->d_reset_counter, (uint32_t)rollouts);
        rcode = check_launch("k_bump_counter");
    }
    return rcode;
}

int bpg_success_count(bpg_batch *b, unsigned long long *d_count, void *stream)
{
    if (!b || !d_count) return fail(BPG_ERR_ARG, "NULL argument");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_success_count, dim3(fat_grid((b->n + 3) / 4)), dim3(1024), 0, (hipStream_t)stream,
                       b->ever, b->n, count_args(b, d_count, nullptr));
    return check_launch("k_success_count");
}

int bpg_discounted_return(const float *d_reward_tb, int64_t n, int T, double gamma, double *d_G, int device_id,
                          void *stream)
{
    return bpg_discounted_return_ld(d_reward_tb, n, n, T, gamma, d_G, device_id, stream);
}

int bpg_discounted_return_ld(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma, double *d_G,
                             int device_id, void *stream)
{
    if (!d_reward_tb || !d_G) return fail(BPG_ERR_ARG, "NULL argument");
    if (n < 1 || T < 1 || T > DR_TMAX) return fail(BPG_ERR_ARG, "n must be >= 1 and T in 1..319");
    if (ld < n) return fail(BPG_ERR_ARG, "ld must be >= n");
    DeviceGuard guard(device_id);
    if ((T == 50 || T == 25 || T == 10) && ld < ((int64_t)1 << 28) && aligned16(d_G)) {
        // register-resident path for the horizons in use (T = max_episode_steps = 50): one env per lane
        const PowTable tab = pow_table(gamma, T);
        const int plain_ok = gamma_is_subtractable(gamma, T) ? 1 : 0;
        const dim3 pgrid((unsigned)((n + 64 * DR_LANE_WPB - 1) / (64 * DR_LANE_WPB))), pblock(64 * DR_LANE_WPB);
        hipStream_t st = (hipStream_t)stream;
        if (T == 50)
            hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, DR_LANE_WPB>), pgrid, pblock, 0, st, tab, d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
        else if (T == 25)
            hipLaunchKernelGGL((k_discounted_return_lane<25, DRM_EXACT, DR_LANE_WPB>), pgrid, pblock, 0, st, tab, d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
        else
            hipLaunchKernelGGL((k_discounted_return_lane<10, DRM_EXACT, DR_LANE_WPB>), pgrid, pblock, 0, st, tab, d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
        return check_launch("k_discounted_return_lane");
    }
    if (T <= 64 && ld < ((int64_t)1 << 28)) {              // every other horizon up to 64: the same form with 32 or 64 accumulators
        const PowTable tab = pow_table(gamma, T);
        const int plain_ok = gamma_is_subtractable(gamma, T) ? 1 : 0;
        const dim3 pgrid((unsigned)((n + 64 * DR_LANE_WPB - 1) / (64 * DR_LANE_WPB))), pblock(64 * DR_LANE_WPB);
        if (T <= 32)
            hipLaunchKernelGGL((k_discounted_return_lane<32, DRM_EXACT, DR_LANE_WPB, G_TILE_ROWS, true>), pgrid, pblock, 0,
                               (hipStream_t)stream, tab, d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
        else
            hipLaunchKernelGGL((k_discounted_return_lane<64, DRM_EXACT, DR_LANE_WPB, G_TILE_ROWS, true>), pgrid, pblock, 0,
                               (hipStream_t)stream, tab, d_reward_tb, n, ld, gamma, d_G, plain_ok, T);
        return check_launch("k_discounted_return_lane<rt>");
    }
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    PowTableBig big;
    std::memset(&big, 0, sizeof(big));
    for (int k = 0; k < T; k++) big.v[k] = std::pow(gamma, (double)k);
    const int R = T | 1;
    const size_t lds = (size_t)64 * R * sizeof(double);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_discounted_return),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_discounted_return, grid, block, lds, (hipStream_t)stream, d_reward_tb, n, ld, T, R, big, d_G);
    return check_launch("k_discounted_return");
}

int bpg_discounted_return_scan(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma, double *d_G,
                               int device_id, void *stream)
{
    if (!d_reward_tb || !d_G) return fail(BPG_ERR_ARG, "NULL argument");
    if (n < 1 || T < 1 || T > DR_TMAX) return fail(BPG_ERR_ARG, "n must be >= 1 and T in 1..319");
    if (ld == 0) ld = n;
    if (ld < n) return fail(BPG_ERR_ARG, "ld must be 0 or >= n");
    DeviceGuard guard(device_id);
    if (T == 50 && ld < ((int64_t)1 << 28) && aligned16(d_G)) {
        PowTable none;
        std::memset(&none, 0, sizeof(none));
        const dim3 pgrid((unsigned)((n + 64 * DR_LANE_WPB - 1) / (64 * DR_LANE_WPB))), pblock(64 * DR_LANE_WPB);
        hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_SCAN, DR_LANE_WPB>), pgrid, pblock, 0, (hipStream_t)stream,
                           none, d_reward_tb, n, ld, gamma, d_G, 0, T);
        return check_launch("k_discounted_return_lane<scan>");
    }
    // (other horizons: the LDS-tile scan below -- the 32 / 64-accumulator register form measured SLOWER for the scan, 178 against
    // 132 us at T = 40 and 1 M envs: an O(T) scan has no arithmetic to hide behind its registers' cost in occupancy)
    const int R = T | 1;
    const size_t lds = (size_t)64 * R * sizeof(double);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_discounted_return_scan),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_discounted_return_scan, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, (hipStream_t)stream,
                       d_reward_tb, n, ld, T, R, gamma, d_G);
    return check_launch("k_discounted_return_scan");
}

int bpg_trim_expert_inputs(const float *d_o, const float *d_g, const float *d_ag, int64_t n, int dimo_in, int dimg_in,
                           int dimo_out, int dimg_out, int num_objs, int is_variation, void *d_o_out, float *d_g_out,
                           float *d_ag_out, uint32_t *d_bad_flag, int device_id, void *stream)
{
    if (!d_o || !d_g || !d_ag || !d_o_out || !d_g_out || !d_ag_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (n < 0) return fail(BPG_ERR_ARG, "n < 0");
    if (dimo_in < 1 || dimg_in < 1 || dimo_out < 1 || dimg_out < 1 || num_objs < 1) return fail(BPG_ERR_ARG, "bad dimensions");
    if (dimg_in > 64 || dimo_in > 1024) return fail(BPG_ERR_UNSUPPORTED, "trim supports dimg <= 64 and dimo <= 1024");
    int max_objs = (int)std::sqrt((double)dimg_in);                 // (int)(g.shape[1] ** 0.5), rollout.py:139
    while ((max_objs + 1) * (max_objs + 1) <= dimg_in) max_objs++;
    while (max_objs * max_objs > dimg_in) max_objs--;
    TrimConst tc{};
    tc.dimo_in = dimo_in; tc.dimg_in = dimg_in; tc.dimo_out = dimo_out; tc.dimg_out = dimg_out;
    int kept = 0;
    for (int i = 0; i < dimg_in; i++)
        if (i / max_objs < num_objs && i % max_objs < num_objs) {
            if (kept < 64) tc.sel[kept] = (uint8_t)i;
            kept++;
        }
    if (kept != dimg_out) return fail(BPG_ERR_ARG, "trim: the kept goal entries are not dimg (assert, rollout.py:146)");
    if (is_variation) {
        tc.num_blocks = num_objs - 2;
        tc.max_blocks = max_objs - 2;
        if (tc.num_blocks < 0 || tc.max_blocks < 0 || tc.max_blocks > TRIM_MAX_SLOTS || tc.num_blocks > TRIM_MAX_SLOTS)
            return fail(BPG_ERR_ARG, "trim: bad number of block slots");
        if (dimo_in < 1 + TRIM_ENV_FEATURES + TRIM_BLOCK * tc.max_blocks)
            return fail(BPG_ERR_ARG, "trim: the observation row is shorter than its block slots");
        if (dimo_out != TRIM_ENV_FEATURES + TRIM_BLOCK_BASE * tc.num_blocks)
            return fail(BPG_ERR_ARG, "trim: o_ is not dimo wide (assert, rollout.py:171)");
        if (!d_bad_flag) return fail(BPG_ERR_ARG, "trim of a Variation observation needs d_bad_flag");
    } else if (dimo_out > dimo_in) {
        return fail(BPG_ERR_ARG, "trim: o_ is not dimo wide (assert, rollout.py:171)");
    }
    if (n == 0) return BPG_OK;
    DeviceGuard guard(device_id);
    tc.vec_o = aligned16(d_o);
    const size_t lds = is_variation ? (size_t)TRIM_ENVS * sizeof(float) * dimo_in : 0;
    const dim3 grid((unsigned)((n + TRIM_ENVS - 1) / TRIM_ENVS)), block(256);
    const void *fn = is_variation ? reinterpret_cast<const void *>(k_trim<true>) : reinterpret_cast<const void *>(k_trim<false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(BPG_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    if (is_variation)
        hipLaunchKernelGGL(k_trim<true>, grid, block, lds, (hipStream_t)stream, d_o, d_g, d_ag, n, tc, d_o_out, d_g_out, d_ag_out, d_bad_flag);
    else
        hipLaunchKernelGGL(k_trim<false>, grid, block, lds, (hipStream_t)stream, d_o, d_g, d_ag, n, tc, d_o_out, d_g_out, d_ag_out, d_bad_flag);
    return check_launch("k_trim");
}

int bpg_nan_flag_f32(const float *d_x, int64_t rows, int cols, int64_t ld, uint32_t *d_flag, int device_id,
                     void *stream)
{
    if (!d_x || !d_flag) return fail(BPG_ERR_ARG, "NULL argument");
    if (rows < 0 || cols < 1 || ld < cols) return fail(BPG_ERR_ARG, "bad rows / cols / ld");
    if (rows == 0) return BPG_OK;
    DeviceGuard guard(device_id);
    if (ld == cols && aligned16(d_x))
        hipLaunchKernelGGL(k_nan_flag<true>, dim3(grid_for((rows * cols + 15) / 16, 256, GRID_UNCAPPED)), dim3(256), 0, (hipStream_t)stream,
                           d_x, rows, cols, ld, d_flag);
    else
        hipLaunchKernelGGL(k_nan_flag<false>, dim3(grid_for(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, d_x, rows,
                           cols, ld, d_flag);
    return check_launch("k_nan_flag");
}

}  // extern "C"
