// state.hip.h -- what every translation unit of libbpgym.so shares: the packed per-env state and its
// layout constants, the handle, one env step on the packed state (step_core), the in-launch reset, the
// build-defined synthetic contact source, the streaming load / store helpers, the sharded success count,
// the float views of the packed state, and the host-side launch helpers.  Device functions are
// __forceinline__, host helpers inline: nothing here needs relocatable device code.
//
// Reference semantics implemented by the library (paths relative to the reference root):
//   _step_callback   gym_blocks/envs/fetch_env.py:148-167
//   compute_reward   fetch_env.py:135-143   (-1.0f / -0.0f)
//   _is_success      fetch_env.py:275-281   (latched)
//   _sample_goal     fetch_env.py:260-273, Variation :682-695
//   _reset_sim       fetch_env.py:247-255,  Variation :646-679
//   RobotEnv.step    gym_blocks/envs/robot_env.py:57-69 (+ gym TimeLimit done)
//   generate_rollouts gym_blocks/rollout.py:75-172 (episode arrays, batch-major)
//
// Design: the symmetric ternary N x N touch matrix is stored as two bit planes over the 21
// unordered pairs of <= 6 objects, 8 bytes per env in two SoA uint32 arrays.  Everything is integer /
// bitmask streaming work: kernels are HBM-bound, use coalesced 8 / 16-byte-per-lane accesses, keep state in
// registers across fused steps, stage per-wave history tiles in LDS so that the float32 episode buffers are
// written as contiguous wave stores, and reduce success counts with ballot / popcount.  No MFMA (nothing here
// is a contraction).
//
// Nothing in csrc/ touches oracle/ and there is no CPU fallback: without a GPU every launch fails with
// BPG_ERR_HIP.
#ifndef BPG_STATE_HIP_H
#define BPG_STATE_HIP_H

#include <hip/hip_runtime.h>

// Bit-exact float64/float32 parity with the reference's NumPy arithmetic needs every product and sum rounded
// separately: HIP's __dmul_rn/__dadd_rn are plain operators, and hipcc contracts a*b+c into an fma by default.
// Nothing in this library is FLOP-bound, so contraction is simply off (also -ffp-contract=off in build.py).
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
#include "hooks.hip.h"

// functions one translation unit of the library calls in another: not part of the C ABI, not exported
#define BPG_INTERNAL __attribute__((visibility("hidden")))

// the thread's last error text (bpg_last_error) lives in abi.hip
BPG_INTERNAL int bpg_fail_(int code, const std::string &msg);
inline int fail(int code, const std::string &msg) { return bpg_fail_(code, msg); }

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(BPG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

constexpr int BPG_LIBRARY_VERSION = 105;                    // bpg_version(): 100 + round
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
                                          // bpg_reserve_return_scratch or the first bpg_rollout_v2 call with d_G (NULL until then)
    bpg_curriculum cur;                   // difficulty + placement ranges (fetch_env.py:340-348, :404-417, :561-563)
    bpg_batch *parent;                    // a view (bpg_create_view): the handle whose envs these are; NULL for a batch of its own
    hipEvent_t ev_fork, ev_join;          // bpg_rollout_train: created on first use (timing disabled), owned by the handle
};

// what a view shares with its parent and never copies: seed, id2obj tables, curriculum
inline const bpg_batch *root_of(const bpg_batch *b) { return b->parent ? b->parent : b; }
inline bpg_batch *root_of(bpg_batch *b) { return b->parent ? b->parent : b; }

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

// build-defined counter-based contact source, a pure function
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
// launch (round 4's tools/tune_round4.hip, profiles/r04_tune_step_train.log) and per kernel with -DBPG_OUT_NT=0 against the
// default on the same box (tools/ab_output_stores.py, profiles/r04_ab_output_stores.log).  The multi-GB episode writer
// and the reset's observation rows are the exceptions (plain stores: profiles/r04_tune_episode_nontemporal.log; reset
// + rows 8.4 us plain against 9.6 us at 65 536 envs), and so is everything a later step of the same launch train re-reads
// (the packed state, contact masks).
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

// ------------------------------ single step ---------------------------------
constexpr int SRC_MASKS = 0, SRC_SYNTH = 1;      // plain ints: see MODE_* in rollout.hip

// the step's contact mask: read from memory, or the synthetic source at the env's elapsed step
template <int SRC>
__device__ __forceinline__ uint32_t fetch_mask(uint32_t m, uint32_t now, uint32_t ever, uint32_t local_env,
                                               const SynthConst &s)
{
    if (SRC == SRC_MASKS) return m;
    return synth_mask(synth_lane(s, local_env, nobj_of(ever)), (now >> ELAPSED_SHIFT) & ELAPSED_MAX);
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

// which per-step streams a fused rollout writes (k_rollout's OUT template argument: rollout.hip)
constexpr int OUT_REWARD = 1, OUT_FLAGS = 2;

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

// gamma**k table of the register-resident return kernels (returns.hip), built on the host
constexpr int DR_TMAX_REG = 64;
struct PowTable { double v[DR_TMAX_REG]; };
constexpr int DR_TMAX = 319;                  // longest horizon of any return kernel: 64 x (T|1) x 8 B <= 160 KB of LDS

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

// ---------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------
// cap = 2048: 256 CUs x 8 blocks, grid-stride the rest.  Pure read streams run faster as one short-lived
// block per tile (blocks beyond the resident set start in address order as earlier ones retire: the
// box's read ceiling is 6.2-6.7 TB/s that way against 5.2-5.5 for a grid-stride loop,
// profiles/r02_stream_ceiling.log) -- those pass GRID_UNCAPPED; kernels with scattered small stores
// (k_fill_goal: 2x slower uncapped) keep the cap.
constexpr int64_t GRID_UNCAPPED = (int64_t)1 << 22;
inline int grid_for(int64_t work_items, int block, int64_t cap = 2048)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

inline StepConst step_const(const bpg_batch *b)
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

inline CountArgs count_args(const bpg_batch *b, unsigned long long *count, unsigned long long *zero_next)
{
    CountArgs ca;
    ca.count = count;
    ca.zero_next = zero_next;
    ca.shards = b->d_count_shards;
    return ca;
}

// gamma ** k as Python's float pow (C pow), k < DR_TMAX_REG
inline PowTable pow_table(double gamma, int T)
{
    PowTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (int k = 0; k < T && k < DR_TMAX_REG; k++) tab.v[k] = std::pow(gamma, (double)k);
    return tab;
}

inline SynthConst synth_const(const bpg_batch *b)
{
    SynthConst s;
    s.seed0 = root_of(b)->seed + 1000ull * (uint64_t)b->env_base;        // rollout.py:209-210 on the global env index
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

inline ResetConst reset_const(const bpg_batch *b, int enabled, const int32_t *d_num_grey)
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

inline int fat_grid(int64_t work_items)
{
    int64_t g = (work_items + 1023) / 1024;
    if (g < 1) g = 1;
    if (g > 256) g = 256;        // one 1024-lane block per CU; <= 256 counter atomics per launch
    return (int)g;
}

inline PairTable pair_table(const bpg_variant_info &info)
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

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BPG_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return BPG_OK;
}

// versioned argument structs (include/bpgym.h): copies what the caller passed into a zeroed struct of this library's
// layout; bytes this library does not know must be zero
template <typename A>
inline int take_args(A &local, const A *args, size_t min_size, const char *what)
{
    const uint32_t tag = args->struct_size;
    const size_t caller_size = tag & 0xFFFFu;
    if ((tag & 0xFFFF0000u) != BPG_ARGS_MAGIC || caller_size < min_size || caller_size > 4096 || caller_size % 8 != 0)
        return fail(BPG_ERR_ARG, std::string(what) + ".struct_size is not BPG_ARGS_TAG(sizeof) (use the _INIT macro of include/bpgym.h; "
                                 "this library is bpg_version() " + std::to_string(BPG_LIBRARY_VERSION) + ")");
    std::memset(&local, 0, sizeof(local));
    std::memcpy(&local, args, caller_size < sizeof(local) ? caller_size : sizeof(local));
    for (size_t i = sizeof(local); i < caller_size; i++)
        if (reinterpret_cast<const unsigned char *>(args)[i] != 0)
            return fail(BPG_ERR_UNSUPPORTED, std::string(what) + " carries fields this library (bpg_version() " +
                                             std::to_string(BPG_LIBRARY_VERSION) + ") does not know");
    return BPG_OK;
}

// gamma for which the conditional-subtract form of the return (returns_from_okbits, the `plain` waves of
// k_discounted_return_lane) has the reference's bits: every gamma**k, k < T, finite and >= 0.  With gamma < 0 the
// reference's `+= gamma**k * r` turns a -0.0 accumulator into +0.0 (the product of an odd power and -0.0 is +0.0), and
// an overflowing power makes inf * -0.0 = NaN; both need the real product + sum.
inline bool gamma_is_subtractable(double gamma, int T)
{
    if (!(gamma >= 0.0) || std::signbit(gamma)) return false;      // -0.0 too: (-0.0)**k * -0.0 is +0.0 for odd k
    return std::isfinite(std::pow(gamma, (double)(T > 1 ? T - 1 : 1)));
}

// ---- launches one translation unit enqueues for another (host functions; no device code crosses a unit) ----
BPG_INTERNAL int enqueue_bump_counter(uint32_t *d_counter, uint32_t by, hipStream_t st);                       // abi.hip
BPG_INTERNAL int launch_returns_from_okbits(const uint64_t *d_okbits, double *d_G, int64_t n, int T, double gamma, hipStream_t st);   // returns.hip
struct EpisodeLaunch {                                                                              // episode.hip
    const uint32_t *d_masks; uint32_t *reward_tb; uint8_t *flags_tb; float *ep_ag, *ep_g, *ep_success;
    int T; int64_t ldm, ldo; ResetConst rc; CountArgs ca; uint64_t *okbits; float *ep_return; uint8_t *ep_fail;
};
BPG_INTERNAL int launch_episode(bpg_batch *b, const EpisodeLaunch &e, hipStream_t st);

}  // namespace bpgk
using namespace bpgk;

#endif
