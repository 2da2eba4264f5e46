// obs_reset_place.hip -- everything around a step that is not the step: reset (robot_env.py:71-82, fetch_env.py:247-255,
// :646-679), the float views of the packed state (_get_obs, fetch_env.py:219-228), the observation assembly
// (:187-228, :567-621), _set_action (:170-185), _randomize_objects (:328-336, 370-399, 448-517, 697-764, 777-787),
// the pg worker's trim and the NaN watch (rollout.py:139-142).
#include "state.hip.h"

namespace bpgk {

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

}  // namespace bpgk

extern "C" {

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
    if (int rc = check_launch("k_reset")) return rc;
    if (b->info.is_variation && !d_num_grey && !b->parent)     // only generator-drawn resets consume a stream index
        return enqueue_bump_counter(b->d_reset_counter, 1u, (hipStream_t)stream);     // (views: bpg_advance_generator on the parent)
    return BPG_OK;
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
    if (int rc = check_launch("k_randomize_objects")) return rc;
    if (!explicit_draws && !b->parent) return enqueue_bump_counter(b->d_reset_counter + 1, 1u, (hipStream_t)stream);
    return BPG_OK;
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
