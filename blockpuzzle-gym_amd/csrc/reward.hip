// reward.hip -- BlocksEnv.compute_reward (fetch_env.py:135-143) as HER calls it (config.py:110-111), on float32 /
// float64 rows in NumPy's summation order and on packed states; and the success count (rollout.py:163-167).
#include "state.hip.h"

namespace bpgk {

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

// d_episode_fail_count -> d_episode_return: out[e] = -(float)count[e] (-0.0f for 0), 16 envs per lane (one 16-byte load, four
// 16-byte stores)
__global__ void __launch_bounds__(256)
k_return_from_fail_count(const uint8_t *__restrict__ cnt, float *__restrict__ out, int64_t n, int vec)
{
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nv = vec ? n >> 4 : 0;
    for (int64_t v = gtid; v < nv; v += stride) {
        const uint4 w = reinterpret_cast<const uint4 *>(cnt)[v];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
            st_f4<OUT_NT>(reinterpret_cast<float4 *>(out) + 4 * v + q,
                          make_float4(-(float)(ws[q] & 255u), -(float)((ws[q] >> 8) & 255u), -(float)((ws[q] >> 16) & 255u),
                                      -(float)(ws[q] >> 24)));
    }
    for (int64_t i = (nv << 4) + gtid; i < n; i += stride) out[i] = -(float)cnt[i];
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

}  // namespace bpgk

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

extern "C" {

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

int bpg_episode_return_from_fail_count(const uint8_t *d_count, float *d_out, int64_t n, int device_id, void *stream)
{
    if (!d_count || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (n <= 0) return n == 0 ? BPG_OK : fail(BPG_ERR_ARG, "n < 0");
    if (!aligned4(d_out)) return fail(BPG_ERR_ARG, "d_out must be 4-byte aligned");
    DeviceGuard guard(device_id);
    const int vec = (aligned16(d_count) && aligned16(d_out)) ? 1 : 0;
    hipLaunchKernelGGL(k_return_from_fail_count, dim3(grid_for((n + 15) / 16, 256, GRID_UNCAPPED)), dim3(256), 0, (hipStream_t)stream,
                       d_count, d_out, n, vec);
    return check_launch("k_return_from_fail_count");
}

int bpg_success_count(bpg_batch *b, unsigned long long *d_count, void *stream)
{
    if (!b || !d_count) return fail(BPG_ERR_ARG, "NULL argument");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_success_count, dim3(fat_grid((b->n + 3) / 4)), dim3(1024), 0, (hipStream_t)stream,
                       b->ever, b->n, count_args(b, d_count, nullptr));
    return check_launch("k_success_count");
}

}  // extern "C"
