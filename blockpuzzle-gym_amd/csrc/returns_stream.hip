// returns_stream.hip -- the policy-gradient return G (policy_gradient/rollout.py:255-258) from a (T, n) float32 reward stream:
// register-resident (T <= 64) and LDS-tile (T <= 319) forms in the reference's summation order, and SURVEY 8(f3)'s O(T) scan.
#include "returns_lane.hip.h"

namespace bpgk {



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

}  // namespace bpgk

extern "C" {

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
    if (T <= 64 && ld < ((int64_t)1 << 28))                // every other horizon up to 64: the same form with 32 or 64 accumulators
        return launch_return_lane_rt(d_reward_tb, n, ld, T, gamma, d_G, (hipStream_t)stream);
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

}  // extern "C"
