// returns.hip -- the policy-gradient return G (policy_gradient/rollout.py:217-231, :255-258) from the step-ok words of a rollout
// launch (k_rollout<AUX_OKB>, k_episode): 8 bytes read and 8 T bytes written per env, bit-identical to the reference's float64
// accumulation.  (From a reward stream: returns_stream.hip.)
#include "returns.hip.h"

namespace bpgk {


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

}  // namespace bpgk

int bpgk::launch_returns_from_okbits(const uint64_t *d_okbits, double *d_G, int64_t n, int T, double gamma, hipStream_t st)
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
