// returns_lane.hip.h -- the register-resident return kernel from a reward stream (one env per lane, T <= 64), shared by the
// two units that instantiate it (returns_stream.hip: T = 50 / 25 / 10 and the scan; returns_stream_rt.hip: any other horizon
// through the 32- / 64-accumulator forms).
#ifndef BPG_RETURNS_LANE_HIP_H
#define BPG_RETURNS_LANE_HIP_H
#include "returns.hip.h"

namespace bpgk {

// ---- lane = env return kernels (T = 50 / 25 / 10) --------------------------------------------------------
// EXACT: the reference's float64 result bit for bit.  Every reward this env produces is -1.0f or -0.0f
// (fetch_env.py:141-143), and for those the reference's product-then-sum is exactly a conditional
// subtraction of gamma**k (returns_from_okbits above): a wave whose 64 x T rewards all lie in {-1.0f, -0.0f}
// takes that path -- half the float64 operations of the general one and no per-lane reward operand --, any
// other wave re-reads its rewards (they are L2-resident by then) and runs the general unfused product + sum,
// in the reference's order.  The choice is per wave and changes no result bit.
// SCAN: SURVEY 8(f3)'s O(T) reverse scan G[t] = fma(gamma, G[t+1], r[t]) -- NOT the reference's summation
// order: |G_scan - G_ref| <= 4 T 2^-53 sum_k gamma^k |r[t+k]| (include/bpgym.h states the bound).
constexpr int DRM_EXACT = 0, DRM_SCAN = 1;

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


// returns_stream_rt.hip: any horizon T <= 64 through the 32- / 64-accumulator instantiations
BPG_INTERNAL int launch_return_lane_rt(const float *d_reward_tb, int64_t n, int64_t ld, int T, double gamma, double *d_G, hipStream_t st);

}  // namespace bpgk
#endif
