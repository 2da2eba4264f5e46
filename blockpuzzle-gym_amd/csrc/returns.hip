// returns.hip -- the policy-gradient return G (policy_gradient/rollout.py:217-231, :255-258) from the step-ok words of a
// rollout launch or from a reward stream, bit-identical to the reference's float64 accumulation; and the O(T) scan.
#include "state.hip.h"

namespace bpgk {

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

}  // extern "C"
