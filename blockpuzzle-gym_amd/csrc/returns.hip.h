// returns.hip.h -- what the two return units share: the policy-gradient return of a rollout from its step-ok word
// (policy_gradient/rollout.py:255-258 as a conditional subtraction of gamma**k, register-resident) and the wave-level store
// of the G rows through a private LDS tile.
#ifndef BPG_RETURNS_HIP_H
#define BPG_RETURNS_HIP_H
#include "state.hip.h"

namespace bpgk {

constexpr int DR_LANE_WPB = 4;              // waves per block (each wave: 64 envs, a private 16-row LDS tile)


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

}  // namespace bpgk
#endif
