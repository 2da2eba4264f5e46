// Standalone tuning harness (not part of the product): cost of candidate counter-based contact
// generators inside the fused rollout loop (contacts generated in-kernel: reward f32 + flags u8
// written per env-step, nothing read but the packed state), against the write-only ceiling of
// the same stores with no generator at all.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/tune_synth tools/tune_synth.hip
//   ./tools/tune_synth [envs]
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hashB(uint32_t x)       // hash-prospector [16 21f0aaad 15 735a2d97 15]
{
    x ^= x >> 16; x *= 0x21f0aaadU;
    x ^= x >> 15; x *= 0x735a2d97U;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t hash1a(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
__device__ __forceinline__ uint32_t hash1b(uint32_t x) { x ^= x >> 16; x *= 0x21f0aaadU; x ^= x >> 15; return x; }
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

struct GenCur {      // round 1: 4 x lowbias32, five fresh 21-bit fields, Bernoulli(5/32)
    uint32_t key, valid, forced;
    __device__ void init(uint32_t seed, uint32_t env, uint32_t v, uint32_t f) { key = synth_key(seed, env); valid = v; forced = f; }
    __device__ uint32_t mask(uint32_t t) const
    {
        const uint32_t base = key + 0x85EBCA6BU + (4u * t) * 0x9E3779B9U;
        const uint32_t w0 = hash32(base), w1 = hash32(base + 0x9E3779B9U), w2 = hash32(base + 2u * 0x9E3779B9U),
                       w3 = hash32(base + 3u * 0x9E3779B9U);
        uint32_t acc = ((w0 >> 21) | ((w1 >> 21) << 11)) & PAIR_BITS;
        acc &= w3; acc |= (w2 & PAIR_BITS); acc &= w1; acc &= w0;
        return (acc & valid) | forced;
    }
};

template <int MODE> struct Gen2 {   // two words per step; MODE 0: 2-round hashes + tail (5/32), 1: 2-round, p = 1/8,
                                    // 2: 1-round hashes + tail, 3: mum folds + tail
    uint32_t ka, kb, valid, forced;
    __device__ void init(uint32_t seed, uint32_t env, uint32_t v, uint32_t f)
    {
        ka = synth_key(seed, env); kb = hash32(ka ^ 0x85EBCA6BU); valid = v; forced = f;
    }
    __device__ uint32_t mask(uint32_t t) const
    {
        const uint32_t u = ka + (t + 1u) * 0x9E3779B9U;
        uint32_t a, b;
        if (MODE == 2) { a = hash1a(u); b = hash1b(u ^ kb); }
        else if (MODE == 3) {
            const uint64_t p = (uint64_t)u * 0xD6E8FEB9U, q = (uint64_t)(u ^ kb) * 0xCA5A8263U;
            const uint32_t x = (uint32_t)(p >> 32) ^ (uint32_t)p, y = (uint32_t)(q >> 32) ^ (uint32_t)q;
            const uint64_t p2 = (uint64_t)x * 0x9FB21C65U, q2 = (uint64_t)y * 0x1D8E4E27U;
            a = (uint32_t)(p2 >> 32) ^ (uint32_t)p2; b = (uint32_t)(q2 >> 32) ^ (uint32_t)q2;
        } else { a = hash32(u); b = hashB(u ^ kb); }
        const uint32_t C = (a >> 21) | ((b >> 21) << 11);
        uint32_t acc;
        if (MODE == 1) acc = a & b & C;
        else acc = a & b & (C | (rotl32(a, 7) & rotl32(b, 13)));
        return (acc & valid) | forced;
    }
};

struct GenLib {      // the library's source (synth_lane / synth_mask)
    SynthLane l;
    __device__ void init(uint32_t seed, uint32_t env, uint32_t v, uint32_t f)
    {
        l.ka = synth_key(seed, env); l.kb = hash32(l.ka ^ 0x85EBCA6BU); l.valid = v; l.forced = f;
    }
    __device__ uint32_t mask(uint32_t t) const { return synth_mask(l, t); }
};

struct GenZero {     // no generator: what the loop costs with a free mask
    uint32_t key, valid, forced;
    __device__ void init(uint32_t seed, uint32_t env, uint32_t v, uint32_t f) { key = seed + env; valid = v; forced = f; }
    __device__ uint32_t mask(uint32_t t) const { return ((key + t) & valid) | forced; }
};

// the rollout loop with the state kept unpacked (elapsed counter and latch in their own registers)
template <class GEN, int EPL, bool UNPACKED, bool DEP = false, bool BATCHLOOP = false>
__global__ void __launch_bounds__(1024)
v_synth(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, uint32_t *__restrict__ reward_tb,
        uint8_t *__restrict__ flags_tb, int64_t n, int T, StepConst c, uint32_t seed)
{
    using Vec = typename LaneVec<EPL>::vec;
    using Fl = typename LaneVec<EPL>::fl;
    const int64_t nvec = n / EPL;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint32_t an[EPL], ee[EPL];
        unpack<EPL>(reinterpret_cast<const Vec *>(now)[v], an);
        unpack<EPL>(reinterpret_cast<const Vec *>(ever)[v], ee);
        GEN g[EPL];
        uint32_t el[EPL], okl[EPL];
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            if (!DEP) {
                an[k] = 0; ee[k] = 6u << NOBJ_SHIFT;
                g[k].init(seed, (uint32_t)(EPL * v + k), 0x1FFFFEu, (1u << 8) | (1u << 9) | (1u << 10));
                el[k] = 0; okl[k] = 0;
            } else {
                const uint32_t nobj = nobj_of(ee[k]);
                g[k].init(seed, (uint32_t)(EPL * v + k), nobj == 6 ? 0x1FFFFEu : 0x0FFFFEu, nobj == 6 ? 0x700u : 0x300u);
                el[k] = (an[k] >> ELAPSED_SHIFT) & 7u; okl[k] = (uint32_t)((int32_t)ee[k] >> 31);
                ee[k] &= ~SUCC_BIT;
            }
        }
        Vec *rrow = reinterpret_cast<Vec *>(reward_tb) + v;
        Fl *frow = reinterpret_cast<Fl *>(flags_tb) + v;
        auto body = [&](int t) {
            uint32_t rb[EPL], fl[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                if (!UNPACKED) {
                    const uint32_t m = g[k].mask((an[k] >> ELAPSED_SHIFT) & ELAPSED_MAX);
                    step_core(an[k], ee[k], m, c, rb[k], fl[k]);
                } else {
                    const uint32_t m = g[k].mask(el[k]);
                    const uint32_t cm = m & c.valid;
                    an[k] = cm;
                    ee[k] |= cm;
                    const uint32_t bad = (c.want_touch & ~cm) | (c.want_never & ee[k]);
                    const uint32_t okm = (uint32_t)((int32_t)(bad - 1u) >> 31);          // all ones when bad == 0
                    okl[k] |= okm;
                    el[k] = min(el[k] + 1u, ELAPSED_MAX);
                    const uint32_t dn = ((c.max_steps - 1u - el[k]) >> 30) & 2u;          // el >= max_steps
                    rb[k] = RBITS_FAIL ^ (okm & 0x3F800000u);
                    fl[k] = (okl[k] & 1u) | (okm & 4u) | dn;
                }
            }
            st_vec<true, EPL>(&rrow[(int64_t)t * nvec], rb);
            uint32_t f = fl[0];
#pragma unroll
            for (int k = 1; k < EPL; k++) f |= fl[k] << (8 * k);
            st_1<true>(&frow[(int64_t)t * nvec], (Fl)f);
        };
        if (!BATCHLOOP) {
#pragma unroll 5
            for (int t = 0; t < T; t++) body(t);
        } else {
            const int nb = T / 5;
            for (int b = 0; b < nb; b++) {
#pragma unroll
                for (int u = 0; u < 5; u++) body(b * 5 + u);
            }
            for (int t = nb * 5; t < T; t++) body(t);
        }
        if (UNPACKED) {
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                an[k] |= el[k] << ELAPSED_SHIFT;
                ee[k] = (ee[k] & ~SUCC_BIT) | (okl[k] & SUCC_BIT);
            }
        }
        reinterpret_cast<Vec *>(now)[v] = pack<EPL>(an);
        reinterpret_cast<Vec *>(ever)[v] = pack<EPL>(ee);
    }
}

int main(int argc, char **argv)
{
    const int64_t B = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int T = 50, REPS = 30, NB = 4;
    uint32_t *now, *ever, *reward[NB]; uint8_t *flags[NB];
    CK(hipMalloc(&now, B * 4)); CK(hipMalloc(&ever, B * 4));
    for (int k = 0; k < NB; k++) { CK(hipMalloc(&reward[k], (size_t)B * T * 4)); CK(hipMalloc(&flags[k], (size_t)B * T)); }
    StepConst c{0x1FFFFF, 1u << 12, 0, 50};
    uint32_t *d_counter; CK(hipMalloc(&d_counter, 8)); CK(hipMemset(d_counter, 0, 8));
    unsigned long long *cnt; CK(hipMalloc(&cnt, 16)); CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, now, ever, B, 6u);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)B * (T * 5 + 16);
    int rot = 0;
    printf("== synthetic-source rollout, B = %lld, T = %d, bytes/launch = %.1f MB\n", (long long)B, T, bytes / 1e6);
#define TIME(name, launch)                                                                          \
    do {                                                                                            \
        for (int r = 0; r < 3; r++) { rot++; launch; }                                              \
        CK(hipEventRecord(e0, 0));                                                                  \
        for (int r = 0; r < REPS; r++) { rot++; launch; }                                           \
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));                                     \
        float tot; CK(hipEventElapsedTime(&tot, e0, e1));                                           \
        const double t = tot / REPS;                                                                \
        printf("%-52s %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, t * 1e3, bytes / t / 1e6, bytes / t / 1e6 / 80.0); \
    } while (0)
#define RUNX(GEN, EPL, DEP, BL, BLOCK, name)                                                                  \
    do {                                                                                                       \
        int64_t grid = (B / EPL + BLOCK - 1) / BLOCK; const int64_t cap = 256 * (1024 / BLOCK);                \
        if (grid > cap) grid = cap;                                                                            \
        TIME(name, hipLaunchKernelGGL((v_synth<GEN, EPL, true, DEP, BL>), dim3((unsigned)grid), dim3(BLOCK), 0, 0, now, ever, \
                                      reward[rot % NB], flags[rot % NB], B, T, c, 12345u));                    \
    } while (0)
#define RUN(GEN, EPL, UNP, BLOCK, name)                                                                       \
    do {                                                                                                       \
        int64_t grid = (B / EPL + BLOCK - 1) / BLOCK; const int64_t cap = 256 * (1024 / BLOCK);                \
        if (grid > cap) grid = cap;                                                                            \
        TIME(name, hipLaunchKernelGGL((v_synth<GEN, EPL, UNP>), dim3((unsigned)grid), dim3(BLOCK), 0, 0, now, ever, \
                                      reward[rot % NB], flags[rot % NB], B, T, c, 12345u));                    \
    } while (0)
    for (int rep = 0; rep < 2; rep++) {
        RUN(GenZero, 2, false, 1024, "no generator            epl2 b1024 packed");
        RUN(GenZero, 2, true, 1024, "no generator            epl2 b1024 unpacked");
        RUN(GenZero, 4, false, 1024, "no generator            epl4 b1024 packed");
        RUN(GenCur, 2, false, 1024, "round-1 4x lowbias32    epl2 b1024 packed");
        RUN(GenCur, 4, false, 1024, "round-1 4x lowbias32    epl4 b1024 packed");
        RUN(Gen2<0>, 2, false, 1024, "2x 2-round + tail 5/32  epl2 b1024 packed");
        RUN(Gen2<0>, 2, true, 1024, "2x 2-round + tail 5/32  epl2 b1024 unpacked");
        RUN(Gen2<0>, 4, true, 1024, "2x 2-round + tail 5/32  epl4 b1024 unpacked");
        RUN(Gen2<0>, 2, true, 256, "2x 2-round + tail 5/32  epl2 b256  unpacked");
        RUN(Gen2<0>, 1, true, 256, "2x 2-round + tail 5/32  epl1 b256  unpacked");
        RUN(GenLib, 2, true, 1024, "LIBRARY 2x 1-mul fold+tail epl2 b1024 unpacked");
        RUN(GenLib, 4, true, 1024, "LIBRARY 2x 1-mul fold+tail epl4 b1024 unpacked");
        RUN(GenLib, 1, true, 256, "LIBRARY 2x 1-mul fold+tail epl1 b256  unpacked");
        RUNX(GenLib, 2, true, false, 1024, "  same epl2 + state-dependent start");
        RUNX(GenLib, 2, false, true, 1024, "  same epl2 + batch loop");
        RUNX(GenLib, 2, true, true, 1024, "  same epl2 + state-dependent start + batch loop");
        RUNX(GenLib, 4, true, true, 1024, "  same epl4 + state-dependent start + batch loop");
        {
            SynthConst sc{}; sc.seed = 12345u; for (int n = 0; n < 8; n++) { sc.valid_by_n[n] = 0x1FFFFEu; sc.forced_by_n[n] = 0x700u; }
            ResetConst rc0{}; rc0.enabled = 1; rc0.is_variation = 1; rc0.base_nobj = 6; rc0.seed = 1; rc0.reset_counter = d_counter;
            TIME("LIBRARY k_rollout<SYNTH,2,out3> 256x1024 +count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 2, 3, true, 5, false>), dim3(256), dim3(1024), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc0, cnt, cnt + 1));
            ResetConst rc1 = rc0; rc1.enabled = 0;
            TIME("LIBRARY k_rollout<SYNTH,2,out3> no reset, no count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 2, 3, false, 5, false>), dim3(256), dim3(1024), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc1, (unsigned long long *)nullptr, (unsigned long long *)nullptr));
            TIME("LIBRARY k_rollout<SYNTH,2,out3> reset, no count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 2, 3, true, 5, false>), dim3(256), dim3(1024), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc0, (unsigned long long *)nullptr, (unsigned long long *)nullptr));
            TIME("LIBRARY k_rollout<SYNTH,2,out3> no reset, count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 2, 3, false, 5, false>), dim3(256), dim3(1024), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc1, cnt, cnt + 1));
            TIME("LIBRARY k_rollout<SYNTH,2,out3> 1024x256 +count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 2, 3, true, 5, false>), dim3(1024), dim3(256), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc0, cnt, cnt + 1));
            TIME("LIBRARY k_rollout<SYNTH,4,out3> 256x1024 +count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 4, 3, true, 5, false>), dim3(256), dim3(1024), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc0, cnt, cnt + 1));
            TIME("LIBRARY k_rollout<SYNTH,1,out3> 1024x256 +count", hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, 1, 3, true, 5, false>), dim3(1024), dim3(256), 0, 0, now, ever, (const uint32_t *)nullptr, reward[rot % NB], flags[rot % NB], B, T, B, B, c, sc, rc0, cnt, cnt + 1));
        }
        RUN(Gen2<1>, 2, true, 1024, "2x 2-round, p = 1/8     epl2 b1024 unpacked");
        RUN(Gen2<2>, 2, true, 1024, "2x 1-round + tail 5/32  epl2 b1024 unpacked");
        RUN(Gen2<3>, 2, true, 1024, "2x mum-fold x2 + tail   epl2 b1024 unpacked");
        printf("--\n");
    }
    return 0;
}
