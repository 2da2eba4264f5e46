// Standalone tuning harness of round 3 (not part of the product): includes the library source and times
//   A  the 2-envs-per-lane pipelined shape with an honest register budget: rows in flight per lane (UNROLL),
//      against 1 env per lane, at the mid batch sizes;
//   B  the success count at small batches: none / one atomic per block on one word / sharded (the library's);
//   C  the policy-gradient path at 1 M envs: rollout + reward stream + return kernel (round 2) against rollout +
//      step-ok words + expansion kernel;
//   D  the return kernels on their own: the lane-per-env exact form (plain / general rewards), the scan, the expansion
//      from step-ok words (round 2's two-lanes-per-env kernel was timed beside them until it was removed: profiles/r03_tune_return_kernels.log).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/tune_round3 tools/tune_round3.hip
//   ./tools/tune_round3 A 131072 262144 | B 65536 | C 1048576 | D 1048576
#define BPG_TIMELINE 1       // lane 0 of every block timestamps a few points of k_rollout (mode T)
#define BPG_TILED_EXPERIMENT 1   // mode L: block-tiled streams when the row stride passed is smaller than the batch
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bufs {
    int64_t B, LD;
    int T, NB;
    uint32_t *now, *ever;
    std::vector<uint32_t *> masks, reward;
    std::vector<uint8_t *> flags;
    std::vector<double *> G;
    uint64_t *okbits;
    unsigned long long *cnt, *shards;
    uint32_t *d_counter;
    hipEvent_t e0, e1;
};

static int setup(Bufs &b, int64_t B, int T, bool withG)
{
    b.B = B; b.T = T;
    b.LD = bpg_row_stride(B) + (getenv("LDPAD") ? atoll(getenv("LDPAD")) : 0);
    const size_t set_bytes = (size_t)b.LD * T * 9 + (withG ? (size_t)B * T * 8 : 0);
    b.NB = (int)((size_t)(getenv("NBMB") ? atoll(getenv("NBMB")) : 640) * 1024 * 1024 / set_bytes) + 1;
    if (b.NB < 3) b.NB = 3;
    if (b.NB > 48) b.NB = 48;
    CK(hipMalloc(&b.now, B * 4)); CK(hipMalloc(&b.ever, B * 4)); CK(hipMalloc(&b.okbits, B * 8));
    b.masks.resize(b.NB); b.reward.resize(b.NB); b.flags.resize(b.NB); b.G.assign(b.NB, nullptr);
    for (int k = 0; k < b.NB; k++) {
        CK(hipMalloc(&b.masks[k], (size_t)b.LD * T * 4)); CK(hipMalloc(&b.reward[k], (size_t)b.LD * T * 4));
        CK(hipMalloc(&b.flags[k], (size_t)b.LD * T));
        if (withG) CK(hipMalloc(&b.G[k], (size_t)B * T * 8));
    }
    SynthConst s{}; s.seed0 = 1;
    for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; }
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, b.now, b.ever, B, 6u);
    for (int k = 0; k < b.NB; k++) {
        s.seed0 = 1 + k;
        for (int t = 0; t < T; t++)
            hipLaunchKernelGGL(k_synth_masks, dim3(2048), dim3(256), 0, 0, b.ever, B, t, 1, b.masks[k] + (int64_t)t * b.LD, b.LD, s);
    }
    CK(hipMalloc(&b.d_counter, 8)); CK(hipMemset(b.d_counter, 0, 8));
    CK(hipMalloc(&b.cnt, 16)); CK(hipMemset(b.cnt, 0, 16));
    CK(hipMalloc(&b.shards, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8)); CK(hipMemset(b.shards, 0, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8));
    CK(hipEventCreate(&b.e0)); CK(hipEventCreate(&b.e1));
    CK(hipDeviceSynchronize());
    return 0;
}

static void teardown(Bufs &b)
{
    (void)hipFree(b.now); (void)hipFree(b.ever); (void)hipFree(b.okbits);
    for (int k = 0; k < b.NB; k++) { (void)hipFree(b.masks[k]); (void)hipFree(b.reward[k]); (void)hipFree(b.flags[k]); if (b.G[k]) (void)hipFree(b.G[k]); }
    (void)hipFree(b.d_counter); (void)hipFree(b.cnt); (void)hipFree(b.shards);
}

static int rot = 0;
#define REPS 40
// per-launch time from an event pair around each launch (median) AND from a train of REPS launches
#define TIME(name, bytes, launch)                                                                  \
    do {                                                                                           \
        std::vector<float> ts;                                                                     \
        for (int r = 0; r < REPS + 3; r++) {                                                       \
            rot++; CK(hipEventRecord(b.e0, 0)); launch; CK(hipEventRecord(b.e1, 0));               \
            CK(hipEventSynchronize(b.e1));                                                         \
            float ms; CK(hipEventElapsedTime(&ms, b.e0, b.e1));                                    \
            if (r >= 3) ts.push_back(ms);                                                          \
        }                                                                                          \
        std::sort(ts.begin(), ts.end());                                                           \
        CK(hipEventRecord(b.e0, 0));                                                               \
        for (int r = 0; r < REPS; r++) { rot++; launch; }                                          \
        CK(hipEventRecord(b.e1, 0)); CK(hipEventSynchronize(b.e1));                                \
        float tot; CK(hipEventElapsedTime(&tot, b.e0, b.e1));                                      \
        const double med = ts[ts.size() / 2], train = tot / REPS;                                  \
        printf("%-58s %8.2f us (%4.1f%%)   train %8.2f us (%6.1f GB/s, %4.1f%% of 8 TB/s)\n", name, med * 1e3, \
               (bytes) / med / 1e6 / 80.0, train * 1e3, (bytes) / train / 1e6, (bytes) / train / 1e6 / 80.0); \
        fflush(stdout);                                                                            \
    } while (0)
#define M b.masks[rot % b.NB]
#define R b.reward[rot % b.NB]
#define F b.flags[rot % b.NB]
#define GG b.G[rot % b.NB]

static StepConst mk_c() { StepConst c{}; c.valid = 0x1FFFFF; c.want_touch = 1u << 12; c.want_never = 0; c.max_steps = 50; c.sticky_on = 1; return c; }
static SynthConst mk_s() { SynthConst s{}; s.seed0 = 1; for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; } return s; }
static ResetConst mk_rc(Bufs &b) { ResetConst rc{}; rc.enabled = 1; rc.is_variation = 1; rc.base_nobj = 6; rc.seed0 = 1; rc.reset_counter = b.d_counter; return rc; }
static PowTable mk_pw(int T) { PowTable p{}; for (int k = 0; k < T && k < 64; k++) p.v[k] = std::pow(0.98, (double)k); return p; }

// COUNT: 0 none, 1 one plain atomic per block, 2 sharded.  OKB: 1 = the launch also stores the step-ok words
#define ROLL(EPL, BLOCK, UNR, PIPE, OUT, COUNT, CAPMUL, OKB, LBV)                                              \
    do {                                                                                                       \
        const int64_t lanes = B / EPL;                                                                         \
        int64_t grid = (lanes + BLOCK - 1) / BLOCK;                                                            \
        const int64_t cap = (int64_t)CAPMUL * 256 * (1024 / BLOCK);                                            \
        if (cap && grid > cap) grid = cap;                                                                     \
        CountArgs ca{COUNT ? b.cnt : nullptr, COUNT ? b.cnt + 1 : nullptr, COUNT == 2 ? b.shards : nullptr};   \
        char nm[120];                                                                                          \
        snprintf(nm, 120, "epl%d block%-4d grid%-5lld unr%-2d %s out%d %s %s", EPL, BLOCK, (long long)grid, UNR, \
                 PIPE ? "pipe " : "plain", OUT, COUNT == 0 ? "nocount" : (COUNT == 1 ? "atomic " : "shards "),  \
                 OKB ? "+okbits" : "");                                                                        \
        const double bytes = (double)B * (T * (4 + ((OUT & 1) ? 4 : 0) + ((OUT & 2) ? 1 : 0)) + 12 + (OKB ? 8 : 0)); \
        TIME(nm, bytes, hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, EPL, OUT, true, UNR, PIPE, true, LBV, (OKB != 0)>), dim3((unsigned)grid), \
                                    dim3(BLOCK), 0, 0, b.now, b.ever, M, R, F, B, T, b.LD, b.LD, c, s, rc, ca,  \
                                    OKB ? b.okbits : nullptr, 1, (uint32_t *)nullptr));                                                \
    } while (0)

// The return kernels' traffic without their transposition: every lane widens float4 -> 4 doubles at the same linear
// index (4 B in, 8 B out per element), one block per tile of 256 lanes x U float4.
template <int U>
__global__ void __launch_bounds__(256) k_widen_linear(const float4 *__restrict__ in, double *__restrict__ out, int64_t nq)
{
    const int64_t q0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t q = q0 + u * 256; v[u] = in[q < nq ? q : nq - 1]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t q = q0 + u * 256;
        if (q < nq) {
            double2 *o = reinterpret_cast<double2 *>(out + 4 * q);
            o[0] = make_double2((double)v[u].x, (double)v[u].y);
            o[1] = make_double2((double)v[u].z, (double)v[u].w);
        }
    }
}

static int run(char mode, int64_t B)
{
    const int T = 50;
    Bufs b;
    if (setup(b, B, T, mode == 'C' || mode == 'D')) return 1;
    const StepConst c = mk_c(); const SynthConst s = mk_s(); const ResetConst rc = mk_rc(b); const PowTable pw = mk_pw(T);
    printf("== mode %c  B = %lld envs, T = %d, LD = %lld, %d rotating buffer sets\n", mode, (long long)B, T, (long long)b.LD, b.NB);
    if (mode == 'A') {
        for (int rep = 0; rep < 2; rep++) {
            ROLL(1, 256, 10, true, 3, 2, 1, 0, 256); ROLL(1, 256, 5, true, 3, 2, 1, 0, 256);
            ROLL(2, 256, 5, false, 3, 2, 1, 0, 256); ROLL(2, 256, 5, false, 3, 2, 0, 0, 256);
            ROLL(2, 1024, 5, false, 3, 2, 1, 0, 1024); ROLL(2, 512, 5, false, 3, 2, 1, 0, 1024);
            ROLL(2, 256, 10, true, 3, 2, 1, 0, 256);
        }
    } else if (mode == 'F') {                    // the bandwidth-bound sizes: capped (resident) vs uncapped grids, rows in flight
        for (int rep = 0; rep < 3; rep++) {
            ROLL(2, 1024, 5, false, 3, 2, 1, 0, 1024); ROLL(2, 1024, 5, false, 3, 2, 0, 0, 1024);
            ROLL(2, 1024, 10, false, 3, 2, 1, 0, 1024); ROLL(2, 1024, 10, false, 3, 2, 0, 0, 1024);
            ROLL(2, 512, 5, false, 3, 2, 1, 0, 1024); ROLL(2, 512, 5, false, 3, 2, 0, 0, 1024);
            ROLL(2, 256, 5, false, 3, 2, 1, 0, 256); ROLL(2, 256, 5, false, 3, 2, 0, 0, 256);
            // 4 envs per lane: ONE lane group per block at 1 M envs (no boundary in the middle of the launch)
            ROLL(4, 1024, 5, false, 3, 2, 1, 0, 1024); ROLL(4, 1024, 10, false, 3, 2, 1, 0, 1024);
        }
    } else if (mode == 'B') {
        TIME("empty kernel (launch floor)", 1.0, hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, 0, b.d_counter, 1u));
        for (int rep = 0; rep < 2; rep++) {
            ROLL(1, 256, 10, true, 3, 0, 1, 0, 256); ROLL(1, 256, 10, true, 3, 1, 1, 0, 256); ROLL(1, 256, 10, true, 3, 2, 1, 0, 256);
            ROLL(1, 128, 10, true, 3, 0, 1, 0, 256); ROLL(1, 128, 10, true, 3, 1, 1, 0, 256); ROLL(1, 128, 10, true, 3, 2, 1, 0, 256);
            ROLL(1, 64, 10, true, 3, 0, 1, 0, 256); ROLL(1, 64, 10, true, 3, 1, 1, 0, 256); ROLL(1, 64, 10, true, 3, 2, 1, 0, 256);
            ROLL(1, 256, 5, true, 3, 2, 1, 0, 256); ROLL(1, 128, 5, true, 3, 2, 1, 0, 256);
        }
    } else if (mode == 'C') {
        const double gbytes = (double)B * (8 + 8 * T), rbytes = (double)B * T * 12;
        constexpr int WPB = DR_LANE_WPB;
        const dim3 lg((unsigned)((B + 64 * WPB - 1) / (64 * WPB))), lb(64 * WPB);
        for (int rep = 0; rep < 2; rep++) {
            ROLL(2, 1024, 5, false, 3, 2, 1, 0, 1024);                       // round 2: reward + flags streams ...
            TIME("  + k_discounted_return_lane<50, exact>", rbytes,
                 hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, WPB>), lg, lb, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50));
            ROLL(2, 1024, 5, false, 2, 2, 1, 1, 1024);                       // flags + step-ok words ...
            ROLL(2, 1024, 5, false, 0, 2, 1, 1, 1024);                       // (G alone: no per-step stream at all)
            TIME("  + k_returns_from_okbits_lane<50>", gbytes,
                 hipLaunchKernelGGL((k_returns_from_okbits_lane<50, WPB>), lg, lb, 0, 0, pw, (const uint64_t *)b.okbits, GG, B));
        }
    } else if (mode == 'D') {
        const double rbytes = (double)B * T * 12;
        constexpr int WPB = DR_LANE_WPB;
        const dim3 lg((unsigned)((B + 64 * WPB - 1) / (64 * WPB))), lb(64 * WPB);
        // rewards of a real rollout in every buffer set
        for (int k = 0; k < b.NB; k++) { rot = k; ROLL(2, 1024, 5, false, 3, 2, 1, 0, 1024); }
        for (int rep = 0; rep < 3; rep++) {
            TIME("k_discounted_return_lane<50, exact> (plain rewards)", rbytes,
                 hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, WPB>), lg, lb, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50));
            TIME("k_discounted_return_lane<50, scan>", rbytes,
                 hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_SCAN, WPB>), lg, lb, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50));
            TIME("k_discounted_return_lane<50, exact> (masks as floats: general path)", rbytes,
                 hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, WPB>), lg, lb, 0, 0, pw, (const float *)M, B, b.LD, 0.98, GG, 1, 50));
            TIME("k_returns_from_okbits_lane<50>", (double)B * (8 + 8 * T),
                 hipLaunchKernelGGL((k_returns_from_okbits_lane<50, WPB>), lg, lb, 0, 0, pw, (const uint64_t *)b.okbits, GG, B));
            {   // the same bytes with no transposition (rows of the padded reward stream are read as one linear array)
                const int64_t nq = B * T / 4;
                TIME("ceiling: linear f32 -> f64 widening copy, 4 float4 per lane", rbytes,
                     hipLaunchKernelGGL(k_widen_linear<4>, dim3((unsigned)((nq + 1023) / 1024)), dim3(256), 0, 0, (const float4 *)R, GG, nq));
                TIME("ceiling: linear f32 -> f64 widening copy, 1 float4 per lane", rbytes,
                     hipLaunchKernelGGL(k_widen_linear<1>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, (const float4 *)R, GG, nq));
            }
#define WPBV(W)                                                                                                                    \
            do {                                                                                                                   \
                const dim3 g_((unsigned)((B + 64 * W - 1) / (64 * W))), b_(64 * W);                                                \
                TIME("  waves/block " #W ": lane<50, exact>", rbytes,                                                              \
                     hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, W>), g_, b_, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50)); \
                TIME("  waves/block " #W ": lane<50, scan>", rbytes,                                                               \
                     hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_SCAN, W>), g_, b_, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50));  \
                TIME("  waves/block " #W ": returns_from_okbits_lane<50>", (double)B * (8 + 8 * T),                                 \
                     hipLaunchKernelGGL((k_returns_from_okbits_lane<50, W>), g_, b_, 0, 0, pw, (const uint64_t *)b.okbits, GG, B)); \
            } while (0)
#define TRV(W, TRR)                                                                                                                \
            do {                                                                                                                   \
                const dim3 g_((unsigned)((B + 64 * W - 1) / (64 * W))), b_(64 * W);                                                \
                TIME("  waves/block " #W " tile rows " #TRR ": lane<50, exact>", rbytes,                                            \
                     hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_EXACT, W, TRR>), g_, b_, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50)); \
                TIME("  waves/block " #W " tile rows " #TRR ": lane<50, scan>", rbytes,                                             \
                     hipLaunchKernelGGL((k_discounted_return_lane<50, DRM_SCAN, W, TRR>), g_, b_, 0, 0, pw, (const float *)R, B, b.LD, 0.98, GG, 1, 50));  \
                TIME("  waves/block " #W " tile rows " #TRR ": returns_from_okbits_lane<50>", (double)B * (8 + 8 * T),               \
                     hipLaunchKernelGGL((k_returns_from_okbits_lane<50, W, TRR>), g_, b_, 0, 0, pw, (const uint64_t *)b.okbits, GG, B)); \
            } while (0)
            if (rep == 0) {
                // persistent forms of the expansion kernel: a capped grid, every wave loops over tiles
                for (int cap : {256 * 4, 256 * 3, 256 * 2, 256 * 8}) {
                    char nm[96]; snprintf(nm, 96, "  returns_from_okbits_lane<50>, grid %d (4 waves/block)", cap);
                    TIME(nm, (double)B * (8 + 8 * T), hipLaunchKernelGGL((k_returns_from_okbits_lane<50, 4>), dim3(cap), dim3(256), 0, 0, pw, (const uint64_t *)b.okbits, GG, B));
                }
                for (int cap : {256 * 8, 256 * 4}) {
                    char nm[96]; snprintf(nm, 96, "  returns_from_okbits_lane<50>, grid %d (2 waves/block)", cap);
                    TIME(nm, (double)B * (8 + 8 * T), hipLaunchKernelGGL((k_returns_from_okbits_lane<50, 2>), dim3(cap), dim3(128), 0, 0, pw, (const uint64_t *)b.okbits, GG, B));
                }
                for (int cap : {256 * 16, 256 * 8}) {
                    char nm[96]; snprintf(nm, 96, "  returns_from_okbits_lane<50>, grid %d (1 wave/block)", cap);
                    TIME(nm, (double)B * (8 + 8 * T), hipLaunchKernelGGL((k_returns_from_okbits_lane<50, 1>), dim3(cap), dim3(64), 0, 0, pw, (const uint64_t *)b.okbits, GG, B));
                }
            }
        }
    }
    teardown(b);
    return 0;
}

// E: where the three streams of a launch sit relative to each other.  Separate allocations of this size all start on a
// 2 MB boundary, so row t of the masks and row t of the rewards (same row stride) alias modulo 2 MB.  One arena per
// buffer set with the reward / flag streams skewed against the mask stream by skew_r / skew_f bytes; per-launch event
// times grouped by buffer set (min / mean / max over the sets show the placement-to-placement variance).
static int run_skew(int64_t B)
{
    const int T = 50, NSET = 6, REPSE = 8;
    const int64_t LD = bpg_row_stride(B);
    const size_t mbytes = (size_t)LD * T * 4, fbytes = (size_t)LD * T;
    uint32_t *now, *ever; CK(hipMalloc(&now, B * 4)); CK(hipMalloc(&ever, B * 4));
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, now, ever, B, 6u);
    uint32_t *d_counter; CK(hipMalloc(&d_counter, 8)); CK(hipMemset(d_counter, 0, 8));
    unsigned long long *cnt, *shards; CK(hipMalloc(&cnt, 16)); CK(hipMemset(cnt, 0, 16));
    CK(hipMalloc(&shards, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8)); CK(hipMemset(shards, 0, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const StepConst c = mk_c(); SynthConst s = mk_s();
    ResetConst rc{}; rc.enabled = 1; rc.is_variation = 1; rc.base_nobj = 6; rc.seed0 = 1; rc.reset_counter = d_counter;
    CountArgs ca{cnt, cnt + 1, shards};
    const double bytes = (double)B * (T * 9 + 12);
    printf("== mode E  B = %lld envs, T = %d, LD = %lld: stream placement inside one arena per buffer set (%d sets)\n", (long long)B, T, (long long)LD, NSET);
    const size_t skews[][2] = {{0, 0}, {1024, 2048}, {3072, 5120}, {65536 + 1024, 131072 + 3072}, {4096, 8192}, {1 << 20, 1 << 19},
                               {(1 << 20) + 1024, (1 << 19) + 3072}, {SIZE_MAX, SIZE_MAX}};
    for (auto &sk : skews) {
        const bool separate = sk[0] == SIZE_MAX;          // three allocations per set (what torch.empty gives)
        std::vector<char *> arena(NSET, nullptr);
        std::vector<uint32_t *> M_(NSET), R_(NSET); std::vector<uint8_t *> F_(NSET);
        for (int k = 0; k < NSET; k++) {
            if (separate) {
                CK(hipMalloc(&M_[k], mbytes)); CK(hipMalloc(&R_[k], mbytes)); CK(hipMalloc(&F_[k], fbytes));
            } else {
                CK(hipMalloc(&arena[k], 2 * mbytes + fbytes + (8 << 20)));
                M_[k] = (uint32_t *)arena[k];
                R_[k] = (uint32_t *)(arena[k] + mbytes + sk[0]);
                F_[k] = (uint8_t *)(arena[k] + 2 * mbytes + sk[0] + sk[1]);
            }
            s.seed0 = 1 + k;
            for (int t = 0; t < T; t++)
                hipLaunchKernelGGL(k_synth_masks, dim3(2048), dim3(256), 0, 0, ever, B, t, 1, M_[k] + (int64_t)t * LD, LD, s);
        }
        CK(hipDeviceSynchronize());
        std::vector<double> per_set(NSET, 0.0);
        for (int r = 0; r < REPSE + 2; r++)
            for (int k = 0; k < NSET; k++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, true, 1024, false>), dim3(256), dim3(1024), 0, 0,
                                   now, ever, M_[k], R_[k], F_[k], B, T, LD, LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) per_set[k] += ms / REPSE;
            }
        // and the launch train over all sets (no event between launches)
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; r++)
            for (int k = 0; k < NSET; k++)
                hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, true, 1024, false>), dim3(256), dim3(1024), 0, 0,
                                   now, ever, M_[k], R_[k], F_[k], B, T, LD, LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float tot; CK(hipEventElapsedTime(&tot, e0, e1));
        double mn = 1e9, mx = 0, mean = 0;
        for (double v : per_set) { mn = std::min(mn, v); mx = std::max(mx, v); mean += v / NSET; }
        if (separate) printf("separate allocations            ");
        else printf("skew reward %8zu flags %8zu", sk[0], sk[1]);
        printf("  per set (event pairs): min %6.2f mean %6.2f max %6.2f us   train %6.2f us (%4.1f%% of 8 TB/s)   sets:", mn * 1e3, mean * 1e3,
               mx * 1e3, tot / (10 * NSET) * 1e3, bytes / (tot / (10 * NSET)) / 1e6 / 80.0);
        for (double v : per_set) printf(" %.1f", v * 1e3);
        printf("\n"); fflush(stdout);
        for (int k = 0; k < NSET; k++) {
            if (separate) { (void)hipFree(M_[k]); (void)hipFree(R_[k]); (void)hipFree(F_[k]); }
            else (void)hipFree(arena[k]);
        }
    }
    return 0;
}

// mode T: where a single 1 M-env launch spends its time -- per-block timestamps (constant-rate clock, 10 ns) of: entry,
// first batch consumed, last step done, state stored (passes 0 and 1), before / after the count, stores drained.
static int run_timeline(int64_t B)
{
    Bufs b;
    const int T = 50;
    if (setup(b, B, T, false)) return 1;
    const StepConst c = mk_c();
    const SynthConst s = mk_s();
    const ResetConst rc = mk_rc(b);
    CountArgs ca{b.cnt, nullptr, b.shards};
    const int grid = 256;
    unsigned long long *tl;
    CK(hipMalloc(&tl, (size_t)grid * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_timeline), &tl, sizeof(tl)));
    std::vector<unsigned long long> h((size_t)grid * 16);
    printf("== mode T  B = %lld envs, grid %d x 1024, epl2 plain unr5: per-block times in us relative to the earliest entry\n", (long long)B, grid);
    for (int rep = 0; rep < 8; rep++) {
        const bool count = rep < 5;
        ca.count = count ? b.cnt : nullptr;
        CK(hipMemset(tl, 0, (size_t)grid * 16 * 8));
        rot++;
        CK(hipEventRecord(b.e0, 0));
        hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, true, 1024, false, false>), dim3(grid), dim3(1024), 0, 0,
                           b.now, b.ever, M, R, F, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
        CK(hipEventRecord(b.e1, 0)); CK(hipEventSynchronize(b.e1));
        float ms; CK(hipEventElapsedTime(&ms, b.e0, b.e1));
        CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        if (rep < 3) continue;
        unsigned long long t0 = ~0ull;
        for (int g = 0; g < grid; g++) t0 = std::min(t0, h[(size_t)g * 16]);
        const char *names[14] = {"entry", "pass0 first batch", "pass0 last step", "pass0 state stored", "", "pass1 first batch",
                                 "pass1 last step", "pass1 state stored", "", "", "", "before count", "after count", "stores drained"};
        printf("launch %d (%s): %.2f us by events\n", rep, count ? "with the success count" : "no count", ms * 1e3);
        for (int slot : {0, 1, 2, 3, 5, 6, 7, 11, 12, 13}) {
            std::vector<double> v;
            for (int g = 0; g < grid; g++) v.push_back((double)(h[(size_t)g * 16 + slot] - t0) / 100.0);   // 100 MHz
            std::sort(v.begin(), v.end());
            printf("  %-20s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f\n", names[slot], v.front(), v[grid / 10], v[grid / 2],
                   v[grid * 9 / 10], v.back());
        }
    }
    teardown(b);
    return 0;
}

// mode S: the same stamps for the SMALL-batch shape (1 env per lane, 256-lane blocks, pipelined requests, 10 rows in
// flight, one block per 256 envs): where a 65 536-env launch spends its 9-10 us.
static int run_timeline_small(int64_t B)
{
    Bufs b;
    const int T = 50;
    if (setup(b, B, T, false)) return 1;
    const StepConst c = mk_c();
    const SynthConst s = mk_s();
    const ResetConst rc = mk_rc(b);
    CountArgs ca{b.cnt, nullptr, b.shards};
    const int grid = (int)((B + 255) / 256);
    unsigned long long *tl;
    CK(hipMalloc(&tl, (size_t)grid * 16 * 8));
    std::vector<unsigned long long> h((size_t)grid * 16);
    unsigned long long *null_tl = nullptr;
    printf("== mode S  B = %lld envs, grid %d x 256, epl1 pipelined unr10: per-block times in us relative to the earliest entry\n", (long long)B, grid);
    for (int rep = 0; rep < 9; rep++) {
        const bool count = rep < 6;
        const bool stamps = rep % 3 != 2;           // every third launch runs without stamps: what the stamps cost
        ca.count = count ? b.cnt : nullptr;
        CK(hipMemset(tl, 0, (size_t)grid * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_timeline), stamps ? &tl : &null_tl, sizeof(tl)));
        if (!getenv("NOROT")) rot++;                // NOROT=1: every launch on the same buffers (Infinity-Cache-resident inputs)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(b.e0, 0));
        hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 1, 3, true, 10, true, true, 256, false, false, false>), dim3(grid), dim3(256), 0, 0,
                           b.now, b.ever, M, R, F, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
        CK(hipEventRecord(b.e1, 0)); CK(hipEventSynchronize(b.e1));
        float ms; CK(hipEventElapsedTime(&ms, b.e0, b.e1));
        CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        printf("launch %d (%s, %s): %.2f us by events (a lone launch between two events: + the events' own ~3 us)\n", rep,
               count ? "with the success count" : "no count", stamps ? "stamped" : "NOT stamped", ms * 1e3);
        if (!stamps || rep < 3) continue;
        unsigned long long t0 = ~0ull;
        for (int g = 0; g < grid; g++) t0 = std::min(t0, h[(size_t)g * 16]);
        const char *names[14] = {"entry", "first 10 steps done", "last step done", "state stored", "", "", "", "", "", "", "", "before count", "after count", "stores drained"};
        for (int slot : {0, 1, 2, 3, 11, 12, 13}) {
            std::vector<double> v;
            for (int g = 0; g < grid; g++) v.push_back((double)(h[(size_t)g * 16 + slot] - t0) / 100.0);   // 100 MHz
            std::sort(v.begin(), v.end());
            printf("  %-20s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f\n", names[slot], v.front(), v[grid / 10], v[grid / 2],
                   v[grid * 9 / 10], v.back());
        }
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_timeline), &null_tl, sizeof(null_tl)));
    teardown(b);
    return 0;
}

// mode N: nontemporal against plain mask LOADS (and reward / flag STORES) on the SAME buffer sets of one process -- the
// placement of the read stream moves a launch by up to 10 % (profiles/r04_stream_placement.log), so variants are only
// comparable on the same arrays.
static int run_nt_on_same_sets(int64_t B)
{
    Bufs b;
    const int T = 50;
    if (setup(b, B, T, false)) return 1;
    const StepConst c = mk_c();
    const SynthConst s = mk_s();
    const ResetConst rc = mk_rc(b);
    CountArgs ca{b.cnt, b.cnt + 1, b.shards};
    const int grid = 256;
    printf("== mode N  B = %lld envs, grid %d x 1024, epl2 plain unr5, %d buffer sets: us per launch on ONE set (re-launched; 472 MB per set)\n",
           (long long)B, grid, b.NB);
    printf("%-6s %12s %12s %12s %12s\n", "set", "nt ld + nt st", "plain ld", "plain st", "plain both");
    for (int k = -1; k < b.NB && k < 6; k++) {      // k = -1: all sets round-robin (nothing comes from the Infinity Cache)
        float us[4];
        for (int v = 0; v < 4; v++) {
            int it = 0;
            auto launch = [&]() {
                const int kk = k < 0 ? (it++ % b.NB) : k;
                uint32_t *m = b.masks[kk], *r = b.reward[kk];
                uint8_t *f = b.flags[kk];
                if (v == 0) hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, true, 1024, false, false, false>), dim3(grid), dim3(1024), 0, 0,
                                               b.now, b.ever, m, r, f, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
                if (v == 1) hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, false, 1024, false, false, false>), dim3(grid), dim3(1024), 0, 0,
                                               b.now, b.ever, m, r, f, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
                if (v == 2) hipLaunchKernelGGL((k_rollout<SRC_MASKS, false, 2, 3, true, 5, false, true, 1024, false, false, false>), dim3(grid), dim3(1024), 0, 0,
                                               b.now, b.ever, m, r, f, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
                if (v == 3) hipLaunchKernelGGL((k_rollout<SRC_MASKS, false, 2, 3, true, 5, false, false, 1024, false, false, false>), dim3(grid), dim3(1024), 0, 0,
                                               b.now, b.ever, m, r, f, B, T, b.LD, b.LD, c, s, rc, ca, (uint64_t *)nullptr, 1, (uint32_t *)nullptr);
            };
            for (int i = 0; i < 9; i++) launch();
            CK(hipEventRecord(b.e0, 0));
            for (int i = 0; i < 30; i++) launch();
            CK(hipEventRecord(b.e1, 0)); CK(hipEventSynchronize(b.e1));
            float ms; CK(hipEventElapsedTime(&ms, b.e0, b.e1));
            us[v] = ms * 1e3f / 30;
        }
        if (k < 0) printf("%-6s %12.2f %12.2f %12.2f %12.2f\n", "all", us[0], us[1], us[2], us[3]);
        else printf("%-6d %12.2f %12.2f %12.2f %12.2f\n", k, us[0], us[1], us[2], us[3]);
    }
    teardown(b);
    return 0;
}

// mode L: time-major rows against block-tiled streams [B / 2048][T][2048] on the SAME buffers (the bytes are the same;
// what differs is how many pages a CU touches: 50 rows 4 MiB apart per stream, or one 400 KB piece).
static int run_layout(int64_t B)
{
    Bufs b;
    const int T = 50;
    if (setup(b, B, T, false)) return 1;
    const StepConst c = mk_c();
    const SynthConst s = mk_s();
    const ResetConst rc = mk_rc(b);
    CountArgs ca{b.cnt, b.cnt + 1, b.shards};
    const int grid = 256;
    printf("== mode L  B = %lld envs, grid %d x 1024, epl2 plain unr5, %d buffer sets, all sets round-robin and each set on its own (re-launched)\n",
           (long long)B, grid, b.NB);
    printf("%-6s %16s %16s %16s\n", "set", "rows ld=B+256", "tiled 2048", "tiled 4096");
    for (int k = -1; k < b.NB && k < 6; k++) {
        float us[3];
        for (int v = 0; v < 3; v++) {
            const int64_t ld = v == 0 ? b.LD : (v == 1 ? 2048 : 4096);
            int it = 0;
            auto launch = [&]() {
                const int kk = k < 0 ? (it++ % b.NB) : k;
                hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 3, true, 5, false, true, 1024, false, false, false>), dim3(grid), dim3(1024), 0, 0,
                                   b.now, b.ever, b.masks[kk], b.reward[kk], b.flags[kk], B, T, ld, ld, c, s, rc, ca, (uint64_t *)nullptr, 1,
                                   (uint32_t *)nullptr);
            };
            for (int i = 0; i < 9; i++) launch();
            CK(hipEventRecord(b.e0, 0));
            for (int i = 0; i < 30; i++) launch();
            CK(hipEventRecord(b.e1, 0)); CK(hipEventSynchronize(b.e1));
            float ms; CK(hipEventElapsedTime(&ms, b.e0, b.e1));
            us[v] = ms * 1e3f / 30;
        }
        if (k < 0) printf("%-6s %16.2f %16.2f %16.2f\n", "all", us[0], us[1], us[2]);
        else printf("%-6d %16.2f %16.2f %16.2f\n", k, us[0], us[1], us[2]);
    }
    teardown(b);
    return 0;
}

// mode P: the policy-gradient call (rollout with flags + step-ok words, then the expansion of the words into G rows) as
// two launches one after the other, against the same work split into env halves on two streams: rollout A, then
// rollout B and the expansion of A side by side (a read-heavy and a write-only kernel share the chip), then the
// expansion of B.
static int run_pg_split(int64_t B)
{
    Bufs b;
    const int T = 50;
    if (setup(b, B, T, true)) return 1;
    const StepConst c = mk_c();
    const SynthConst s = mk_s();
    ResetConst rc = mk_rc(b);
    const PowTable pw = mk_pw(T);
    CountArgs ca{};
    hipStream_t s0, s1;
    hipEvent_t eA, eJ, t0, t1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&eA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eJ, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    constexpr int WPB = DR_LANE_WPB;
    printf("== mode P  B = %lld envs, T = %d\n", (long long)B, T);
    auto roll = [&](hipStream_t st, int64_t off, int64_t n) {
        const RolloutGeometry geo = rollout_geometry(n, b.LD, b.LD, M + off, nullptr, F + off);
        if (geo.epl == 2 && geo.block == 1024)
            hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 2, true, 5, false, true, 1024, true, false>), dim3(geo.grid), dim3(geo.block), 0, st,
                               b.now + off, b.ever + off, M + off, (uint32_t *)nullptr, F + off, n, T, b.LD, b.LD, c, s, rc, ca, b.okbits + off, 1, (uint32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 2, 2, true, 5, false, true, 256, true, false>), dim3(geo.grid), dim3(geo.block), 0, st,
                               b.now + off, b.ever + off, M + off, (uint32_t *)nullptr, F + off, n, T, b.LD, b.LD, c, s, rc, ca, b.okbits + off, 1, (uint32_t *)nullptr);
    };
    auto expand = [&](hipStream_t st, int64_t off, int64_t n) {
        hipLaunchKernelGGL((k_returns_from_okbits_lane<50, WPB>), dim3((unsigned)((n + 64 * WPB - 1) / (64 * WPB))), dim3(64 * WPB), 0, st, pw,
                           (const uint64_t *)(b.okbits + off), GG + off * T, n);
    };
    for (int rep = 0; rep < 4; rep++) {
        for (int variant = 0; variant < 4; variant++) {
            const int nparts = variant == 0 ? 1 : (variant == 1 ? 2 : (variant == 2 ? 4 : 2));
            const bool serial_halves = variant == 3;          // the halves one after the other on ONE stream (what the split alone costs)
            const int REP = 20;
            float best = 1e9f;
            for (int outer = 0; outer < 3; outer++) {
                CK(hipEventRecord(t0, s0));
                for (int r = 0; r < REP; r++) {
                    rot++;
                    if (nparts == 1) { roll(s0, 0, B); expand(s0, 0, B); continue; }
                    const int64_t part = B / nparts;
                    if (serial_halves) {
                        for (int p = 0; p < nparts; p++) roll(s0, p * part, part);
                        for (int p = 0; p < nparts; p++) expand(s0, p * part, part);
                        continue;
                    }
                    CK(hipEventRecord(eJ, s0)); CK(hipStreamWaitEvent(s1, eJ, 0));     // fork
                    for (int p = 0; p < nparts; p++) {
                        roll(s0, p * part, part);
                        if (p + 1 < nparts) {
                            CK(hipEventRecord(eA, s0)); CK(hipStreamWaitEvent(s1, eA, 0));
                            expand(s1, p * part, part);
                        } else {
                            expand(s0, p * part, part);
                        }
                    }
                    CK(hipEventRecord(eJ, s1)); CK(hipStreamWaitEvent(s0, eJ, 0));     // join
                }
                CK(hipEventRecord(t1, s0)); CK(hipEventSynchronize(t1));
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                best = std::min(best, ms / REP);
            }
            const char *nm[4] = {"one rollout launch + one expansion launch (the library)", "2 env halves, expansion of A beside rollout B (2 streams)",
                                 "4 env quarters, expansions beside the next rollouts (2 streams)", "2 env halves one after the other on one stream"};
            printf("%-66s %8.2f us per pg rollout (%5.1f GB/s)\n", nm[variant], best * 1e3, (double)B * (T * 5 + 28 + 8 + 8 * T) / best / 1e6);
        }
    }
    teardown(b);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && argv[1][0] == 'P') return run_pg_split(atoll(argv[2]));
    if (argc >= 3 && argv[1][0] == 'T') return run_timeline(atoll(argv[2]));
    if (argc >= 3 && argv[1][0] == 'S') return run_timeline_small(atoll(argv[2]));
    if (argc >= 3 && argv[1][0] == 'N') return run_nt_on_same_sets(atoll(argv[2]));
    if (argc >= 3 && argv[1][0] == 'L') return run_layout(atoll(argv[2]));
    if (argc < 3) { printf("usage: tune_round3 A|B|C|D|E|F|T envs...\n"); return 1; }
    if (argv[1][0] == 'E') return run_skew(atoll(argv[2]));
    for (int i = 2; i < argc; i++)
        if (run(argv[1][0], atoll(argv[i]))) return 1;
    return 0;
}
