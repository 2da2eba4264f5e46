// Standalone tuning harness (not part of the product): launch geometry of the fused rollout
// kernel (envs per lane x block size x grid) at the batch sizes of BASELINE.json's configs,
// with enough rotating buffer sets that every launch reads its masks from HBM, not from the
// 256 MiB Infinity Cache.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/tune_geometry tools/tune_geometry.hip
//   ./tools/tune_geometry 65536 524288 1048576
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(int64_t B)
{
    const int T = getenv("GEO_T") ? atoi(getenv("GEO_T")) : 50, REPS = 40;
    const int64_t LD = B + (getenv("LDPAD") ? atoll(getenv("LDPAD")) : 0);     // row stride of the (T, LD) streams
    const size_t set_bytes = (size_t)LD * T * 9;
    int NB = (int)((size_t)640 * 1024 * 1024 / set_bytes) + 1;      // > 2x the Infinity Cache in flight
    if (NB < 4) NB = 4;
    if (NB > 48) NB = 48;
    uint32_t *now, *ever;
    std::vector<uint32_t *> masks(NB), reward(NB);
    std::vector<uint8_t *> flags(NB);
    CK(hipMalloc(&now, B * 4)); CK(hipMalloc(&ever, B * 4));
    for (int k = 0; k < NB; k++) {
        CK(hipMalloc(&masks[k], (size_t)LD * T * 4)); CK(hipMalloc(&reward[k], (size_t)LD * T * 4));
        CK(hipMalloc(&flags[k], (size_t)LD * T));
    }
    SynthConst s{}; s.seed = 1; s.env_base = 0;
    for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; }
    StepConst c{0x1FFFFF, 1u << 12, 0, 50};
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, now, ever, B, 6u);
    for (int k = 0; k < NB; k++) {
        s.seed = 1 + k;
        for (int t = 0; t < T; t++)
            hipLaunchKernelGGL(k_synth_masks, dim3(2048), dim3(256), 0, 0, ever, B, t, 1, masks[k] + (int64_t)t * LD, LD, s);
    }
    CK(hipDeviceSynchronize());
    uint32_t *d_counter; CK(hipMalloc(&d_counter, 8)); CK(hipMemset(d_counter, 0, 8));
    unsigned long long *cnt; CK(hipMalloc(&cnt, 16)); CK(hipMemset(cnt, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ResetConst rc{}; rc.enabled = 1; rc.is_variation = 1; rc.base_nobj = 6; rc.seed = 1; rc.reset_counter = d_counter;
    const double bytes = (double)B * (T * 9 + 16);
    int rot = 0;
    printf("== LD = %lld ", (long long)LD);
    printf("== B = %lld envs, T = %d, %d rotating buffer sets (%.0f MB each), bytes/launch = %.1f MB\n", (long long)B, T,
           NB, set_bytes / 1e6, bytes / 1e6);
    // back-to-back launches on one stream; per-launch time from events around each launch AND from
    // the whole train (events around REPS launches) -- the second includes launch gaps
#define TIME(name, launch)                                                                         \
    do {                                                                                           \
        std::vector<float> ts;                                                                     \
        for (int r = 0; r < REPS + 3; r++) {                                                       \
            rot++; CK(hipEventRecord(e0, 0)); launch; CK(hipEventRecord(e1, 0));                   \
            CK(hipEventSynchronize(e1));                                                           \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                        \
            if (r >= 3) ts.push_back(ms);                                                          \
        }                                                                                          \
        std::sort(ts.begin(), ts.end());                                                           \
        CK(hipEventRecord(e0, 0));                                                                 \
        for (int r = 0; r < REPS; r++) { rot++; launch; }                                          \
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));                                    \
        float tot; CK(hipEventElapsedTime(&tot, e0, e1));                                          \
        const double med = ts[ts.size() / 2], train = tot / REPS;                                  \
        printf("%-40s %8.2f us (%6.1f GB/s, %4.1f%%)   train %8.2f us (%6.1f GB/s, %4.1f%%)\n", name, med * 1e3, \
               bytes / med / 1e6, bytes / med / 1e6 / 80.0, train * 1e3, bytes / train / 1e6,     \
               bytes / train / 1e6 / 80.0);                                                        \
    } while (0)
#define M masks[rot % NB]
#define R reward[rot % NB]
#define F flags[rot % NB]
#define GEO(EPL, BLOCK, UNR, PIPE, NTL, COUNT) GEOC(EPL, BLOCK, UNR, PIPE, NTL, COUNT, 1)
    // CAPMUL: the grid is capped at CAPMUL x (1024 lanes' worth of blocks per CU); 0 = one block per
    // tile of the batch, however many that is (blocks beyond the resident set start in index order
    // as earlier ones retire)
#define GEOC(EPL, BLOCK, UNR, PIPE, NTL, COUNT, CAPMUL)                                                        \
    do {                                                                                                       \
        const int64_t lanes = B / EPL;                                                                         \
        int64_t grid = (lanes + BLOCK - 1) / BLOCK;                                                            \
        const int64_t cap = (int64_t)CAPMUL * 256 * (1024 / BLOCK);                                            \
        if (cap && grid > cap) grid = cap;                                                                     \
        char nm[96];                                                                                           \
        snprintf(nm, 96, "epl%d block%-4d grid%-5lld unr%-2d %s %s %s", EPL, BLOCK, (long long)grid, UNR,      \
                 PIPE ? "pipe " : "plain", NTL ? "ntl" : "   ", COUNT ? "+count" : "      ");                  \
        TIME(nm, hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, EPL, 3, true, UNR, PIPE, NTL>), dim3((unsigned)grid), \
                                    dim3(BLOCK), 0, 0, now, ever, M, R, F, B, T, LD, LD, c, s, rc, COUNT ? cnt : nullptr, \
                                    COUNT ? cnt + 1 : nullptr));                                               \
    } while (0)
    TIME("empty kernel (launch floor)", hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, 0, d_counter));
#define GEOS(EPL, BLOCK, UNR, CAPMUL)                                                                          \
    do {                                                                                                       \
        const int64_t lanes = B / EPL;                                                                         \
        int64_t grid = (lanes + BLOCK - 1) / BLOCK;                                                            \
        const int64_t cap = (int64_t)CAPMUL * 256 * (1024 / BLOCK);                                            \
        if (cap && grid > cap) grid = cap;                                                                     \
        char nm[96];                                                                                           \
        snprintf(nm, 96, "SYNTH epl%d block%-4d grid%-5lld unr%-2d", EPL, BLOCK, (long long)grid, UNR);         \
        TIME(nm, hipLaunchKernelGGL((k_rollout<SRC_SYNTH, true, EPL, 3, true, UNR, false, true>), dim3((unsigned)grid), \
                                    dim3(BLOCK), 0, 0, now, ever, (const uint32_t *)nullptr, R, F, B, T, LD, LD, c, s, rc, cnt, \
                                    cnt + 1));                                                                 \
    } while (0)
    if (getenv("GEO_SHORT") && atoi(getenv("GEO_SHORT")) == 5) {       // rows in flight per lane at the fat shape
        for (int rep = 0; rep < 2; rep++) {
            GEOC(2, 1024, 5, false, true, 1, 1); GEOC(2, 1024, 2, false, true, 1, 1); GEOC(2, 1024, 3, false, true, 1, 1);
            GEOC(2, 1024, 10, false, true, 1, 1); GEOC(2, 512, 5, false, true, 1, 1); GEOC(2, 512, 10, false, true, 1, 1);
            GEOC(2, 1024, 5, false, false, 1, 1);
        }
    } else if (getenv("GEO_SHORT") && atoi(getenv("GEO_SHORT")) == 4) {       // small batches: more rows in flight per lane
        for (int rep = 0; rep < 2; rep++) {
            GEOC(1, 256, 10, true, true, 1, 1); GEOC(1, 256, 25, true, true, 1, 1); GEOC(1, 256, 50, false, true, 1, 1);
            GEOC(1, 128, 25, true, true, 1, 1); GEOC(1, 64, 25, true, true, 1, 1); GEOC(1, 64, 50, false, true, 1, 1);
            GEOC(2, 256, 10, true, true, 1, 1); GEOC(2, 256, 25, true, true, 1, 1); GEOC(2, 128, 25, true, true, 1, 1);
            GEOC(2, 64, 25, true, true, 1, 1);
        }
    } else if (getenv("GEO_SHORT") && atoi(getenv("GEO_SHORT")) == 3) {       // in-kernel contacts: write-only
        for (int rep = 0; rep < 2; rep++) {
            GEOS(2, 1024, 5, 1); GEOS(2, 1024, 5, 0); GEOS(2, 512, 5, 1); GEOS(2, 512, 5, 0);
            GEOS(2, 256, 5, 1); GEOS(2, 256, 5, 0); GEOS(1, 256, 5, 0); GEOS(1, 512, 5, 0); GEOS(1, 1024, 5, 0);
            GEOS(2, 256, 10, 0); GEOS(1, 256, 10, 0); GEOS(4, 256, 5, 0); GEOS(4, 1024, 5, 1);
        }
    } else if (getenv("GEO_SHORT") && atoi(getenv("GEO_SHORT")) == 2) {
        GEOC(2, 1024, 5, false, true, 1, 1);
        GEOC(1, 256, 10, true, true, 1, 0);
    } else if (getenv("GEO_SHORT")) {                // the candidates around the library's thresholds
        for (int rep = 0; rep < 2; rep++) {
            GEOC(2, 1024, 5, false, true, 1, 1);
            GEOC(2, 1024, 5, false, true, 1, 0);
            GEOC(2, 256, 10, true, true, 1, 1);
            GEOC(2, 256, 10, true, true, 1, 0);
            GEOC(1, 256, 10, true, true, 1, 1);
            GEOC(1, 256, 10, true, true, 1, 0);
        }
    } else
    for (int rep = 0; rep < 2; rep++) {
        GEO(4, 1024, 5, false, false, 1);
        GEO(4, 1024, 5, false, true, 1);
        GEO(2, 1024, 5, false, false, 1);
        GEO(2, 1024, 5, false, true, 1);
        GEO(2, 256, 10, true, false, 1);
        GEO(2, 256, 10, true, true, 1);
        GEO(1, 256, 10, true, false, 1);
        GEO(1, 256, 10, true, true, 1);
        GEOC(2, 256, 10, true, true, 1, 0);
        GEOC(2, 256, 10, true, true, 1, 2);
        GEOC(1, 256, 10, true, true, 1, 0);
        GEOC(1, 256, 10, true, true, 1, 2);
        GEOC(2, 256, 5, false, true, 1, 0);
        GEOC(1, 256, 5, false, true, 1, 0);
        GEOC(2, 1024, 5, false, true, 1, 0);
        GEOC(2, 512, 5, false, true, 1, 0);
        GEOC(2, 512, 5, false, true, 1, 1);
        GEOC(1, 512, 5, false, true, 1, 0);
        GEOC(2, 128, 10, true, true, 1, 0);
        GEOC(1, 128, 10, true, true, 1, 0);
        GEOC(2, 64, 10, true, true, 1, 0);
        GEOC(1, 64, 10, true, true, 1, 0);
    }
    // what the library picks
    {
        const RolloutGeometry g = rollout_geometry(B, LD, LD, masks[0], reward[0], flags[0]);
        printf("library picks epl%d block%d grid%d\n", g.epl, g.block, g.grid);
    }
    CK(hipFree(now)); CK(hipFree(ever));
    for (int k = 0; k < NB; k++) { CK(hipFree(masks[k])); CK(hipFree(reward[k])); CK(hipFree(flags[k])); }
    CK(hipFree(d_counter)); CK(hipFree(cnt));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return run(1 << 20);
    for (int i = 1; i < argc; i++)
        if (run(atoll(argv[i]))) return 1;
    return 0;
}
