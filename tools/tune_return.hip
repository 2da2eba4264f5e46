// Standalone tuning harness (not part of the product): waves per block of the discounted-return
// kernel and the row stride of its (T, ld) reward stream, A/B on one box.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/tune_return tools/tune_return.hip
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int64_t B = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int T = 50, NB = 3;
    PowTable tab{};
    for (int k = 0; k < T; k++) tab.v[k] = std::pow(0.98, (double)k);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int64_t pad : {(int64_t)0, (int64_t)256}) {
        const int64_t LD = B + pad;
        std::vector<float *> r(NB); std::vector<double *> G(NB);
        for (int k = 0; k < NB; k++) {
            CK(hipMalloc(&r[k], (size_t)LD * T * 4)); CK(hipMalloc(&G[k], (size_t)B * T * 8));
            CK(hipMemset(r[k], 0x80 + k, (size_t)LD * T * 4));      // some negative float pattern
        }
        const double bytes = (double)B * T * 12;
        int rot = 0;
#define TIME(name, launch)                                                                         \
        do {                                                                                       \
            for (int w = 0; w < 3; w++) { rot++; launch; }                                         \
            CK(hipEventRecord(e0, 0));                                                             \
            for (int q = 0; q < 12; q++) { rot++; launch; }                                        \
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));                                \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 12;                              \
            printf("ld = B + %-4lld %-22s %8.2f us  %7.1f GB/s  %.3f of 8 TB/s\n", (long long)pad, name, ms * 1e3, \
                   bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);                                   \
        } while (0)
#define RUN(WPB)                                                                                   \
        TIME("waves/block " #WPB, hipLaunchKernelGGL((k_discounted_return_reg<50, WPB, false>),           \
             dim3((unsigned)((B + 32 * WPB - 1) / (32 * WPB))), dim3(64 * WPB), 0, 0, r[rot % NB], B, LD, tab, G[rot % NB]))
#define RUNS(WPB, W)                                                                               \
        TIME("staged, waves " #WPB " W " #W, hipLaunchKernelGGL((k_discounted_return_reg<50, WPB, true, W>), \
             dim3((unsigned)((B + 32 * WPB - 1) / (32 * WPB))), dim3(64 * WPB), 0, 0, r[rot % NB], B, LD, tab, G[rot % NB]))
        for (int rep = 0; rep < 2; rep++) { RUN(1); RUN(4); RUNS(4, 16); RUNS(4, 8); RUNS(4, 4); RUNS(2, 8); RUNS(8, 8); RUNS(1, 8); }
        for (int k = 0; k < NB; k++) { CK(hipFree(r[k])); CK(hipFree(G[k])); }
    }
    return 0;
}
