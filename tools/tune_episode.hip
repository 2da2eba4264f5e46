// Standalone harness (not part of the product): what bounds the episode writer (k_episode, 0.65 of the HBM peak
// against 0.78 for the other write-dominated kernels)?  Pure-write kernels with the writer's address pattern but no
// LDS reads and no expansion arithmetic, against the plain grid-stride fill, then the writer itself.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o /tmp/tune_episode tools/tune_episode.hip
//   ./tune_episode 1048576
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// (1) the plain fill: consecutive lanes, consecutive float4, grid-stride
__global__ void __launch_bounds__(256) w_fill(float4 *o, int64_t nq)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) o[q] = make_float4(1.f, 0.f, -1.f, 0.f);
}

// (2) every block (WAVES waves) owns `region_q` consecutive float4 and walks it front to back, lanes on consecutive
// float4 (the writer's pattern: region = 64 envs x (T+1) rows x 36 floats); lds_bytes of dynamic LDS only limit occupancy
template <bool NT>
__global__ void __launch_bounds__(1024) w_regions(float4 *o, int64_t region_q, int64_t nregions)
{
    extern __shared__ unsigned char smem_[];
    if (threadIdx.x == 0 && region_q < 0) smem_[0] = 1;      // keep the allocation
    for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        float4 *p = o + r * region_q;
        for (int64_t q = threadIdx.x; q < region_q; q += blockDim.x) {
            st_f4<NT>(&p[q], make_float4(1.f, 0.f, -1.f, 0.f));
        }
    }
}

// (3) same regions, but the block's waves are spread over `nsplit` row slabs of every region at once (wave w of a
// block works on slab w of the region): more open rows per block, shorter runs
__global__ void __launch_bounds__(1024) w_regions_slabs(float4 *o, int64_t region_q, int64_t nregions)
{
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    const int64_t slab = (region_q + nw - 1) / nw;
    for (int64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        float4 *p = o + r * region_q;
        const int64_t q1 = (w + 1) * slab < region_q ? (w + 1) * slab : region_q;
        for (int64_t q = w * slab + lane; q < q1; q += 64) p[q] = make_float4(1.f, 0.f, -1.f, 0.f);
    }
}

static hipEvent_t e0, e1;
#define TIME(name, bytes, launch)                                                                  \
    do {                                                                                           \
        for (int r = 0; r < 2; r++) { launch; }                                                    \
        CK(hipEventRecord(e0, 0));                                                                 \
        for (int r = 0; r < 5; r++) { launch; }                                                    \
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));                                    \
        float tot; CK(hipEventElapsedTime(&tot, e0, e1));                                          \
        const double ms = tot / 5;                                                                 \
        printf("%-74s %9.1f us  %7.1f GB/s  %5.3f of 8 TB/s\n", name, ms * 1e3, (bytes) / ms / 1e6, (bytes) / ms / 1e6 / 8000.0); \
        fflush(stdout);                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int64_t B = argc > 1 ? atoll(argv[1]) : 1048576;
    const int T = 50, DIMG = 36;
    const int64_t region_q = (int64_t)64 * (T + 1) * DIMG / 4, nregions = B / 64;
    const int64_t nq = region_q * nregions;
    const double bytes = (double)nq * 16;
    float4 *o; CK(hipMalloc(&o, (size_t)nq * 16));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("== %lld envs: %lld regions of %lld bytes, %.2f GB\n", (long long)B, (long long)nregions, (long long)region_q * 16, bytes / 1e9);
    char nm[200];
    for (int g : {2048, 8192, 65536}) {
        snprintf(nm, 200, "fill, grid-stride, %d blocks x 256", g);
        TIME(nm, bytes, hipLaunchKernelGGL(w_fill, dim3(g), dim3(256), 0, 0, o, nq));
    }
    // 9 KB per single-wave block (the k_get_ag pattern), 64 envs x 36 floats
    TIME("regions of 9216 B, one wave each, block order", bytes,
         hipLaunchKernelGGL(w_regions<false>, dim3((unsigned)(nq / 576)), dim3(64), 0, 0, o, (int64_t)576, nq / 576));
    for (int waves : {1, 2, 4, 8, 16})
        for (int lds : {0, 27 * 1024, 80 * 1024}) {
            if (lds == 80 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(w_regions<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            snprintf(nm, 200, "regions of 470016 B, %2d waves per region, %2d KB LDS per block", waves, lds / 1024);
            TIME(nm, bytes, hipLaunchKernelGGL(w_regions<false>, dim3((unsigned)nregions), dim3(64 * waves), lds, 0, o, region_q, nregions));
        }
    TIME("regions of 470016 B,  2 waves, 27 KB LDS, nontemporal stores", bytes,
         hipLaunchKernelGGL(w_regions<true>, dim3((unsigned)nregions), dim3(128), 27 * 1024, 0, o, region_q, nregions));
    for (int waves : {2, 4, 8})  {
        snprintf(nm, 200, "regions of 470016 B, %2d waves on %2d slabs of the region, 27 KB LDS", waves, waves);
        TIME(nm, bytes, hipLaunchKernelGGL(w_regions_slabs, dim3((unsigned)nregions), dim3(64 * waves), 27 * 1024, 0, o, region_q, nregions));
    }
    // persistent: as many blocks as fit, each walking regions r, r + grid, ...
    for (int per_cu : {1, 2, 4, 6}) {
        snprintf(nm, 200, "regions of 470016 B,  2 waves, persistent %d blocks per CU", per_cu);
        TIME(nm, bytes, hipLaunchKernelGGL(w_regions<false>, dim3(256 * per_cu), dim3(128), 27 * 1024, 0, o, region_q, nregions));
    }
    for (int per_cu : {1, 2}) {
        snprintf(nm, 200, "regions of 470016 B, 16 waves, persistent %d blocks per CU", per_cu);
        TIME(nm, bytes, hipLaunchKernelGGL(w_regions<false>, dim3(256 * per_cu), dim3(1024), 27 * 1024, 0, o, region_q, nregions));
    }

    // the writer itself: ag only (the same bytes as above + 4 B masks per env-step), synthetic contacts
    {
        uint32_t *now, *ever, *d_counter;
        CK(hipMalloc(&now, B * 4)); CK(hipMalloc(&ever, B * 4)); CK(hipMalloc(&d_counter, 8)); CK(hipMemset(d_counter, 0, 8));
        hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, now, ever, B, 6u);
        StepConst c{}; c.valid = 0x1FFFFF; c.want_touch = 1u << 12; c.max_steps = 50; c.sticky_on = 1;
        SynthConst s{}; s.seed0 = 1; for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; }
        ResetConst rc{}; rc.enabled = 1; rc.is_variation = 1; rc.base_nobj = 6; rc.seed0 = 1; rc.reset_counter = d_counter;
        PairTable tab{};
        for (int q = 0; q < 9; q++) tab.quad[q] = (uint32_t)((4 * q) % 21) | ((4 * q + 1) % 21) << 5 | ((4 * q + 2) % 21) << 10 | ((4 * q + 3) % 21) << 15;
        const int RC = T + 1, R = RC | 1;
        const size_t lds = (size_t)64 * R * sizeof(uint2) + sizeof(PairTable);
        CountArgs ca{};
        TIME("k_episode<6, synth, NONTEMPORAL stores>, ag only, 2 waves per block", bytes,
             hipLaunchKernelGGL((k_episode<6, SRC_SYNTH, true>), dim3((unsigned)nregions), dim3(128), lds, 0, now, ever,
                                (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, reinterpret_cast<float *>(o),
                                (float *)nullptr, (float *)nullptr, B, T, B, B, RC, R, c, s, tab, rc, ca, (uint64_t *)nullptr, (float *)nullptr));
        for (int waves : {1, 2}) {                  // the kernel's launch bound is 128 lanes
            snprintf(nm, 200, "k_episode<6, synth>, ag only, %d waves per block", waves);
            TIME(nm, bytes, hipLaunchKernelGGL((k_episode<6, SRC_SYNTH, false>), dim3((unsigned)nregions), dim3(64 * waves), lds, 0, now, ever,
                                               (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, reinterpret_cast<float *>(o),
                                               (float *)nullptr, (float *)nullptr, B, T, B, B, RC, R, c, s, tab, rc, ca, (uint64_t *)nullptr, (float *)nullptr));
        }
    }
    return 0;
}
