// Streaming ceilings of the box: read-only, write-only, copy, at several shapes.
//   hipcc -O3 --offload-arch=gfx950 -o tools/stream_ceiling tools/stream_ceiling.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const uint4 *__restrict__ in, int64_t nvec, uint32_t *__restrict__ sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) x[u] = in[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    for (; v < nvec; v += stride) { uint4 x = in[v]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NT>
__global__ void __launch_bounds__(256) k_write(uint4 *__restrict__ out, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 x = make_uint4((uint32_t)v, 1, 2, 3);
        if (NT) {
            __builtin_nontemporal_store(x.x, &out[v].x); __builtin_nontemporal_store(x.y, &out[v].y);
            __builtin_nontemporal_store(x.z, &out[v].z); __builtin_nontemporal_store(x.w, &out[v].w);
        } else out[v] = x;
    }
}

template <int UNROLL>
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) x[u] = in[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) out[v + u * stride] = x[u];
    }
    for (; v < nvec; v += stride) out[v] = in[v];
}

// R vectors read per W vectors written (R:W = 4:5 is the fused rollout's 214:270 MB, 1:2 the
// discounted return's 210:419 MB), every lane's accesses contiguous across the wave
template <int R, int W>
__global__ void __launch_bounds__(256) k_mix(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t nunit)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nunit; v += stride) {
        uint4 x[R];
#pragma unroll
        for (int u = 0; u < R; u++) x[u] = in[v + u * nunit];
        uint4 acc = x[0];
#pragma unroll
        for (int u = 1; u < R; u++) { acc.x ^= x[u].x; acc.y ^= x[u].y; acc.z ^= x[u].z; acc.w ^= x[u].w; }
#pragma unroll
        for (int u = 0; u < W; u++) {
            uint4 y = make_uint4(acc.x + u, acc.y, acc.z, acc.w);
            __builtin_nontemporal_store(y.x, &out[v + u * nunit].x); __builtin_nontemporal_store(y.y, &out[v + u * nunit].y);
            __builtin_nontemporal_store(y.z, &out[v + u * nunit].z); __builtin_nontemporal_store(y.w, &out[v + u * nunit].w);
        }
    }
}

int main()
{
    const size_t BYTES = (size_t)1 << 30;             // 1 GiB per buffer, far beyond the 256 MiB LLC
    const int64_t nvec = BYTES / 16;
    uint4 *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, BYTES)); CK(hipMemset(b, 2, BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define TIME(name, nbytes, launch)                                              \
    do {                                                                        \
        std::vector<float> ts;                                                  \
        for (int r = 0; r < 13; r++) {                                          \
            CK(hipEventRecord(e0, 0)); launch; CK(hipEventRecord(e1, 0));      \
            CK(hipEventSynchronize(e1));                                        \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));                     \
            if (r >= 3) ts.push_back(ms);                                       \
        }                                                                       \
        std::sort(ts.begin(), ts.end());                                        \
        printf("%-40s %8.1f us  %7.1f GB/s\n", name, ts[5] * 1e3, (double)(nbytes) / ts[5] / 1e6); \
    } while (0)
    {
        // mixes of the path's kernels: nunit vectors per stream, R + W streams of 16 B per lane
        const int64_t nunit = nvec / 5;
        auto k45 = k_mix<4, 5>; auto k12 = k_mix<1, 2>; auto k15 = k_mix<1, 5>;
        for (int grid : {1024, 2048, 8192, (int)((nunit + 255) / 256)}) {
            char nm[64];
            snprintf(nm, 64, "mix 4r:5w (rollout) grid %d", grid);
            TIME(nm, 9 * nunit * 16, hipLaunchKernelGGL(k45, dim3(grid), dim3(256), 0, 0, a, b, nunit));
            snprintf(nm, 64, "mix 1r:2w (return)  grid %d", grid);
            TIME(nm, 3 * nunit * 16, hipLaunchKernelGGL(k12, dim3(grid), dim3(256), 0, 0, a, b, nunit));
            snprintf(nm, 64, "mix 1r:5w            grid %d", grid);
            TIME(nm, 6 * nunit * 16, hipLaunchKernelGGL(k15, dim3(grid), dim3(256), 0, 0, a, b, nunit));
        }
    }
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536, 262144}) {
        char nm[64];
        snprintf(nm, 64, "read  U1 grid %d", grid);  TIME(nm, BYTES, hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, a, nvec, sink));
        snprintf(nm, 64, "read  U4 grid %d", grid);  TIME(nm, BYTES, hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, nvec, sink));
        snprintf(nm, 64, "write plain grid %d", grid); TIME(nm, BYTES, hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, b, nvec));
        snprintf(nm, 64, "write nt    grid %d", grid); TIME(nm, BYTES, hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, b, nvec));
        snprintf(nm, 64, "copy  U1 grid %d", grid);  TIME(nm, 2 * BYTES, hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, nvec));
        snprintf(nm, 64, "copy  U4 grid %d", grid);  TIME(nm, 2 * BYTES, hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, nvec));
    }
    TIME("hipMemcpyAsync d2d", 2 * BYTES, CK(hipMemcpyAsync(b, a, BYTES, hipMemcpyDeviceToDevice, 0)));
    return 0;
}
