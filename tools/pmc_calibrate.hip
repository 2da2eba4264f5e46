// Standalone calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths
// this library uses (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of a 16-B-per-lane read
// stream; other widths are uncalibrated).  Each kernel moves a KNOWN byte count over a buffer far
// larger than the 256 MiB Infinity Cache; tools/summarize_pmc.py divides known bytes by the counter.
//   hipcc -O3 --offload-arch=gfx950 -o tools/pmc_calibrate tools/pmc_calibrate.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -- ./tools/pmc_calibrate     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename V> __global__ void __launch_bounds__(256) cal_read(const V *__restrict__ in, int64_t n, uint32_t *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const V v = in[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void __launch_bounds__(256) cal_read_u8(const uint8_t *__restrict__ in, int64_t n, uint32_t *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
    if (acc == 0x12345u) *sink = acc;
}
template <typename V, bool NT> __global__ void __launch_bounds__(256) cal_write(V *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V v;
        uint8_t *b = reinterpret_cast<uint8_t *>(&v);
        for (unsigned k = 0; k < sizeof(V); k++) b[k] = (uint8_t)(i + k);
        if (NT) {
            if constexpr (sizeof(V) == 16) {
                uint4 u = *reinterpret_cast<uint4 *>(&v);
                uint4 *p = reinterpret_cast<uint4 *>(&out[i]);
                __builtin_nontemporal_store(u.x, &p->x); __builtin_nontemporal_store(u.y, &p->y);
                __builtin_nontemporal_store(u.z, &p->z); __builtin_nontemporal_store(u.w, &p->w);
            } else if constexpr (sizeof(V) == 8) {
                uint2 u = *reinterpret_cast<uint2 *>(&v);
                uint2 *p = reinterpret_cast<uint2 *>(&out[i]);
                __builtin_nontemporal_store(u.x, &p->x); __builtin_nontemporal_store(u.y, &p->y);
            } else {
                __builtin_nontemporal_store(v, &out[i]);
            }
        } else out[i] = v;
    }
}

int main()
{
    const size_t BYTES = (size_t)1 << 30;
    void *buf; uint32_t *sink;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, BYTES));
    CK(hipDeviceSynchronize());
    const dim3 g(4096), b(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(cal_read<uint4>, g, b, 0, 0, (const uint4 *)buf, (int64_t)(BYTES / 16), sink);
        hipLaunchKernelGGL(cal_read<uint2>, g, b, 0, 0, (const uint2 *)buf, (int64_t)(BYTES / 8), sink);
        hipLaunchKernelGGL(cal_read<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, (int64_t)(BYTES / 4), sink);
        hipLaunchKernelGGL(cal_read_u8, g, b, 0, 0, (const uint8_t *)buf, (int64_t)(BYTES / 4), sink);   // 256 MiB of bytes
        hipLaunchKernelGGL((cal_write<uint4, true>), g, b, 0, 0, (uint4 *)buf, (int64_t)(BYTES / 16));
        hipLaunchKernelGGL((cal_write<uint4, false>), g, b, 0, 0, (uint4 *)buf, (int64_t)(BYTES / 16));
        hipLaunchKernelGGL((cal_write<uint2, true>), g, b, 0, 0, (uint2 *)buf, (int64_t)(BYTES / 8));
        hipLaunchKernelGGL((cal_write<uint2, false>), g, b, 0, 0, (uint2 *)buf, (int64_t)(BYTES / 8));
        hipLaunchKernelGGL((cal_write<uint32_t, true>), g, b, 0, 0, (uint32_t *)buf, (int64_t)(BYTES / 4));
        hipLaunchKernelGGL((cal_write<uint16_t, true>), g, b, 0, 0, (uint16_t *)buf, (int64_t)(BYTES / 8));  // 256 MiB
        hipLaunchKernelGGL((cal_write<uint8_t, true>), g, b, 0, 0, (uint8_t *)buf, (int64_t)(BYTES / 8));    // 128 MiB
        hipLaunchKernelGGL((cal_write<double, false>), g, b, 0, 0, (double *)buf, (int64_t)(BYTES / 8));
        CK(hipDeviceSynchronize());
    }
    printf("known bytes: read uint4/uint2/u32 %zu, read u8 %zu, write uint4/uint2/u32/double %zu, write u16 %zu, write u8 %zu\n",
           BYTES, BYTES / 4, BYTES, BYTES / 4, BYTES / 8);
    return 0;
}
