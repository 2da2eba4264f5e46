// Standalone microbenchmark (not part of the product): issue cost of the integer VALU
// instructions a counter-based generator can be built from, on gfx950.  One block per CU,
// k waves per SIMD, 8 independent chains per lane so the loop is throughput- not latency-bound.
// Prints shader cycles (s_memtime) per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/valu_rates tools/valu_rates.hip && ./tools/valu_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 2000;

#define CHAIN8(OP)                                                                                   \
    OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)

#define KERNEL(NAME, ASM1)                                                                           \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, unsigned long long *cyc, uint32_t c)  \
    {                                                                                                \
        uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,   \
                 x6 = x0 + 6, x7 = x0 + 7;                                                           \
        uint32_t y0 = 1, y1 = 2, y2 = 3, y3 = 4, y4 = 5, y5 = 6, y6 = 7, y7 = 8;                     \
        (void)y0; (void)y1; (void)y2; (void)y3; (void)y4; (void)y5; (void)y6; (void)y7;               \
        __syncthreads();                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int i = 0; i < ITERS; i++) {                                                            \
            ASM1(x0, y0) ASM1(x1, y1) ASM1(x2, y2) ASM1(x3, y3) ASM1(x4, y4) ASM1(x5, y5) ASM1(x6, y6) ASM1(x7, y7) \
            ASM1(x0, y0) ASM1(x1, y1) ASM1(x2, y2) ASM1(x3, y3) ASM1(x4, y4) ASM1(x5, y5) ASM1(x6, y6) ASM1(x7, y7) \
        }                                                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7; \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define A_ADD(x, y) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_XOR(x, y) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_SHRXOR(x, y) asm volatile("v_lshrrev_b32 %1, 15, %0\n v_xor_b32 %0, %0, %1" : "+v"(x), "+v"(y));
#define A_MULLO(x, y) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_MULHI(x, y) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_MUL24(x, y) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_MULHI24(x, y) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
#define A_MAD24(x, y) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(c));
#define A_ALIGNBIT(x, y) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
#define A_ANDOR(x, y) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(x) : "v"(c));
#define A_PERM(x, y) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "v"(c));
#define A_BFE(x, y) asm volatile("v_bfe_u32 %0, %0, 3, 21" : "+v"(x));
#define A_LSHLOR(x, y) asm volatile("v_lshl_or_b32 %0, %0, 11, %1" : "+v"(x) : "v"(c));
#define A_CNDMASK(x, y) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
#define A_CMP(x, y) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
#define A_XORSDWA(x, y) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x));

KERNEL(k_add, A_ADD)
KERNEL(k_xor, A_XOR)
KERNEL(k_shrxor, A_SHRXOR)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_mulhi24, A_MULHI24)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_andor, A_ANDOR)
KERNEL(k_perm, A_PERM)
KERNEL(k_bfe, A_BFE)
KERNEL(k_lshlor, A_LSHLOR)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp, A_CMP)
KERNEL(k_xorsdwa, A_XORSDWA)

// v_mad_u64_u32 needs register pairs: written with the 64-bit C++ form instead
__global__ void __launch_bounds__(1024) k_mad64(uint32_t *out, unsigned long long *cyc, uint32_t c)
{
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; i++) {
#define M64(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"((uint32_t)a), "v"(c) : "vcc");
        M64(a0) M64(a1) M64(a2) M64(a3) M64(a4) M64(a5) M64(a6) M64(a7)
        M64(a0) M64(a1) M64(a2) M64(a3) M64(a4) M64(a5) M64(a6) M64(a7)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 ^ a7) >> 32);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// f64: register pairs, written through the C++ operators (FP contraction off on the command line)
#define F64KERNEL(NAME, EXPR)                                                                        \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, unsigned long long *cyc, uint32_t c)  \
    {                                                                                                \
        double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        const double k = 1.0 + (double)c * 1e-12, m = 0.5;                                          \
        __syncthreads();                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int i = 0; i < ITERS; i++) {                                                            \
            EXPR(x0) EXPR(x1) EXPR(x2) EXPR(x3) EXPR(x4) EXPR(x5) EXPR(x6) EXPR(x7)                  \
            EXPR(x0) EXPR(x1) EXPR(x2) EXPR(x3) EXPR(x4) EXPR(x5) EXPR(x6) EXPR(x7)                  \
        }                                                                                            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7); \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }
#define D_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define D_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(m));
#define D_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(m));
F64KERNEL(k_dmul, D_MUL)
F64KERNEL(k_dadd, D_ADD)
F64KERNEL(k_dfma, D_FMA)

// the discounted-return inner pattern: 8 products by scalar (SGPR-pair) constants into their own
// registers, then 8 additions into 8 accumulators
__global__ void __launch_bounds__(1024) k_dpattern(uint32_t *out, unsigned long long *cyc, uint32_t c)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double p0, p1, p2, p3, p4, p5, p6, p7;
    const double r = 1.0 + threadIdx.x * 1e-9;
    double k = 1.0 + (double)c * 1e-12;
    k = __builtin_bit_cast(double, ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(unsigned long long, k) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(unsigned long long, k)));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; i++) {
#define PM(p) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "s"(k), "v"(r));
#define PA(a, p) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(p));
        PM(p0) PM(p1) PM(p2) PM(p3) PM(p4) PM(p5) PM(p6) PM(p7)
        PA(a0, p0) PA(a1, p1) PA(a2, p2) PA(a3, p3) PA(a4, p4) PA(a5, p5) PA(a6, p6) PA(a7, p7)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

int main()
{
    uint32_t *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 16 * 8));
    std::vector<unsigned long long> h(256 * 16);
    struct K { const char *name; void (*fn)(uint32_t *, unsigned long long *, uint32_t); int per; };
    const K ks[] = {{"v_add_u32", k_add, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshrrev+v_xor (pair)", k_shrxor, 1},
                    {"v_xor_b32_sdwa (x ^= x>>16)", k_xorsdwa, 1},
                    {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mad_u64_u32", k_mad64, 1},
                    {"v_mul_u32_u24", k_mul24, 1}, {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_mad_u32_u24", k_mad24, 1},
                    {"v_alignbit_b32", k_alignbit, 1}, {"v_and_or_b32", k_andor, 1}, {"v_perm_b32", k_perm, 1},
                    {"v_bfe_u32", k_bfe, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_cndmask_b32", k_cndmask, 1},
                    {"v_cmp_eq_u32", k_cmp, 1},
                    {"v_mul_f64", k_dmul, 1}, {"v_add_f64", k_dadd, 1}, {"v_fma_f64", k_dfma, 1},
                    {"8x v_mul_f64 v,s,v + 8x v_add_f64", k_dpattern, 1}};
    printf("%-30s %10s %10s %10s %10s   (shader cycles per wave-instruction per SIMD)\n", "instruction", "1 w/SIMD",
           "2 w/SIMD", "4 w/SIMD", "8 w/CU*");
    for (const K &k : ks) {
        printf("%-30s", k.name);
        for (int wps : {1, 2, 4}) {
            const int block = 256 * wps;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(block), 0, 0, out, cyc, 0x9E3779B9u);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), cyc, 256 * (block / 64) * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v(h.begin(), h.begin() + 256 * (block / 64));
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            // `wps` waves share the SIMD: each wave-instruction's issue share = elapsed / (instr per wave * wps)
            printf(" %10.2f", med / ((double)ITERS * 16) / wps);
        }
        printf("\n");
    }
    return 0;
}
