// Standalone tuning harness (not part of the product): times variants of the
// fused rollout kernel against a pure streaming kernel of the same byte mix.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/tune tools/tune_rollout.hip && /tmp/tune
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ uint4 ld4(const uint4 *p)
{
    if (NT) return make_uint4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y),
                              __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w));
    return *p;
}
template <bool NT> __device__ __forceinline__ void st4(uint4 *p, uint4 v)
{
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else *p = v;
}
template <bool NT> __device__ __forceinline__ void st1(uint32_t *p, uint32_t v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int BLOCK, int UNROLL, bool NTL, bool NTS>
__global__ void __launch_bounds__(BLOCK)
v_rollout(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
          uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb, int64_t n, int T, StepConst c)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 a = reinterpret_cast<uint4 *>(now)[v];
        uint4 e = reinterpret_cast<uint4 *>(ever)[v];
        uint32_t an[4] = {a.x, a.y, a.z, a.w}, ee[4] = {e.x, e.y, e.z, e.w};
        const uint4 *mrow = reinterpret_cast<const uint4 *>(masks) + v;
        uint4 *rrow = reinterpret_cast<uint4 *>(reward_tb) + v;
        uint32_t *frow = reinterpret_cast<uint32_t *>(flags_tb) + v;
#pragma unroll UNROLL
        for (int t = 0; t < T; t++) {
            const uint4 m = ld4<NTL>(&mrow[(int64_t)t * nvec]);
            uint32_t mm[4] = {m.x, m.y, m.z, m.w}, rb[4], fl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) step_core(an[k], ee[k], mm[k], c, rb[k], fl[k]);
            st4<NTS>(&rrow[(int64_t)t * nvec], make_uint4(rb[0], rb[1], rb[2], rb[3]));
            st1<NTS>(&frow[(int64_t)t * nvec], fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24));
        }
        reinterpret_cast<uint4 *>(now)[v] = make_uint4(an[0], an[1], an[2], an[3]);
        reinterpret_cast<uint4 *>(ever)[v] = make_uint4(ee[0], ee[1], ee[2], ee[3]);
    }
}

// time-major rows with a padded row stride (ld4 in uint4 units): do the T rows in flight alias
// in the DRAM channel / bank mapping when the stride is a power of two?
template <int BLOCK, int UNROLL>
__global__ void __launch_bounds__(BLOCK)
v_rollout_ld(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
             uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb, int64_t n, int T, int64_t ld4,
             StepConst c)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 a = reinterpret_cast<uint4 *>(now)[v];
        uint4 e = reinterpret_cast<uint4 *>(ever)[v];
        uint32_t an[4] = {a.x, a.y, a.z, a.w}, ee[4] = {e.x, e.y, e.z, e.w};
        const uint4 *mrow = reinterpret_cast<const uint4 *>(masks) + v;
        uint4 *rrow = reinterpret_cast<uint4 *>(reward_tb) + v;
        uint32_t *frow = reinterpret_cast<uint32_t *>(flags_tb) + v;
#pragma unroll UNROLL
        for (int t = 0; t < T; t++) {
            const uint4 m = mrow[(int64_t)t * ld4];
            uint32_t mm[4] = {m.x, m.y, m.z, m.w}, rb[4], fl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) step_core(an[k], ee[k], mm[k], c, rb[k], fl[k]);
            st4<true>(&rrow[(int64_t)t * ld4], make_uint4(rb[0], rb[1], rb[2], rb[3]));
            st1<true>(&frow[(int64_t)t * ld4], fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24));
        }
        reinterpret_cast<uint4 *>(now)[v] = make_uint4(an[0], an[1], an[2], an[3]);
        reinterpret_cast<uint4 *>(ever)[v] = make_uint4(ee[0], ee[1], ee[2], ee[3]);
    }
}

// wave-tiled layout: masks / reward / flags stored as [tile][t][64 lanes] (uint4 per lane), so a
// wave streams ONE contiguous T KiB run per stream instead of T pieces 4 MiB apart
template <int BLOCK, bool NTS>
__global__ void __launch_bounds__(BLOCK)
v_rollout_tiled(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
                uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb, int64_t n, int T, StepConst c)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 a = reinterpret_cast<uint4 *>(now)[v];
        uint4 e = reinterpret_cast<uint4 *>(ever)[v];
        uint32_t an[4] = {a.x, a.y, a.z, a.w}, ee[4] = {e.x, e.y, e.z, e.w};
        const int64_t tile = v >> 6, lane = v & 63;
        const uint4 *mrow = reinterpret_cast<const uint4 *>(masks) + tile * T * 64 + lane;
        uint4 *rrow = reinterpret_cast<uint4 *>(reward_tb) + tile * T * 64 + lane;
        uint32_t *frow = reinterpret_cast<uint32_t *>(flags_tb) + tile * T * 64 + lane;
#pragma unroll 5
        for (int t = 0; t < T; t++) {
            const uint4 m = mrow[t * 64];
            uint32_t mm[4] = {m.x, m.y, m.z, m.w}, rb[4], fl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) step_core(an[k], ee[k], mm[k], c, rb[k], fl[k]);
            st4<NTS>(&rrow[t * 64], make_uint4(rb[0], rb[1], rb[2], rb[3]));
            st1<NTS>(&frow[t * 64], fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24));
        }
        reinterpret_cast<uint4 *>(now)[v] = make_uint4(an[0], an[1], an[2], an[3]);
        reinterpret_cast<uint4 *>(ever)[v] = make_uint4(ee[0], ee[1], ee[2], ee[3]);
    }
}

// same byte mix, no env logic: the streaming ceiling for this access pattern
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
v_stream(const uint32_t *__restrict__ masks, uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb,
         int64_t n, int T)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const uint4 *mrow = reinterpret_cast<const uint4 *>(masks) + v;
        uint4 *rrow = reinterpret_cast<uint4 *>(reward_tb) + v;
        uint32_t *frow = reinterpret_cast<uint32_t *>(flags_tb) + v;
#pragma unroll 5
        for (int t = 0; t < T; t++) {
            const uint4 m = mrow[(int64_t)t * nvec];
            rrow[(int64_t)t * nvec] = m;
            frow[(int64_t)t * nvec] = m.x ^ m.w;
        }
    }
}

// plain float4 copy: the guide's "6.29 TB/s measured" reference point on this box
__global__ void __launch_bounds__(256) v_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) out[v] = in[v];
}

template <bool NT>
__global__ void __launch_bounds__(256) v_fill(float4 *__restrict__ out, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride)
        st_f4<NT>(&out[v], make_float4(1.f, 0.f, -1.f, (float)(v & 1)));
}

// a float4 fill from one-wave blocks that each reserve `lds` bytes of LDS (never touched): how
// much write bandwidth do 160 KB / lds waves per CU sustain?  (the episode writer holds 26 KB/wave)
__global__ void __launch_bounds__(64) v_fill_occ(float4 *__restrict__ out, int64_t nvec_per_block)
{
    extern __shared__ float4 dummy[];
    if (threadIdx.x == 999) dummy[0] = make_float4(0, 0, 0, 0);
    float4 *o = out + (int64_t)blockIdx.x * nvec_per_block;
    for (int64_t v = threadIdx.x; v < nvec_per_block; v += 64) o[v] = make_float4(1.f, 0.f, -1.f, (float)(v & 1));
}

int main(int argc, char **argv)
{
    const int64_t B = argc > 1 ? atoll(argv[1]) : (1 << 20);
    if (B % 4096 != 0 || B < 4096) { printf("B must be a multiple of 4096\n"); return 1; }
    const int T = 50, REPS = 30;
    constexpr int NB = 4;
    uint32_t *now, *ever, *masks_[NB], *reward_[NB]; uint8_t *flags_[NB];
    CK(hipMalloc(&now, B * 4)); CK(hipMalloc(&ever, B * 4));
    for (int k = 0; k < NB; k++) {
        CK(hipMalloc(&masks_[k], (size_t)(B + 65536) * T * 4)); CK(hipMalloc(&reward_[k], (size_t)(B + 65536) * T * 4));
        CK(hipMalloc(&flags_[k], (size_t)(B + 65536) * T));
    }
    int rot = 0;
#define masks masks_[rot % NB]
#define reward reward_[rot % NB]
#define flags flags_[rot % NB]
    SynthConst s{}; s.seed = 1; s.env_base = 0;
    for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; }
    StepConst c{0x1FFFFF, 1u << 12, 0, 50};
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, now, ever, B, 6u);
    for (rot = 0; rot < NB; rot++) {
        s.seed = 1 + rot;
        hipLaunchKernelGGL(k_synth_masks, dim3(2048), dim3(256), 0, 0, ever, B, 0, T, masks, B, s);
    }
    CK(hipDeviceSynchronize());
    uint32_t *d_counter; CK(hipMalloc(&d_counter, 8)); CK(hipMemset(d_counter, 0, 8));   // ResetConst::reset_counter
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)B * (T * 9 + 16);
    auto report = [&](const char *name, double ms, double nbytes) {
        printf("%-44s %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms * 1e3, nbytes / ms / 1e6,
               nbytes / ms / 1e6 / 80.0);
    };
#define TIME(name, nbytes, launch)                                         \
    do {                                                                   \
        std::vector<float> ts;                                             \
        for (int r = 0; r < REPS + 3; r++) {                               \
            rot++; CK(hipEventRecord(e0, 0)); launch; CK(hipEventRecord(e1, 0)); \
            CK(hipEventSynchronize(e1));                                   \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));                \
            if (r >= 3) ts.push_back(ms);                                  \
        }                                                                  \
        std::sort(ts.begin(), ts.end());                                   \
        report(name, ts[ts.size() / 2], nbytes);                           \
    } while (0)
    const int64_t nvec = B / 4;
#define ROLL(BLOCK, UNROLL, NTL, NTS, GRID)                                                              \
    TIME("rollout B" #BLOCK " U" #UNROLL " ntl" #NTL " nts" #NTS " g" #GRID, bytes,                       \
         hipLaunchKernelGGL((v_rollout<BLOCK, UNROLL, NTL, NTS>), dim3(GRID), dim3(BLOCK), 0, 0, now, ever, \
                            masks, reward, flags, B, T, c))
    for (int rep = 0; rep < 2; rep++) {
        // masks and reward are both exactly B*T*4 bytes: copy one onto the other
        TIME("copy float4 (B*T*4 bytes)", 2.0 * (double)B * T * 4,
             hipLaunchKernelGGL(v_copy, dim3(2048), dim3(256), 0, 0, (const uint4 *)masks, (uint4 *)reward,
                                (int64_t)((size_t)B * T * 4 / 16)));
        TIME("stream same byte mix B256 g1024", (double)B * T * 9,
             hipLaunchKernelGGL(v_stream<256>, dim3(nvec / 256), dim3(256), 0, 0, masks, reward, flags, B, T));
        TIME("stream same byte mix B1024 g256", (double)B * T * 9,
             hipLaunchKernelGGL(v_stream<1024>, dim3(nvec / 1024), dim3(1024), 0, 0, masks, reward, flags, B, T));
        {
            ResetConst rc0{}; rc0.enabled = 1; rc0.is_variation = 1; rc0.base_nobj = 6; rc0.seed = 1; rc0.reset_counter = d_counter;
            TIME("LIB k_rollout<masks,nt,4> 256x1024", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(256), dim3(1024), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, (unsigned long long *)nullptr,
                                    (unsigned long long *)nullptr));
        }
        {
            ResetConst rc0{}; rc0.enabled = 1; rc0.is_variation = 1; rc0.base_nobj = 6; rc0.seed = 1; rc0.reset_counter = d_counter;
            static unsigned long long *cnt = nullptr;
            if (!cnt) { CK(hipMalloc(&cnt, 16)); CK(hipMemset(cnt, 0, 16)); }
            TIME("LIB +count 256 blocks x 1024", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(256), dim3(1024), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, cnt, cnt + 1));
            TIME("LIB +count 512 blocks x 512", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(512), dim3(512), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, cnt, cnt + 1));
            TIME("LIB +count 1024 blocks x 256", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(1024), dim3(256), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, cnt, cnt + 1));
            TIME("LIB nocount 1024 blocks x 256", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(1024), dim3(256), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, (unsigned long long *)nullptr,
                                    (unsigned long long *)nullptr));
            TIME("LIB +count 2048 blocks x 128", bytes,
                 hipLaunchKernelGGL((k_rollout<SRC_MASKS, true, 4, 3, true>), dim3(2048), dim3(128), 0, 0, now, ever, masks,
                                    reward, flags, B, T, B, B, c, s, rc0, cnt, cnt + 1));
        }
        ROLL(1024, 5, false, false, 256);
        ROLL(256, 5, false, false, 1024);
        ROLL(256, 5, false, false, 2048);
        ROLL(512, 5, false, false, 512);
        ROLL(256, 10, false, false, 1024);
        ROLL(256, 2, false, false, 1024);
        ROLL(256, 1, false, false, 1024);
        ROLL(256, 5, true, false, 1024);
        ROLL(256, 5, false, true, 1024);
        ROLL(256, 5, true, true, 1024);
        ROLL(1024, 5, true, true, 256);
        TIME("rollout TILED B256 nts g1024", bytes,
             hipLaunchKernelGGL((v_rollout_tiled<256, true>), dim3(1024), dim3(256), 0, 0, now, ever, masks, reward,
                                flags, B, T, c));
        TIME("rollout TILED B256 plain g1024", bytes,
             hipLaunchKernelGGL((v_rollout_tiled<256, false>), dim3(1024), dim3(256), 0, 0, now, ever, masks, reward,
                                flags, B, T, c));
        TIME("rollout TILED B1024 nts g256", bytes,
             hipLaunchKernelGGL((v_rollout_tiled<1024, true>), dim3(256), dim3(1024), 0, 0, now, ever, masks, reward,
                                flags, B, T, c));
        TIME("rollout TILED B64 nts g4096", bytes,
             hipLaunchKernelGGL((v_rollout_tiled<64, true>), dim3(4096), dim3(64), 0, 0, now, ever, masks, reward,
                                flags, B, T, c));
        ROLL(256, 25, false, true, 1024);
        ROLL(256, 50, false, true, 1024);
        ROLL(128, 50, false, true, 2048);
#define ROLLLD(PAD)                                                                                            \
    TIME("rollout ld = B/4 + " #PAD " uint4 (nts)", bytes,                                                      \
         hipLaunchKernelGGL((v_rollout_ld<256, 5>), dim3(1024), dim3(256), 0, 0, now, ever, masks, reward, flags, B, \
                            T, (int64_t)(B / 4 + PAD), c))
        ROLLLD(0); ROLLLD(16); ROLLLD(64); ROLLLD(256); ROLLLD(1024); ROLLLD(4096); ROLLLD(4112);
        ROLL(1024, 10, true, true, 256);
        printf("--\n");
    }
    // ---------------- episode writer ----------------
    {
        const int SIDE = 6, DIMG = 36;
        float *ag, *gg, *sc;
        const size_t n_ag = (size_t)B * (T + 1) * DIMG, n_g = (size_t)B * T * DIMG, n_s = (size_t)B * T;
        CK(hipMalloc(&ag, n_ag * 4)); CK(hipMalloc(&gg, n_g * 4)); CK(hipMalloc(&sc, n_s * 4));
        bpg_variant_info info; fill_variant_info(BPG_BLOCKS_TOUCH_VARIATION, &info);
        const PairTable tab = pair_table(info);
        const int RC = T + 1, R = RC | 1;              // one chunk, as bpg_rollout picks for T <= 63
        const size_t lds = (size_t)64 * R * sizeof(uint2) + sizeof(PairTable);
        ResetConst rc{}; rc.enabled = 1; rc.is_variation = 1; rc.base_nobj = 6; rc.seed = 1; rc.reset_counter = d_counter;
        const double ebytes = (double)B * (16 + T * 4) + 4.0 * (n_ag + n_g + n_s);
        TIME("fill float4 plain (ag buffer)", 4.0 * n_ag,
             hipLaunchKernelGGL(v_fill<false>, dim3(2048), dim3(256), 0, 0, (float4 *)ag, (int64_t)(n_ag / 4)));
        TIME("fill float4 nt    (ag buffer)", 4.0 * n_ag,
             hipLaunchKernelGGL(v_fill<true>, dim3(2048), dim3(256), 0, 0, (float4 *)ag, (int64_t)(n_ag / 4)));
        for (int ldsk : {1, 8, 13, 20, 26, 40}) {
            char nm[64]; snprintf(nm, 64, "fill, 1-wave blocks holding %d KB LDS", ldsk);
            const int64_t per_block = (int64_t)(n_ag / 4) / (B / 64);
            TIME(nm, 4.0 * n_ag, hipLaunchKernelGGL(v_fill_occ, dim3((unsigned)(B / 64)), dim3(64), ldsk * 1024, 0,
                                                   (float4 *)ag, per_block));
        }
        TIME("episode<6> plain stores", ebytes,
             hipLaunchKernelGGL((k_episode<SIDE, SRC_MASKS, false>), dim3((unsigned)(B / 64)), dim3(64), lds, 0, now,
                                ever, masks, (uint32_t *)nullptr, (uint8_t *)nullptr, ag, gg, sc, B, T, B, B, RC, R, c, s, tab,
                                rc, (unsigned long long *)nullptr, (unsigned long long *)nullptr));
        TIME("episode<6> nt stores", ebytes,
             hipLaunchKernelGGL((k_episode<SIDE, SRC_MASKS, true>), dim3((unsigned)(B / 64)), dim3(64), lds, 0, now,
                                ever, masks, (uint32_t *)nullptr, (uint8_t *)nullptr, ag, gg, sc, B, T, B, B, RC, R, c, s, tab,
                                rc, (unsigned long long *)nullptr, (unsigned long long *)nullptr));
        TIME("episode<6> plain, ag + success only", (double)B * (16 + T * 4) + 4.0 * (n_ag + n_s),
             hipLaunchKernelGGL((k_episode<SIDE, SRC_MASKS, false>), dim3((unsigned)(B / 64)), dim3(64), lds, 0, now,
                                ever, masks, (uint32_t *)nullptr, (uint8_t *)nullptr, ag, (float *)nullptr, sc, B, T, B, B, RC, R,
                                c, s, tab, rc, (unsigned long long *)nullptr, (unsigned long long *)nullptr));
        TIME("episode<6> nt, ag only", (double)B * (16 + T * 4) + 4.0 * n_ag,
             hipLaunchKernelGGL((k_episode<SIDE, SRC_MASKS, true>), dim3((unsigned)(B / 64)), dim3(64), lds, 0, now,
                                ever, masks, (uint32_t *)nullptr, (uint8_t *)nullptr, ag, (float *)nullptr,
                                (float *)nullptr, B, T, B, B, RC, R, c, s, tab, rc, (unsigned long long *)nullptr,
                                (unsigned long long *)nullptr));
        TIME("episode<6> nt, g only", (double)B * (16 + T * 4) + 4.0 * n_g,
             hipLaunchKernelGGL((k_episode<SIDE, SRC_MASKS, true>), dim3((unsigned)(B / 64)), dim3(64), lds, 0, now,
                                ever, masks, (uint32_t *)nullptr, (uint8_t *)nullptr, (float *)nullptr, gg,
                                (float *)nullptr, B, T, B, B, RC, R, c, s, tab, rc, (unsigned long long *)nullptr,
                                (unsigned long long *)nullptr));
    }
    return 0;
}
