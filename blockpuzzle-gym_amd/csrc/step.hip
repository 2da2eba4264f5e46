// step.hip -- one env.step per launch (robot_env.py:57-69 after the physics): packed pair masks, geom-id contact
// lists (fetch_env.py:158-167), the synthetic source, and the rollout worker's step that also writes the float32
// achieved_goal row / is_success / action copy (rollout.py:121-131, :144-150).
#include "state.hip.h"

namespace bpgk {

// 4 envs per lane: 16-byte coalesced loads/stores of now / ever / masks / reward,
// one packed dword of 4 flag bytes.  Scalar tail for n % 4.
template <int SRC>
__global__ void __launch_bounds__(256)
k_step_vec4(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
            uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c, SynthConst s)
{
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = gtid; v < nvec; v += stride) {
        uint4 a = reinterpret_cast<uint4 *>(now)[v];
        uint4 e = reinterpret_cast<uint4 *>(ever)[v];
        uint4 m = make_uint4(0, 0, 0, 0);
        if (SRC == SRC_MASKS) m = ld_vec<true, 4>(reinterpret_cast<const uint4 *>(masks) + v);   // read once
        uint32_t an[4] = {a.x, a.y, a.z, a.w}, ee[4] = {e.x, e.y, e.z, e.w}, mm[4] = {m.x, m.y, m.z, m.w};
        uint32_t rb[4], fl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mk = fetch_mask<SRC>(mm[k], an[k], ee[k], (uint32_t)(4 * v + k), s);
            step_core(an[k], ee[k], mk, c, rb[k], fl[k]);
        }
        // (the state stays a plain store: nontemporal state stores change nothing here -- 5.1-5.4 us per 1 M-env launch
        // with the state cache-resident, 5.9-6.1 us HBM-cold, either way: the launch is its fixed cost, a load and a
        // store round trip between two launch boundaries; profiles/r04_ab_output_stores.log)
        reinterpret_cast<uint4 *>(now)[v] = make_uint4(an[0], an[1], an[2], an[3]);
        reinterpret_cast<uint4 *>(ever)[v] = make_uint4(ee[0], ee[1], ee[2], ee[3]);
        // write-once streams: nontemporal, so that they do not push the state (re-read by the next
        // step) out of the caches
        if (reward) st_u4<true>(reinterpret_cast<uint4 *>(reward) + v, make_uint4(rb[0], rb[1], rb[2], rb[3]));
        if (flags)
            st_1<true>(reinterpret_cast<uint32_t *>(flags) + v, fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24));
    }
    const int64_t tail0 = nvec << 2;
    if (gtid < n - tail0) {
        const int64_t i = tail0 + gtid;
        uint32_t a = now[i], e = ever[i], rb, fl;
        const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
        step_core(a, e, mk, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

// Unaligned-pointer fallback: one env per lane, 4-byte accesses.
template <int SRC>
__global__ void __launch_bounds__(256)
k_step_scalar(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
              uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c, SynthConst s)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t a = now[i], e = ever[i], rb, fl;
        const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
        step_core(a, e, mk, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

// Contacts as the reference reads them (fetch_env.py:158-167): per env a list
// of geom-id pairs.  One env per lane; the id2obj tables live in LDS.
__device__ __forceinline__ uint32_t pair_bit_of(const int8_t *tab, uint32_t pr)
{
    const uint32_t g1 = pr & 0xFFFFu, g2 = pr >> 16;          // geom1 | geom2 << 16
    if (g1 < BPG_MAX_GEOMS && g2 < BPG_MAX_GEOMS) {
        const int o1 = tab[g1], o2 = tab[g2];
        if (o1 >= 0 && o2 >= 0) {                              // both not None (:165)
            const int lo = o1 < o2 ? o1 : o2, hi = o1 < o2 ? o2 : o1;
            return 1u << (lo * BPG_MAX_OBJS - lo * (lo - 1) / 2 + (hi - lo));
        }
    }
    return 0u;
}

// The pair mask of one padded list row.  VW = 4: the row is walked in 16-byte steps -- every lane strides through a row
// of its own, so a wave's load touches 64 cache lines whatever its width, and it is the number of such loads, not the
// bytes, that bounds the walk (dword steps: 0.40 of the HBM peak at maxcon = 16; 16-byte steps:
// profiles/r04_contacts_to_masks.log).  Needs maxcon % 4 == 0 and a 16-byte aligned list.
template <int VW>
__device__ __forceinline__ uint32_t row_pair_mask(const int8_t *tab, const uint32_t *row, int nc)
{
    uint32_t m = 0;
    if (VW == 4) {
        const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
        for (int k = 0; k < nc; k += 4) {
            const uint4 v = row4[k >> 2];
            m |= pair_bit_of(tab, v.x);
            if (k + 1 < nc) m |= pair_bit_of(tab, v.y);
            if (k + 2 < nc) m |= pair_bit_of(tab, v.z);
            if (k + 3 < nc) m |= pair_bit_of(tab, v.w);
        }
    } else {
        for (int k = 0; k < nc; k++) m |= pair_bit_of(tab, row[k]);
    }
    return m;
}

// One env per lane; the id2obj tables live in LDS.  (Round 2 staged the block's 256 rows through LDS with coalesced
// 16-byte loads; the 16-byte row walk of row_pair_mask<4> is as fast or faster at every list length -- 1 M envs,
// maxcon 8 / 16 / 32: 14.6 / 19.6 / 28.4 us against 16.2 / 19.9 / 33.9 staged and 20.1 / 29.1 / 50.0 with dword steps,
// profiles/r04_contacts_to_masks.log -- and needs no tile.)
template <int VW>
__global__ void __launch_bounds__(256)
k_step_contacts(uint32_t *__restrict__ now, uint32_t *__restrict__ ever,
                const int16_t *__restrict__ contacts, const int32_t *__restrict__ ncon, int maxcon,
                const int8_t *__restrict__ id2obj, int is_variation,
                uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, int64_t n, StepConst c)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t a = now[i], e = ever[i], rb, fl;
        int model = 0;
        if (is_variation) {
            model = (int)nobj_of(e) - 4;                   // fetch_env.py:650-651
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        int nc = ncon[i];
        nc = nc < 0 ? 0 : (nc > maxcon ? maxcon : nc);
        const uint32_t m = row_pair_mask<VW>(s_tab + model * BPG_MAX_GEOMS, reinterpret_cast<const uint32_t *>(contacts) + i * maxcon, nc);
        step_core(a, e, m, c, rb, fl);
        now[i] = a; ever[i] = e;
        if (reward) reward[i] = rb;
        if (flags) flags[i] = (uint8_t)fl;
    }
}

__global__ void __launch_bounds__(256)
k_synth_masks(const uint32_t *__restrict__ ever, int64_t n, int t0, int T, uint32_t *__restrict__ out,
              int64_t ld, SynthConst s)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const SynthLane l = synth_lane(s, (uint32_t)i, nobj_of(ever[i]));
        for (int t = 0; t < T; t++) {
            uint32_t tt = (uint32_t)(t0 + t);
            tt = tt > ELAPSED_MAX ? ELAPSED_MAX : tt;
            out[(int64_t)t * ld + i] = synth_mask(l, tt);
        }
    }
}

// One env.step as the rollout worker consumes it (rollout.py:121-131): the step, the
// float32 obs['achieved_goal'] row (at out + e*ld: a dense (n, dimg) array or the time slot
// of an episode buffer) and info['is_success'] as float32, in ONE launch -- the packed state
// never makes a second trip through HBM and the host issues one launch instead of five.
// A wave owns tiles of 64 consecutive envs: lane = env for the step, then the wave streams the tile's
// contiguous 64 x dimg float region from a wave-private LDS copy of the packed states -- no block-wide
// barrier, so the WPB waves of a block come and go on their own.  count (optional): += the number of envs
// whose has_succeeded latch is set after this step -- `successful = successes[-1]` of rollout.py:163-167
// inside the LAST step's launch of a rollout instead of a launch of its own.
// NT: the write-once outputs (achieved_goal rows, is_success, reward, flags, the action copy) leave as nontemporal
// stores.  Plain stores allocate in the L2 and are written back when the kernel ends, behind the last wave: in a
// graph-replayed train of step launches at 65 536 envs that put 1.4 us of store drain inside every launch and 0.8 us
// of write-back behind it (5.9 -> 4.1 us per step; 35.0 -> 33.3 us at 1 M envs, 152 -> 115 us at 4 M;
// round 4's tools/tune_round4.hip, profiles/r04_tune_step_train.log).  The packed state, re-read by the next step, stays plain.
// BPG_TL_STEP(slot, agp): hooks.hip.h (empty in the library build).
template <int SIDE, int SRC, bool VEC4, int WPB = 1, bool NT = false>
__global__ void __launch_bounds__(64 * WPB)
k_step_record(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
              uint32_t *__restrict__ reward, uint8_t *__restrict__ flags, float *__restrict__ succ_f32,
              float *__restrict__ ag, int64_t ld, int64_t n, StepConst c, SynthConst s, PairTable tabv,
              const float4 *__restrict__ u_in, float4 *__restrict__ u_out, CountArgs ca, float *__restrict__ ret_acc)
{
    constexpr int DIMG = SIDE * SIDE;
    __shared__ uint2 hist_all[WPB][64];
    __shared__ PairTable tab_all[WPB];
    BPG_TL_STEP(0, ag);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint2 *hist = hist_all[wave];
    PairTable *tab = &tab_all[wave];
    if (lane == 0) *tab = tabv;
    uint32_t nsucc = 0;
    const int64_t ntiles = (n + 63) >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < ntiles; tile += (int64_t)gridDim.x * WPB) {
        const int64_t base = tile * 64;
        const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
        if (lane < nvalid) {
            const int64_t i = base + lane;
            uint32_t a = now[i], e = ever[i], rb, fl;
            const uint32_t mk = fetch_mask<SRC>(SRC == SRC_MASKS ? masks[i] : 0u, a, e, (uint32_t)i, s);
            step_core(a, e, mk, c, rb, fl);
            now[i] = a; ever[i] = e;
            hist[lane] = make_uint2(a, e);
            nsucc += fl & BPG_FLAG_SUCCESS;
            if (reward) st_1<NT>(&reward[i], rb);
            if (flags) st_1<NT>(&flags[i], (uint8_t)fl);
            if (succ_f32) st_1<NT>(&succ_f32[i], (fl & BPG_FLAG_SUCCESS) ? 1.0f : 0.0f);
            if (u_out) st_f4<NT>(&u_out[i], u_in[i]);   // acts.append(u.copy()) of rollout.py:147, same launch
            if (ret_acc) ret_acc[i] = ret_acc[i] + __uint_as_float(rb);          // bpg_step_args.d_return_acc (re-read next step: plain)
        }
        wave_lds_sync();
        if (ag) {
            float *o = ag + base * ld;
            if (VEC4) {
                constexpr int QPR = DIMG / 4;
                for (int cidx = lane; cidx < nvalid * QPR; cidx += 64) {
                    const int e = cidx / QPR, q = cidx - e * QPR;
                    const uint2 h = hist[e];
                    const uint32_t pw = tab->quad[q];
                    float4 v;
                    v.x = tern(h.x, h.y, pw & 31u);
                    v.y = tern(h.x, h.y, (pw >> 5) & 31u);
                    v.z = tern(h.x, h.y, (pw >> 10) & 31u);
                    v.w = tern(h.x, h.y, (pw >> 15) & 31u);
                    st_f4<NT>(reinterpret_cast<float4 *>(o + e * ld + 4 * q), v);
                }
            } else {
                for (int f = lane; f < nvalid * DIMG; f += 64) {
                    const int e = f / DIMG, k = f - e * DIMG;
                    const uint2 h = hist[e];
                    st_1<NT>(&o[e * ld + k], tern(h.x, h.y, tab->flat[k]));
                }
            }
        }
        wave_lds_sync();                                 // the tile is written again by the next trip
    }
    if (ca.count) block_count_add(nsucc, ca);            // uniform
    BPG_TL_STEP(1, ag);
}

// Geom-id contact lists -> packed pair masks (fetch_env.py:158-167 up to the matrix write), so
// a backend that produces contact lists can feed the mask-based fused kernels.
template <int VW>
__global__ void __launch_bounds__(256)
// T = 1, ld = n: one step's lists.  T > 1: the lists of a whole recorded rollout, time-major (T, n, maxcon, 2) / (T, n), into the
// (T, ld) mask stream of the fused rollout; num_grey (optional): the model of env e is the one of the episode that is ABOUT
// to be drawn (Variation picks its sim by num_objs = 4 + num_grey, fetch_env.py:647-651), not the one in the state.
k_contacts_to_masks(const uint32_t *__restrict__ ever, const int16_t *__restrict__ contacts,
                    const int32_t *__restrict__ ncon, int maxcon, const int8_t *__restrict__ id2obj,
                    int is_variation, uint32_t valid, uint32_t *__restrict__ out, int64_t n, int T, int64_t ld,
                    const int32_t *__restrict__ num_grey)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, total = n * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = T == 1 ? 0 : i / n, e = i - t * n;
        int model = 0;
        if (is_variation) {
            model = num_grey ? num_grey[e] : (int)nobj_of(ever[e]) - 4;
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        const int8_t *tab = s_tab + model * BPG_MAX_GEOMS;
        int nc = ncon[i];
        nc = nc < 0 ? 0 : (nc > maxcon ? maxcon : nc);
        const uint32_t *row = reinterpret_cast<const uint32_t *>(contacts) + i * maxcon;
        out[t * ld + e] = row_pair_mask<VW>(tab, row, nc) & valid;
    }
}

// The same for RAGGED lists (CSR): env i owns contacts offsets[i] .. offsets[i+1]-1 of one packed
// (total, 2) geom-id array -- what a physics backend produces when it compacts sim.data.contact
// of all envs instead of padding every env to the longest list.  Consecutive lanes own consecutive
// segments, so a wave's loads fall into one contiguous stretch of the array.
__global__ void __launch_bounds__(256)
k_contacts_csr_to_masks(const uint32_t *__restrict__ ever, const uint32_t *__restrict__ pairs,
                        const int64_t *__restrict__ offsets, int64_t total, const int8_t *__restrict__ id2obj,
                        int is_variation, uint32_t valid, uint32_t *__restrict__ out, int64_t n)
{
    __shared__ int8_t s_tab[BPG_MAX_MODELS * BPG_MAX_GEOMS];
    for (int k = threadIdx.x; k < BPG_MAX_MODELS * BPG_MAX_GEOMS; k += blockDim.x) s_tab[k] = id2obj[k];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int model = 0;
        if (is_variation) {
            model = (int)nobj_of(ever[i]) - 4;
            model = model < 0 ? 0 : (model > 2 ? 2 : model);
        }
        const int8_t *tab = s_tab + model * BPG_MAX_GEOMS;
        int64_t lo = offsets[i], hi = offsets[i + 1];
        lo = lo < 0 ? 0 : (lo > total ? total : lo);                 // never index outside the packed array
        hi = hi < lo ? lo : (hi > total ? total : hi);
        uint32_t m = 0;
        for (int64_t k = lo; k < hi; k++) m |= pair_bit_of(tab, pairs[k]);
        out[i] = m & valid;
    }
}

}  // namespace bpgk

extern "C" {

static int launch_step(bpg_batch *b, int src, const uint32_t *d_masks, float *d_reward, uint8_t *d_flags,
                       void *stream)
{
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    uint32_t *rw = reinterpret_cast<uint32_t *>(d_reward);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(d_masks) && aligned16(d_reward) && aligned4(d_flags);
    if (vec) {
        const dim3 grid(grid_for((b->n + 3) / 4, 256)), block(256);
        if (src == SRC_MASKS)
            hipLaunchKernelGGL(k_step_vec4<SRC_MASKS>, grid, block, 0, st, b->now, b->ever, d_masks, rw, d_flags,
                               b->n, c, s);
        else
            hipLaunchKernelGGL(k_step_vec4<SRC_SYNTH>, grid, block, 0, st, b->now, b->ever, d_masks, rw, d_flags,
                               b->n, c, s);
    } else {
        const dim3 grid(grid_for(b->n, 256)), block(256);
        if (src == SRC_MASKS)
            hipLaunchKernelGGL(k_step_scalar<SRC_MASKS>, grid, block, 0, st, b->now, b->ever, d_masks, rw,
                               d_flags, b->n, c, s);
        else
            hipLaunchKernelGGL(k_step_scalar<SRC_SYNTH>, grid, block, 0, st, b->now, b->ever, d_masks, rw,
                               d_flags, b->n, c, s);
    }
    return check_launch("k_step");
}

int bpg_step_masks(bpg_batch *b, const uint32_t *d_masks, float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (!aligned4(d_masks) || !aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    return launch_step(b, SRC_MASKS, d_masks, d_reward, d_flags, stream);
}

int bpg_step_synth(bpg_batch *b, float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (!aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    return launch_step(b, SRC_SYNTH, nullptr, d_reward, d_flags, stream);
}

int bpg_step_contacts(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                      float *d_reward, uint8_t *d_flags, void *stream)
{
    if (!b || !d_contacts || !d_ncon) return fail(BPG_ERR_ARG, "NULL argument");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_reward)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (root_of(b)->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    const dim3 grid(grid_for(b->n, 256, 4096)), block(256);
    if (maxcon % 4 == 0 && aligned16(d_contacts))
        hipLaunchKernelGGL(k_step_contacts<4>, grid, block, 0, (hipStream_t)stream, b->now, b->ever, d_contacts, d_ncon, maxcon,
                           b->d_id2obj, b->info.is_variation, reinterpret_cast<uint32_t *>(d_reward), d_flags, b->n, step_const(b));
    else
        hipLaunchKernelGGL(k_step_contacts<1>, grid, block, 0, (hipStream_t)stream, b->now, b->ever, d_contacts, d_ncon, maxcon,
                           b->d_id2obj, b->info.is_variation, reinterpret_cast<uint32_t *>(d_reward), d_flags, b->n, step_const(b));
    return check_launch("k_step_contacts");
}

static int launch_contacts_to_masks(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                                    uint32_t *d_masks, int T, int64_t ld, const int32_t *d_num_grey, hipStream_t st)
{
    const dim3 grid(grid_for(b->n * T, 256)), block(256);
    if (maxcon % 4 == 0 && aligned16(d_contacts))
        hipLaunchKernelGGL(k_contacts_to_masks<4>, grid, block, 0, st, b->ever, d_contacts, d_ncon, maxcon, b->d_id2obj,
                           b->info.is_variation, b->info.valid_pairs, d_masks, b->n, T, ld, d_num_grey);
    else
        hipLaunchKernelGGL(k_contacts_to_masks<1>, grid, block, 0, st, b->ever, d_contacts, d_ncon, maxcon, b->d_id2obj,
                           b->info.is_variation, b->info.valid_pairs, d_masks, b->n, T, ld, d_num_grey);
    return check_launch("k_contacts_to_masks");
}

int bpg_contacts_to_masks(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int maxcon,
                          uint32_t *d_masks, void *stream)
{
    if (!b || !d_contacts || !d_ncon || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_masks)) return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (root_of(b)->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    return launch_contacts_to_masks(b, d_contacts, d_ncon, maxcon, d_masks, 1, b->n, nullptr, (hipStream_t)stream);
}

int bpg_contacts_to_masks_tb(bpg_batch *b, const int16_t *d_contacts, const int32_t *d_ncon, int T, int maxcon,
                             const int32_t *d_num_grey, uint32_t *d_masks, int64_t ld_masks, void *stream)
{
    if (!b || !d_contacts || !d_ncon || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (T < 1 || T > (int)ELAPSED_MAX) return fail(BPG_ERR_ARG, "T must be in 1..1023");
    if (maxcon < 1 || maxcon > 4096) return fail(BPG_ERR_ARG, "maxcon must be in 1..4096");
    if (!aligned4(d_contacts) || !aligned4(d_masks) || !aligned4(d_ncon) || !aligned4(d_num_grey))
        return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    const int64_t ld = ld_masks ? ld_masks : b->n;
    if (ld < b->n) return fail(BPG_ERR_ARG, "ld_masks must be 0 or >= n_envs");
    // element offsets i * maxcon with i < n_envs * T, and ld * T for the mask rows, are formed in 64 bits: keep them well inside
    if (b->n > (((int64_t)1 << 62) / T) / maxcon || ld > ((int64_t)1 << 62) / T)
        return fail(BPG_ERR_ARG, "n_envs * T * maxcon is too large");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (root_of(b)->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    return launch_contacts_to_masks(b, d_contacts, d_ncon, maxcon, d_masks, T, ld, d_num_grey, (hipStream_t)stream);
}

int bpg_contacts_csr_to_masks(bpg_batch *b, const int16_t *d_pairs, const int64_t *d_offsets, int64_t total,
                              uint32_t *d_masks, void *stream)
{
    if (!b || !d_offsets || !d_masks || (!d_pairs && total > 0)) return fail(BPG_ERR_ARG, "NULL argument");
    if (total < 0) return fail(BPG_ERR_ARG, "total < 0");
    if (!aligned4(d_pairs) || !aligned4(d_masks) || ((uintptr_t)d_offsets & 7u))
        return fail(BPG_ERR_ARG, "pairs / masks must be 4-byte, offsets 8-byte aligned");
    const int models = b->info.is_variation ? BPG_MAX_MODELS : 1;
    for (int m = 0; m < models; m++)
        if (root_of(b)->ngeom[m] == 0) return fail(BPG_ERR_ARG, "id2obj table not set (bpg_set_id2obj)");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_contacts_csr_to_masks, dim3(grid_for(b->n, 256)), dim3(256), 0, (hipStream_t)stream, b->ever,
                       reinterpret_cast<const uint32_t *>(d_pairs), d_offsets, total, b->d_id2obj, b->info.is_variation,
                       b->info.valid_pairs, d_masks, b->n);
    return check_launch("k_contacts_csr_to_masks");
}

int bpg_step_record_v2(bpg_batch *b, const bpg_step_args *args, void *stream)
{
    if (!b || !args) return fail(BPG_ERR_ARG, "NULL argument");
    static_assert(sizeof(bpg_step_args) == BPG_STEP_ARGS_MIN_SIZE, "BPG_STEP_ARGS_MIN_SIZE");
    bpg_step_args a;
    if (int rc = take_args(a, args, BPG_STEP_ARGS_MIN_SIZE, "bpg_step_args")) return rc;
    if ((uintptr_t)a.d_success_count & 7u) return fail(BPG_ERR_ARG, "counters must be 8-byte aligned");
    if ((a.d_u_out != nullptr) != (a.d_u_in != nullptr)) return fail(BPG_ERR_ARG, "d_u_in and d_u_out go together");
    if (!aligned16(a.d_u_in) || !aligned16(a.d_u_out)) return fail(BPG_ERR_ARG, "action buffers must be 16-byte aligned");
    if (a.d_ag && a.ld_ag < b->info.dimg) return fail(BPG_ERR_ARG, "ld_ag must be >= dimg");
    if (!aligned4(a.d_masks) || !aligned4(a.d_reward) || !aligned4(a.d_success_f32) || !aligned4(a.d_ag) || !aligned4(a.d_return_acc))
        return fail(BPG_ERR_ARG, "pointers must be 4-byte aligned");
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    const PairTable tab = pair_table(b->info);
    const CountArgs ca = count_args(b, a.d_success_count, nullptr);
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(64);
    const bool vec = (b->info.side % 2 == 0) && (a.ld_ag % 4 == 0) && aligned16(a.d_ag);
    const bool synth = (a.d_masks == nullptr);
    uint32_t *rw = reinterpret_cast<uint32_t *>(a.d_reward);
    hipStream_t st = (hipStream_t)stream;
    int rc = dispatch_side(b->info.side, [&](auto side) {
        constexpr int S = decltype(side)::value;
        constexpr bool EVEN = (S % 2 == 0);
#define BPG_LAUNCH_SR(SRC, V)                                                                                  \
        hipLaunchKernelGGL((k_step_record<S, SRC, V, 1, true>), grid, block, 0, st, b->now, b->ever, a.d_masks, rw, a.d_flags, \
                           a.d_success_f32, a.d_ag, a.ld_ag, b->n, c, s, tab, reinterpret_cast<const float4 *>(a.d_u_in), \
                           reinterpret_cast<float4 *>(a.d_u_out), ca, a.d_return_acc)
        if (synth) {
            if (vec) BPG_LAUNCH_SR(SRC_SYNTH, EVEN); else BPG_LAUNCH_SR(SRC_SYNTH, false);
        } else {
            if (vec) BPG_LAUNCH_SR(SRC_MASKS, EVEN); else BPG_LAUNCH_SR(SRC_MASKS, false);
        }
#undef BPG_LAUNCH_SR
        return BPG_OK;
    });
    return rc != BPG_OK ? rc : check_launch("k_step_record");
}

int bpg_synth_masks(bpg_batch *b, int t0, int T, uint32_t *d_masks, void *stream)
{
    return bpg_synth_masks_ld(b, t0, T, d_masks, b ? b->n : 0, stream);
}

int bpg_synth_masks_ld(bpg_batch *b, int t0, int T, uint32_t *d_masks, int64_t ld, void *stream)
{
    if (!b || !d_masks) return fail(BPG_ERR_ARG, "NULL argument");
    if (t0 < 0 || T < 1) return fail(BPG_ERR_ARG, "bad t0/T");
    if (ld < b->n) return fail(BPG_ERR_ARG, "ld must be >= n_envs");
    DeviceGuard guard(b->device);
    hipLaunchKernelGGL(k_synth_masks, dim3(grid_for(b->n, 256)), dim3(256), 0, (hipStream_t)stream, b->ever, b->n,
                       t0, T, d_masks, ld, synth_const(b));
    return check_launch("k_synth_masks");
}

}  // extern "C"
