// episode.hip -- the fused rollout that also writes the batch-major float32 episode arrays the reference returns
// (rollout.py:155-172).
#include "state.hip.h"

namespace bpgk {

// ------------------------------ episode writer ------------------------------
// rollout.py:100-172 for one wave = 64 consecutive envs per block.  Phase 1:
// lane = env, T fused steps, packed (now, ever) history of every step kept in
// LDS (row stride R odd -> conflict-free ds_write_b64).  Phase 2: the wave
// streams its contiguous slices of episode['ag'] (B,T+1,dimg), episode['g']
// (B,T,dimg) and episode['info_is_success'] (B,T,1) as float32.
template <int SIDE, int SRC, bool NT>
__global__ void __launch_bounds__(128)
k_episode(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks,
          uint32_t *__restrict__ reward_tb, uint8_t *__restrict__ flags_tb,
          float *__restrict__ ep_ag, float *__restrict__ ep_g, float *__restrict__ ep_succ,
          int64_t n, int T, int64_t ldm, int64_t ldo, int RC, int R, StepConst c, SynthConst s, PairTable tabv,
          ResetConst rc, CountArgs ca, uint64_t *__restrict__ okbits_out, float *__restrict__ ep_return, uint8_t *__restrict__ ep_fail)
{
    // The T+1 matrices of an episode are produced in chunks of RC rows (one chunk = the whole
    // episode for T <= 63): a short step phase (lane = env, state in registers, packed rows of
    // the chunk into LDS, [64][R] x 8 B with R = RC | 1 odd -> conflict-free ds_write_b64)
    // followed by the streaming phase for those rows.
    constexpr int DIMG = SIDE * SIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *hist = reinterpret_cast<uint2 *>(smem);                       // [64][R]
    PairTable *tab = reinterpret_cast<PairTable *>(smem + (size_t)64 * R * sizeof(uint2));
    const int lane = threadIdx.x, nlanes = blockDim.x;     // wave 0 (lane < 64) steps the envs, every wave streams
    if (lane == 0) *tab = tabv;
    if (ca.zero_next && blockIdx.x == 0 && lane == 0) *ca.zero_next = 0ull;
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nvalid = (int)((n - base) < 64 ? (n - base) : 64);
    const bool active = lane < nvalid;                        // nvalid <= 64: only wave 0 can be active
    const int64_t i = base + (active ? lane : 0);
    uint32_t a = 0, e = 0;
    uint64_t okb = 0ull;                 // bit t = the success condition held at step t (t < 64): the step rewards
    uint32_t nfail = 0;                  // steps whose success condition did not hold: the episode return is -nfail
    SynthLane sl = {};
    if (active) {
        a = now[i]; e = ever[i];
        if (rc.enabled) reset_core(a, e, i, rc);
        if (SRC == SRC_SYNTH) sl = synth_lane(s, (uint32_t)i, nobj_of(e));
    }
    uint2 *my = hist + lane * R;
    constexpr int EP_BATCH = 10;        // masks of 10 steps are requested together: one memory
                                        // latency per batch instead of one per step
    for (int r0 = 0; r0 <= T; r0 += RC) {
        const int r1 = (r0 + RC < T + 1) ? r0 + RC : T + 1;      // rows [r0, r1) of every env
        if (active) {
            if (r0 == 0) my[0] = make_uint2(a, e);               // row 0: initial_ag (rollout.py:82-83)
            const int tb = (r0 == 0) ? 0 : r0 - 1, te = r1 - 1;  // steps tb .. te-1 produce rows tb+1 .. te
            for (int t0 = tb; t0 < te; t0 += EP_BATCH) {
                uint32_t mb[EP_BATCH];
                if (SRC == SRC_MASKS) {
#pragma unroll
                    for (int u = 0; u < EP_BATCH; u++)
                        mb[u] = (t0 + u < te) ? masks[(int64_t)(t0 + u) * ldm + i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < EP_BATCH; u++) {
                    const int t = t0 + u;
                    if (t < te) {
                        uint32_t m, rb, fl;
                        if (SRC == SRC_MASKS) m = mb[u];
                        else m = synth_mask(sl, (a >> ELAPSED_SHIFT) & ELAPSED_MAX);
                        step_core(a, e, m, c, rb, fl);
                        my[t + 1 - r0] = make_uint2(a, e);
                        okb |= (uint64_t)((fl >> 2) & 1u) << (t & 63);
                        nfail += ((fl >> 2) & 1u) ^ 1u;
                        if (reward_tb) reward_tb[(int64_t)t * ldo + i] = rb;
                        if (flags_tb) flags_tb[(int64_t)t * ldo + i] = (uint8_t)fl;
                    }
                }
            }
        }
        __syncthreads();
        if (ep_ag)
            wave_write_matrices<SIDE, NT>(ep_ag + (base * (int64_t)(T + 1) + r0) * DIMG, nvalid, r1 - r0,
                                          (T + 1) * DIMG, hist, R, tab, lane, nlanes);
        if (ep_succ) {
            // info['is_success'] after step t = latch in row t+1 (rollout.py:127-128,131)
            const int rs = (r0 == 0) ? 1 : r0, cnt = r1 - rs;    // rows rs .. r1-1 -> steps rs-1 .. r1-2
            float *o = ep_succ + base * (int64_t)T + (rs - 1);
            const int total = nvalid * cnt;
            int ee = 0, rem = lane;
            while (cnt > 0 && rem >= cnt) { rem -= cnt; ee++; }
            for (int f = lane; f < total; f += nlanes) {
                st_1<NT>(&o[(int64_t)ee * T + rem], (hist[ee * R + (rs - r0) + rem].y >> 31) ? 1.0f : 0.0f);
                rem += nlanes;
                while (rem >= cnt) { rem -= cnt; ee++; }
            }
        }
        __syncthreads();                                         // the tile is reused by the next chunk
    }
    if (active) {
        now[i] = a; ever[i] = e;
        // the step rewards of the episode as one word per env: k_returns_from_okbits turns it into the env's
        // G row (policy_gradient/rollout.py:255-258)
        if (okbits_out) okbits_out[i] = okb;
        if (ep_return) ep_return[i] = -(float)nfail;              // bpg_rollout_args.d_episode_return
        if (ep_fail) ep_fail[i] = (uint8_t)nfail;                 // ... d_episode_fail_count (the host guarantees T <= 255)
    }
    if (ca.count && lane < 64) {
        const unsigned long long nb = (unsigned long long)__popcll(__ballot(active && (e >> 31) != 0));
        if (lane == 0) shard_count_add(nb, ca);
    }
    if (ep_g) {
        // episode['g']: T copies of the constant goal row per env (rollout.py:151)
        float *o = ep_g + base * (int64_t)T * DIMG;
        const int total = nvalid * T * DIMG;
        if (SIDE % 2 == 0) {
            constexpr int QPR = DIMG / 4;
            float4 *o4 = reinterpret_cast<float4 *>(o);
            for (int cidx = lane; cidx < total / 4; cidx += nlanes) {
                const int q = cidx % QPR;
                st_f4<NT>(&o4[cidx], make_float4(tab->goal[4 * q], tab->goal[4 * q + 1], tab->goal[4 * q + 2],
                                                 tab->goal[4 * q + 3]));
            }
        } else {
            for (int f = lane; f < total; f += nlanes) st_1<NT>(&o[f], tab->goal[f % DIMG]);
        }
    }
}

int launch_episode(bpg_batch *b, const EpisodeLaunch &e, hipStream_t st)
{
    // Rows per chunk.  One chunk whenever the whole episode fits a 64-row (33 KB) tile: the wave then
    // streams ONE contiguous 64 x (T+1) x dimg region.  Splitting T = 50 into two 26-row chunks was
    // measured 30 % SLOWER (4.1 vs 3.1 ms for the full episode, 2.4 vs 1.9 ms ag-only) although it
    // doubles the waves per CU: the per-env 3.7 KB segments with 3.7 KB holes lose the long contiguous
    // write streams.  Longer horizons are chunked because the tile would not fit.
    const int rows_total = e.T + 1;
    int nchunks = 1;
    while ((rows_total + nchunks - 1) / nchunks > 64) nchunks++;
    const int RC = (rows_total + nchunks - 1) / nchunks;
    const int R = RC | 1;
    const size_t lds = (size_t)64 * R * sizeof(uint2) + sizeof(PairTable);
    const PairTable tab = pair_table(b->info);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    // two waves per 64-env block: the first steps the envs, both stream the block's region (2.5-3 %
    // over one wave on three boxes; three or four waves add nothing)
    const dim3 grid((unsigned)((b->n + 63) / 64)), block(128);
    const bool synth = (e.d_masks == nullptr);
    int rcode = dispatch_side(b->info.side, [&](auto side) {
        constexpr int S = decltype(side)::value;
        if (synth)
            hipLaunchKernelGGL((k_episode<S, SRC_SYNTH, false>), grid, block, lds, st, b->now, b->ever, e.d_masks, e.reward_tb,
                               e.flags_tb, e.ep_ag, e.ep_g, e.ep_success, b->n, e.T, e.ldm, e.ldo, RC, R, c, s, tab, e.rc, e.ca,
                               e.okbits, e.ep_return, e.ep_fail);
        else
            hipLaunchKernelGGL((k_episode<S, SRC_MASKS, false>), grid, block, lds, st, b->now, b->ever, e.d_masks, e.reward_tb,
                               e.flags_tb, e.ep_ag, e.ep_g, e.ep_success, b->n, e.T, e.ldm, e.ldo, RC, R, c, s, tab, e.rc, e.ca,
                               e.okbits, e.ep_return, e.ep_fail);
        return BPG_OK;
    });
    return rcode != BPG_OK ? rcode : check_launch("k_episode");
}

}  // namespace bpgk
