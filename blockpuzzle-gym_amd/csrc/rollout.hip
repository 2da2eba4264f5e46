// rollout.hip -- RolloutStudent.generate_rollouts' inner loop (rollout.py:100-153) for T steps in ONE launch: the
// packed state stays in registers, masks / rewards / flags are time-major coalesced streams.
#include "state.hip.h"

namespace bpgk {

// k_rollout<SRC, EPL, OUT, MODE, UNROLL, PIPE, LB, AUX>: T steps in one launch, EPL envs per lane, state in registers.
// Masks / rewards / flags are time-major (T, ld) so every access is an (EPL * 4)-byte-per-lane coalesced stream.
// SRC: SRC_MASKS (contacts read from the mask stream) or SRC_SYNTH (generated in-kernel, build-defined).
// OUT: which per-step streams are written (bit 0 reward, bit 1 flags) -- compile-time, because a store behind a run-time
//   `if (pointer)` makes the compiler assume "maybe not issued" when it counts outstanding memory operations, and every wait
//   then also drains the stores.
// MODE: how the launch starts.  MODE_CONT: from the state as it is.  MODE_RESET: with reset_all_rollouts (rollout.py:79) --
//   compile-time because the elapsed counter is then the same for every lane (step t of the launch), so the time-limit test,
//   the saturation and the generator's Weyl term run on the scalar unit.  MODE_MULTI: `rollouts` consecutive rollouts in this
//   ONE launch (each: reset, T steps, success count into its own counter word), rollout r on rows r*T .. r*T+T-1 of the
//   streams -- exactly what `rollouts` launches in a row would do, without their fixed cost (5-6 us per launch, which is all a
//   65 536-env rollout has left to lose).  Envs are independent and a lane only ever touches its own, so nothing has to be
//   synchronised between the rollouts of a launch; the counts go out as plain atomics that drain behind the next rollout's steps.
// UNROLL, PIPE: steps per batch; PIPE = software-pipelined mask requests (see the loop), for the latency-bound batch sizes.  The
//   plain loop (UNROLL requests, UNROLL steps, repeat) is what the bandwidth-bound fat shape runs: with >= 16 waves per CU
//   other waves cover the round trips.
// LB: the largest block the instantiation is ever launched with (the register budget follows from it: 128 VGPRs at 1024
//   lanes, 512 / waves-per-SIMD below).
// AUX: AUX_OKB -- the launch also stores every env's step-ok word (bit t = the success condition held at step t, i.e. the step
//   reward was -0.0f; T <= 64) into okbits[n], 8 bytes per env instead of a 4 T-byte reward stream, from which
//   k_returns_from_okbits computes the policy-gradient return.  (Computing G inside this launch, right after a wave's steps,
//   was built and measured: 141-150 us against 52 + 83 us for the two launches at 1 M envs, profiles/r03_tune_pg_path.log --
//   the memory-bound step phase and the VALU-bound return phase of the same waves do not overlap -- and removed.)
//   AUX_RET -- a per-env counter of held steps in registers, for the episode return / fail count when there is no step-ok
//   word to take it from (compile-time: the contacts-in-kernel form lives on occupancy and loses a wave per SIMD to the extra
//   registers).  AUX_NONE -- neither.
// The by-value constants a form does not use are not parameters of its instantiations (NoConst): the kernarg segment of the
// masks forms carries no SynthConst, that of MODE_CONT no ResetConst.  BPG_TL(slot) / BPG_ROLLOUT_MINB: hooks.hip.h.
// (MODE_* / AUX_* / SRC_* / OUT_* are plain ints, not enumerators of an unnamed enum: such a type in a kernel's
// template-dependent signature is mangled with a per-pass counter -- `$_0` on the host, `$_1` on the device -- and the launch
// then dies with "Cannot find Symbol".)
constexpr int MODE_CONT = 0, MODE_RESET = 1, MODE_MULTI = 2;
constexpr int AUX_NONE = 0, AUX_OKB = 1, AUX_RET = 2;
struct NoConst {};
template <int SRC> using SynthArg = std::conditional_t<SRC == SRC_SYNTH, SynthConst, NoConst>;
template <int MODE> using ResetArg = std::conditional_t<MODE != MODE_CONT, ResetConst, NoConst>;

template <int SRC, int EPL, int OUT, int MODE, int UNROLL, bool PIPE, int LB, int AUX>
__global__ void __launch_bounds__(LB, BPG_ROLLOUT_MINB)
k_rollout(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, const uint32_t *__restrict__ masks_all,
          uint32_t *__restrict__ reward_all, uint8_t *__restrict__ flags_all, int64_t n, int T,
          int64_t ld_masks, int64_t ld_out,
          StepConst c, SynthArg<SRC> s, ResetArg<MODE> rc, CountArgs ca, uint64_t *__restrict__ okbits, int rollouts,
          uint32_t *__restrict__ ep_return_all, uint8_t *__restrict__ ep_fail_all, int64_t ld_env)
{
    constexpr bool RESET = MODE != MODE_CONT, MULTI = MODE == MODE_MULTI, OKB = AUX == AUX_OKB, RET = AUX == AUX_RET;
    constexpr bool NT = true, NTL = true;       // write-once streams leave nontemporal, the read-once mask stream arrives so
    static_assert(!MULTI || !OKB, "several rollouts per launch: no step-ok words");
    using Vec = typename LaneVec<EPL>::vec;
    using Fl = typename LaneVec<EPL>::fl;
    const int64_t nvec = n / EPL;               // the host guarantees n, ld_masks, ld_out % EPL == 0
    const int64_t ldm = ld_masks / EPL, ldo = ld_out / EPL;      // row strides of the (T, ld) streams
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (ca.zero_next && blockIdx.x == 0 && threadIdx.x == 0) *ca.zero_next = 0ull;
    BPG_TL(0);
    int tl_pass = 0;
    (void)tl_pass;
    for (int ro = 0; ro < (MULTI ? rollouts : 1); ro++) {
    const uint32_t *__restrict__ masks = masks_all + (MULTI ? (int64_t)ro * T * ld_masks : 0);
    uint32_t *__restrict__ reward_tb = reward_all + (MULTI ? (int64_t)ro * T * ld_out : 0);
    uint8_t *__restrict__ flags_tb = flags_all + (MULTI ? (int64_t)ro * T * ld_out : 0);
    // rollout ro of a MULTI launch owns row ro of the (rollouts, ld_env) per-env arrays (ld_env = n unless the caller says otherwise)
    uint32_t *__restrict__ ep_return = ep_return_all ? ep_return_all + (MULTI ? (int64_t)ro * ld_env : 0) : nullptr;
    uint8_t *__restrict__ ep_fail = ep_fail_all ? ep_fail_all + (MULTI ? (int64_t)ro * ld_env : 0) : nullptr;
    uint32_t nsucc = 0;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint32_t an[EPL] = {}, ee[EPL] = {};
        // a Variation reset (fetch_env.py:646-679) overwrites the whole state: nothing to read, and
        // no wave starts its episode behind a memory round trip
        bool keeps_state = true;
        if constexpr (RESET) keeps_state = !rc.is_variation;
        if (keeps_state) {
            unpack<EPL>(reinterpret_cast<const Vec *>(now)[v], an);
            unpack<EPL>(reinterpret_cast<const Vec *>(ever)[v], ee);
        }
        SynthLane sl[EPL];
        // step_core() with the state unpacked for the duration of the launch: the touching set,
        // the ever set (+ num_objs bits, untouched), the elapsed counter and the success latch (as
        // an all-ones / zero mask) each in a register of their own, and the two comparisons of
        // step_core turned into sign-bit arithmetic -- 11 instead of ~20 instructions per env-step.
        uint32_t cm[EPL], ev[EPL], el[EPL], lat[EPL], stk[EPL];
        uint64_t okb[EPL];                          // OKB: bit t = the success condition held at step t
        // RET: steps of this launch whose success condition held, counted for every env (one subtraction per
        // env-step): the undiscounted episode return is -(T - nok), bpg_rollout_args.d_episode_return.  Compile-time,
        // because the contacts-in-kernel form lives on occupancy and loses a wave per SIMD to the two extra registers.
        // With OKB the count is the population of the step-ok word and costs nothing per step.
        constexpr bool COUNT = RET && !OKB;
        uint32_t nok[EPL] = {};
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            if constexpr (RESET) reset_core(an[k], ee[k], EPL * v + k, rc, MULTI ? (uint32_t)ro : 0u, MULTI ? (int64_t)ro * ld_env : 0);
            if constexpr (SRC == SRC_SYNTH) sl[k] = synth_lane(s, (uint32_t)(EPL * v + k), nobj_of(ee[k]));
            cm[k] = an[k] & PAIR_BITS;
            el[k] = RESET ? 0u : (an[k] >> ELAPSED_SHIFT) & ELAPSED_MAX;
            ev[k] = ee[k] & ~SUCC_BIT;
            lat[k] = (uint32_t)((int32_t)ee[k] >> 31);
            // pairs that are never demoted (StepConst); the synthetic source only emits pairs of present objects
            stk[k] = (SRC == SRC_SYNTH) ? 0u : sticky_of(c, ee[k]);
            okb[k] = 0ull;
        }
        const Vec *mrow = reinterpret_cast<const Vec *>(masks) + v;
        Vec *rrow = reinterpret_cast<Vec *>(reward_tb) + v;
        Fl *frow = reinterpret_cast<Fl *>(flags_tb) + v;
        // ubit = 1 << (step within its batch of UNROLL): the step-ok bits of a batch collect in a 32-bit word with
        // compile-time masks and join the 64-bit word once per batch (OKB)
        uint32_t okpart[EPL] = {};
        auto one_step = [&](int t, const uint32_t *mt, uint32_t ubit) {
            uint32_t rb[EPL], fl[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                if (RESET) el[k] = (uint32_t)t;            // uniform: elapsed == step of the launch (T <= 1023)
                if (SRC == SRC_SYNTH) cm[k] = synth_mask(sl[k], el[k]) & c.valid;
                else cm[k] = (mt[k] & c.valid) | (cm[k] & stk[k]);
                ev[k] |= cm[k];
                const uint32_t bad = (c.want_touch & ~cm[k]) | (c.want_never & ev[k]);    // < 2^21
                const uint32_t okm = (uint32_t)((int32_t)(bad - 1u) >> 31);               // ~0 iff bad == 0
                lat[k] |= okm;
                if (COUNT) nok[k] -= okm;                                                 // okm is 0 or -1
                if (OKB) okpart[k] |= okm & ubit;
                el[k] = min(el[k] + 1u, ELAPSED_MAX);
                const uint32_t dn = ((c.max_steps - 1u - el[k]) >> 30) & (uint32_t)BPG_FLAG_DONE;   // el >= max_steps
                rb[k] = RBITS_FAIL ^ (okm & (RBITS_FAIL ^ RBITS_OK));
                fl[k] = (lat[k] & (uint32_t)BPG_FLAG_SUCCESS) | (okm & (uint32_t)BPG_FLAG_STEP_OK) | dn;
            }
            if (OUT & OUT_REWARD) st_vec<NT, EPL>(&rrow[(int64_t)t * ldo], rb);
            if (OUT & OUT_FLAGS) {
                uint32_t f = fl[0];
#pragma unroll
                for (int k = 1; k < EPL; k++) f |= fl[k] << (8 * k);
                st_1<NT>(&frow[(int64_t)t * ldo], (Fl)f);
            }
            // The contacts-in-kernel form is arithmetic-co-limited and lives on occupancy.  Left alone, the
            // scheduler interleaves the generators of all UNROLL steps (47 VGPRs in one build, 107-140 in the next,
            // depending on which side of the SGPR limit a few unrelated kernel arguments push it); a scheduling
            // fence per step keeps every step's ~70 instructions together: 50 VGPRs, 7-8 waves per SIMD.
            if (SRC == SRC_SYNTH) __builtin_amdgcn_sched_barrier(0);
        };
        auto flush_ok = [&](int t0) {
            if (OKB) {
#pragma unroll
                for (int k = 0; k < EPL; k++) {
                    okb[k] |= (uint64_t)okpart[k] << t0;
                    okpart[k] = 0u;
                }
            }
        };
        // Steps run in batches of UNROLL.  The masks of batch b+1 are requested BEFORE the rewards /
        // flags of batch b are stored: vector-memory operations retire in issue order (one vmcnt
        // counter for loads and stores), so a load issued behind a store cannot be consumed until
        // that store has been acknowledged by the memory side -- with the loads in front, a batch
        // waits for one memory latency instead of a store round trip plus a load round trip.
        const int nb = T / UNROLL;
        if constexpr (PIPE && SRC == SRC_MASKS) {
            Vec bufA[UNROLL], bufB[UNROLL];
            if (nb > 0) {
#pragma unroll
                for (int u = 0; u < UNROLL; u++) bufA[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)u * ldm]);
            }
            auto batch = [&](int t0, const Vec *cur, Vec *nxt, bool prefetch) {
                if (prefetch) {
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) nxt[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)(t0 + UNROLL + u) * ldm]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the requests in front of this batch's stores
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    uint32_t mt[EPL];
                    unpack<EPL>(cur[u], mt);
                    one_step(t0 + u, mt, 1u << u);
                }
                flush_ok(t0);
                if (t0 == 0 && tl_pass < 2) BPG_TL(1 + 4 * tl_pass);
            };
            // Two batches per trip with ping-pong register buffers (a register copy cur = nxt would
            // have to wait for the prefetch it copies), and the tail peeled so that every batch in
            // the loop prefetches unconditionally (a conditional prefetch makes the compiler place
            // the wait for the merged worst case at the loop head).
            int b = 0;
            for (; b + 2 < nb; b += 2) {
                batch(b * UNROLL, bufA, bufB, true);
                batch((b + 1) * UNROLL, bufB, bufA, true);
            }
            if (nb - b == 2) {
                batch(b * UNROLL, bufA, bufB, true);
                batch((b + 1) * UNROLL, bufB, bufA, false);
            } else if (nb - b == 1) {
                batch(b * UNROLL, bufA, bufB, false);
            }
        } else {
            for (int b = 0; b < nb; b++) {
                Vec cur[UNROLL];
                if (SRC == SRC_MASKS) {
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) cur[u] = ld_vec<NTL, EPL>(&mrow[(int64_t)(b * UNROLL + u) * ldm]);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    uint32_t mt[EPL] = {};
                    if (SRC == SRC_MASKS) unpack<EPL>(cur[u], mt);
                    one_step(b * UNROLL + u, mt, 1u << u);
                }
                flush_ok(b * UNROLL);
                if (b == 0 && tl_pass < 2) BPG_TL(1 + 4 * tl_pass);
            }
        }
        if (tl_pass < 2) BPG_TL(2 + 4 * tl_pass);
        for (int t = nb * UNROLL; t < T; t++) {          // T % UNROLL leftover steps
            uint32_t mt[EPL] = {};
            if (SRC == SRC_MASKS) unpack<EPL>(mrow[(int64_t)t * ldm], mt);
            one_step(t, mt, 1u << (t - nb * UNROLL));
        }
        flush_ok(nb * UNROLL);
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            an[k] = cm[k] | (el[k] << ELAPSED_SHIFT);
            ee[k] = ev[k] | (lat[k] & SUCC_BIT);
            nsucc += lat[k] & 1u;
        }
        // (plain stores: nontemporal state stores were measured 0.8 us SLOWER per 1 M-env launch, 84.8 against 84.0 us
        // on one box, profiles/r04_ab_state_stores.log -- the 8 MB a launch leaves dirty are not what the boundary costs)
        reinterpret_cast<Vec *>(now)[v] = pack<EPL>(an);
        reinterpret_cast<Vec *>(ever)[v] = pack<EPL>(ee);
        if constexpr (OKB) {
#pragma unroll
            for (int k = 0; k < EPL; k++) okbits[EPL * v + k] = okb[k];          // adjacent: one 8 * EPL-byte store
        }
        if constexpr (RET || OKB) {
            // steps of this rollout whose success condition did NOT hold, per env: T - (held steps)
            uint32_t nf[EPL];
#pragma unroll
            for (int k = 0; k < EPL; k++) nf[k] = (uint32_t)T - (OKB ? (uint32_t)__popcll(okb[k]) : nok[k]);
            if (ep_return) {                        // uniform (an OKB launch may or may not be asked for the return)
                // r[0] + r[1] + ... of T float32 rewards in {-1.0f, -0.0f}: -(number of -1.0f), -0.0f if there is none
                uint32_t rt[EPL];
#pragma unroll
                for (int k = 0; k < EPL; k++) rt[k] = __float_as_uint(-(float)nf[k]);
                reinterpret_cast<Vec *>(ep_return)[v] = pack<EPL>(rt);
            }
            if (ep_fail) {                          // the same number as one byte per env (the host guarantees T <= 255)
                uint32_t f = nf[0];
#pragma unroll
                for (int k = 1; k < EPL; k++) f |= nf[k] << (8 * k);
                reinterpret_cast<Fl *>(ep_fail)[v] = (Fl)f;
            }
        }
        if (tl_pass < 2) BPG_TL(3 + 4 * tl_pass);
        tl_pass++;
    }
    BPG_TL(11);
    if (ca.count) {
        if (MULTI) {
            block_count_add(nsucc, CountArgs{ca.count + ro, nullptr, nullptr});
            __syncthreads();                       // the partial sums are written again by the next rollout
        } else {
            block_count_add(nsucc, ca);
        }
    }
    BPG_TL(12);
    }
    BPG_TL(13);
}

// Launch shape of the fused rollout, chosen from the batch size (round 3's tools/tune_round3.hip mode A,
// profiles/r03_tune_mid_shapes.log; round 2: profiles/r02_tune_geometry.log; MI355X, T = 50, masks read cold
// from HBM).  The kernel is a pure stream with T independent requests per lane, so what matters is (a) enough
// resident waves on every CU to cover the HBM latency and (b), when there are not, that no wave waits for a
// store round trip before it can consume a load (the pipelined loop).  rocprofv3 means per launch
// (profiles/r03_pmc/kernel_means.json), round 1's plain 4/lane x 1024 shape in parentheses:
//     65 536      9.9 us  (49)   1/lane, 256-lane blocks, pipelined, 10 rows in flight
//     131 072    14.5 us  (45)   1/lane, 256, pipelined, 5 rows in flight
//     262 144    23.5 us  (52)   2/lane, 256, plain loop
//     524 288    43.8 us  (58)   2/lane, 1024, plain
//     1 048 576  82-91 us (89)   2/lane, 1024, plain (shifted rows, bpg_row_stride; box-dependent)
//     4 194 304  351 us   (372)  2/lane, 256, plain, when the rows are a power of two apart (1024 lanes: 468 us)
// 8-byte accesses (512 B per wave instruction) stream as fast as 16-byte ones here, so there is no
// 4-per-lane instantiation.  One returning atomic per block on a sharded counter (block_count_add).
constexpr int64_t ROLLOUT_UNROLL5_FROM = 81920, ROLLOUT_EPL2_FROM = 229376;
struct RolloutGeometry { int epl, block, grid, unroll; bool pipe; };

RolloutGeometry rollout_geometry(int64_t n, int64_t ld_masks, int64_t ld_out, const void *masks, const void *reward,
                                 const void *flags, const void *ep_return = nullptr, const void *ep_fail = nullptr, int64_t ld_env = 0)
{
    // 2 envs per lane need an even batch, even row strides and 8-byte aligned mask / reward streams
    // (2-byte flags)
    const bool ok2 = n % 2 == 0 && ld_masks % 2 == 0 && ld_out % 2 == 0 && ((uintptr_t)masks & 7u) == 0 &&
                     ((uintptr_t)reward & 7u) == 0 && ((uintptr_t)flags & 1u) == 0 && ((uintptr_t)ep_return & 7u) == 0 &&
                     ((uintptr_t)ep_fail & 1u) == 0 && ld_env % 2 == 0;
    RolloutGeometry g;
    g.unroll = 5;
    if (ok2 && n >= 131072 && !masks) {
        // contacts generated in-kernel: write-only and co-limited by the generator's arithmetic.  One
        // short-lived 256-lane block per 512 envs (no grid cap: blocks start in address order as
        // earlier ones retire), 10 steps per trip: 51 vs 53-58 us at 1 M envs, 178 vs 190-220 us at
        // 4 M (profiles/r02_tune_geometry_synth.log)
        g.epl = 2;
        g.block = 256;
        g.pipe = false;
        g.unroll = 10;
        g.grid = (int)((n / 2 + 255) / 256);
        return g;
    }
    if (ok2 && n >= ROLLOUT_EPL2_FROM) {
        // 2 envs per lane, plain loop.  From 524 288 envs on in 1024-lane blocks (one per CU, 16 waves cover
        // the round trips); below that -- and from 4 M envs on when the rows are a power of two apart -- in
        // 256-lane blocks.  (Round 2 ran a software-pipelined 2-per-lane loop here whose ping-pong buffers
        // spilled to scratch under the 1024-lane register bound it was compiled with; with an honest bound it
        // needs 200 VGPRs and is no faster than the plain loop: profiles/r03_tune_mid_shapes.log.)
        g.epl = 2;
        const bool aliased = ld_masks % 4096 == 0 || ld_out % 4096 == 0;
        const bool fat = n >= BPG_ROLLOUT_FAT_FROM && (n < 4194304 || !aliased);
        g.block = fat ? 1024 : 256;
        g.pipe = false;
    } else {
        // 1 env per lane, software-pipelined requests (latency-bound sizes): 10 rows in flight per lane below
        // ~100 K envs, 5 above (profiles/r03_tune_mid_shapes.log)
        g.epl = 1;
        g.block = 256;
        g.pipe = true;
        g.unroll = n < ROLLOUT_UNROLL5_FROM ? 10 : 5;
    }
    const int64_t lanes = n / g.epl;
    int64_t grid = (lanes + g.block - 1) / g.block;
    if (grid < 1) grid = 1;
    const int64_t cap = 256 * (g.block >= BPG_ROLLOUT_CU_LANES ? 1 : BPG_ROLLOUT_CU_LANES / g.block);   // at most 1024 lanes' worth of blocks per CU
    g.grid = (int)(grid > cap ? cap : grid);
    return g;
}


// ---- dispatch: one launcher per instantiation, looked up in constexpr tables -------------------------------------
// A launch form is (shape, OUT, MODE, AUX).  The six shapes are the only ones rollout_geometry() hands out; LB is the block
// size the shape is launched with.
template <int SRC_, int EPL_, int UNROLL_, bool PIPE_, int LB_> struct Shape {
    static constexpr int SRC = SRC_, EPL = EPL_, UNROLL = UNROLL_, LB = LB_;
    static constexpr bool PIPE = PIPE_;
};
using ShapeMasksFat = Shape<SRC_MASKS, 2, 5, false, 1024>;     // >= 524 288 envs
using ShapeMasksMid = Shape<SRC_MASKS, 2, 5, false, 256>;      // >= 229 376 envs (and aliased rows at >= 4 M)
using ShapeMasksPipe5 = Shape<SRC_MASKS, 1, 5, true, 256>;     // >= 81 920 envs, or an odd batch / unaligned streams
using ShapeMasksPipe10 = Shape<SRC_MASKS, 1, 10, true, 256>;   // below
using ShapeSynthWide = Shape<SRC_SYNTH, 2, 10, false, 256>;    // contacts generated in-kernel: nothing to request
using ShapeSynthNarrow = Shape<SRC_SYNTH, 1, 5, false, 256>;

struct RolloutLaunch {                 // everything a k_rollout launch takes, whatever its form
    dim3 grid, block;
    hipStream_t st;
    uint32_t *now, *ever;
    const uint32_t *masks;
    uint32_t *reward;
    uint8_t *flags;
    int64_t n, ldm, ldo;
    int T, rollouts;
    StepConst c;
    SynthConst s;
    ResetConst rc;
    CountArgs ca;
    uint64_t *okbits;
    uint32_t *ep_return;
    uint8_t *ep_fail;
    int64_t ld_env;
};
using RolloutLauncher = void (*)(const RolloutLaunch &);

template <class SH, int OUT, int MODE, int AUX> void launch_rollout_as(const RolloutLaunch &l)
{
    SynthArg<SH::SRC> s{};
    if constexpr (SH::SRC == SRC_SYNTH) s = l.s;
    ResetArg<MODE> rc{};
    if constexpr (MODE != MODE_CONT) rc = l.rc;
    hipLaunchKernelGGL((k_rollout<SH::SRC, SH::EPL, OUT, MODE, SH::UNROLL, SH::PIPE, SH::LB, AUX>), l.grid, l.block, 0, l.st,
                       l.now, l.ever, l.masks, l.reward, l.flags, l.n, l.T, l.ldm, l.ldo, l.c, s, rc, l.ca, l.okbits, l.rollouts,
                       l.ep_return, l.ep_fail, l.ld_env);
}

constexpr int N_OUT = 4, N_MODE = 3, N_AUX = 3, N_FORMS = N_OUT * N_MODE * N_AUX;
constexpr int form_index(int out, int mode, int aux) { return (out * N_MODE + mode) * N_AUX + aux; }

template <class SH, int I> constexpr RolloutLauncher form_launcher()
{
    constexpr int out = I / (N_MODE * N_AUX), mode = (I / N_AUX) % N_MODE, aux = I % N_AUX;
    if constexpr (mode == MODE_MULTI && aux == AUX_OKB) return nullptr;       // no step-ok words with several rollouts per launch
    else return &launch_rollout_as<SH, out, mode, aux>;
}

template <class SH> struct FormTable {
    RolloutLauncher at[N_FORMS];
    template <int... I> static constexpr FormTable make(std::integer_sequence<int, I...>) { return FormTable{{form_launcher<SH, I>()...}}; }
};
template <class SH> constexpr FormTable<SH> kForms = FormTable<SH>::make(std::make_integer_sequence<int, N_FORMS>());

RolloutLauncher rollout_launcher(const RolloutGeometry &geo, bool synth, int out, int mode, int aux)
{
    const int f = form_index(out, mode, aux);
    if (synth) return geo.epl == 2 ? kForms<ShapeSynthWide>.at[f] : kForms<ShapeSynthNarrow>.at[f];
    if (geo.epl == 2) return geo.block == 1024 ? kForms<ShapeMasksFat>.at[f] : kForms<ShapeMasksMid>.at[f];
    return geo.unroll == 5 ? kForms<ShapeMasksPipe5>.at[f] : kForms<ShapeMasksPipe10>.at[f];
}

}  // namespace bpgk

extern "C" {

int bpg_rollout_v2(bpg_batch *b, const bpg_rollout_args *args, void *stream)
{
    if (!b || !args) return fail(BPG_ERR_ARG, "NULL argument");
    // struct versioning (include/bpgym.h): fields the caller's header did not have are zero; bytes this library does
    // not know must be zero
    static_assert(offsetof(bpg_rollout_args, gamma) + sizeof(double) == BPG_ROLLOUT_ARGS_MIN_SIZE, "BPG_ROLLOUT_ARGS_MIN_SIZE");
    static_assert(sizeof(bpg_rollout_args) < 0x10000 && sizeof(bpg_rollout_args) % 8 == 0, "BPG_ARGS_TAG keeps the size in 16 bits");
    bpg_rollout_args local;
    if (int rc = take_args(local, args, BPG_ROLLOUT_ARGS_MIN_SIZE, "bpg_rollout_args")) return rc;
    const bpg_rollout_args *a = &local;
    if (a->T < 1 || a->T > (int)ELAPSED_MAX) return fail(BPG_ERR_ARG, "T must be in 1..1023");
    if ((a->d_success_count && ((uintptr_t)a->d_success_count & 7u)) ||
        (a->d_zero_count && ((uintptr_t)a->d_zero_count & 7u)))
        return fail(BPG_ERR_ARG, "counters must be 8-byte aligned");
    if (a->d_success_count && a->d_success_count == a->d_zero_count)
        return fail(BPG_ERR_ARG, "d_zero_count must differ from d_success_count");
    DeviceGuard guard(b->device);
    const StepConst c = step_const(b);
    const SynthConst s = synth_const(b);
    const ResetConst rc = reset_const(b, a->reset_first ? 1 : 0, a->d_num_grey);
    hipStream_t st = (hipStream_t)stream;
    uint32_t *rw = reinterpret_cast<uint32_t *>(a->d_reward_tb);
    const CountArgs ca = count_args(b, a->d_success_count, a->d_zero_count);
    const bool episode = a->d_ep_ag || a->d_ep_g || a->d_ep_success;
    const int64_t ldm = a->ld_masks ? a->ld_masks : b->n, ldo = a->ld_out ? a->ld_out : b->n;
    if (ldm < b->n || ldo < b->n) return fail(BPG_ERR_ARG, "ld_masks / ld_out must be 0 or >= n_envs");
    // The policy-gradient return (policy_gradient/rollout.py:255-258).  Up to 64 steps the step rewards of an
    // env are one 64-bit word: the rollout launch stores it (8 bytes per env, handle-owned) and a second launch
    // on the same stream expands the words into the G rows -- no reward stream is written or read.  Longer
    // horizons go through the reward stream.
    enum { G_NONE, G_OKBITS, G_FROM_REWARD } gmode = G_NONE;
    if (a->d_G) {
        if (!(a->gamma == a->gamma)) return fail(BPG_ERR_ARG, "gamma is NaN");
        if (((uintptr_t)a->d_G & 7u) != 0) return fail(BPG_ERR_ARG, "d_G must be 8-byte aligned");
        if (a->T <= 64 && gamma_is_subtractable(a->gamma, a->T)) gmode = G_OKBITS;
        else if (a->T > DR_TMAX) return fail(BPG_ERR_ARG, "d_G needs T <= 319");
        else if (!a->d_reward_tb)
            return fail(BPG_ERR_ARG, "d_G with T > 64, a negative gamma or overflowing powers of gamma needs d_reward_tb "
                                     "(the return is then computed from the reward stream)");
        else gmode = G_FROM_REWARD;
    }
    if (a->d_episode_return && ((uintptr_t)a->d_episode_return & 3u) != 0)
        return fail(BPG_ERR_ARG, "d_episode_return must be 4-byte aligned");
    if (a->ld_rollouts && a->ld_rollouts < b->n) return fail(BPG_ERR_ARG, "ld_rollouts must be 0 or >= n_envs");
    if (a->d_episode_fail_count && a->T > 255) return fail(BPG_ERR_ARG, "d_episode_fail_count is one byte per env: T <= 255");
    if (gmode == G_OKBITS && !b->d_okbits) {
        // the step-ok words are handle-owned scratch, allocated by the first call that asks for d_G: a hipMalloc is not
        // a stream operation, so that call must not sit inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone)
            return fail(BPG_ERR_ARG, "d_G needs the handle's step-ok words and a stream capture cannot allocate them: call "
                                     "bpg_reserve_return_scratch(handle) once before capturing");
        if (int rc2 = bpg_reserve_return_scratch(b)) return rc2;      // fallback: set-up code should have called it
    }
    uint64_t *okbits = gmode == G_OKBITS ? b->d_okbits : nullptr;
    // several rollouts in one launch (stream path only): rollout r = reset + T steps + count on rows r*T .. of the streams
    const int rollouts = a->rollouts > 1 ? a->rollouts : 1;
    if (a->rollouts < 0 || a->rollouts > 4096) return fail(BPG_ERR_ARG, "rollouts must be in 0..4096");
    if (rollouts > 1 && (!a->reset_first || episode || a->d_G))
        return fail(BPG_ERR_ARG, "rollouts > 1 needs reset_first and takes neither episode arrays nor d_G");
    int rcode = BPG_OK;
    if (!episode) {
        const int out = (rw ? OUT_REWARD : 0) | (a->d_flags_tb ? OUT_FLAGS : 0);
        const RolloutGeometry geo = rollout_geometry(b->n, ldm, ldo, a->d_masks, a->d_reward_tb, a->d_flags_tb,
                                                     a->d_episode_return, a->d_episode_fail_count, a->ld_rollouts);
        const bool per_env = a->d_episode_return || a->d_episode_fail_count;
        const int mode = rollouts > 1 ? MODE_MULTI : (a->reset_first ? MODE_RESET : MODE_CONT);
        // the in-loop counter only where a per-env count is asked for and there is no step-ok word to take it from
        const int aux = okbits ? AUX_OKB : (per_env ? AUX_RET : AUX_NONE);
        RolloutLaunch l;
        l.grid = dim3(geo.grid); l.block = dim3(geo.block); l.st = st;
        l.now = b->now; l.ever = b->ever; l.masks = a->d_masks; l.reward = rw; l.flags = a->d_flags_tb;
        l.n = b->n; l.ldm = ldm; l.ldo = ldo; l.T = a->T; l.rollouts = rollouts;
        l.c = c; l.s = s; l.rc = rc; l.ca = ca; l.okbits = okbits;
        l.ep_return = reinterpret_cast<uint32_t *>(a->d_episode_return); l.ep_fail = a->d_episode_fail_count;
        l.ld_env = a->ld_rollouts ? a->ld_rollouts : b->n;
        RolloutLauncher fn = rollout_launcher(geo, a->d_masks == nullptr, out, mode, aux);
        if (!fn) return fail(BPG_ERR_ARG, "rollouts > 1 takes no d_G");
        fn(l);
        rcode = check_launch("k_rollout");
    } else {
        if ((a->d_ep_ag && !aligned16(a->d_ep_ag)) || (a->d_ep_g && !aligned16(a->d_ep_g)) ||
            (a->d_ep_success && !aligned4(a->d_ep_success)))
            return fail(BPG_ERR_ARG, "episode buffers must be 16-byte aligned");
        EpisodeLaunch e;
        e.d_masks = a->d_masks; e.reward_tb = rw; e.flags_tb = a->d_flags_tb;
        e.ep_ag = a->d_ep_ag; e.ep_g = a->d_ep_g; e.ep_success = a->d_ep_success;
        e.T = a->T; e.ldm = ldm; e.ldo = ldo; e.rc = rc; e.ca = ca; e.okbits = okbits; e.ep_return = a->d_episode_return;
        e.ep_fail = a->d_episode_fail_count;
        rcode = launch_episode(b, e, st);
    }
    if (rcode == BPG_OK && gmode == G_OKBITS) rcode = launch_returns_from_okbits(b->d_okbits, a->d_G, b->n, a->T, a->gamma, st);
    if (rcode == BPG_OK && gmode == G_FROM_REWARD)
        rcode = bpg_discounted_return_ld(a->d_reward_tb, b->n, ldo, a->T, a->gamma, a->d_G, b->device, stream);
    if (rcode == BPG_OK && a->reset_first && b->info.is_variation && !a->d_num_grey && !b->parent) {
        rcode = enqueue_bump_counter(b->d_reset_counter, (uint32_t)rollouts, st);
    }
    return rcode;
}

int bpg_rollout_train(bpg_batch *const *handles, void *const *streams, int n_shards, const bpg_rollout_args *args,
                      int64_t args_stride_bytes, int n_launches, void *caller_stream)
{
    if (!handles || !streams || !args) return fail(BPG_ERR_ARG, "NULL argument");
    if (n_shards < 1 || n_shards > 64 || n_launches < 1) return fail(BPG_ERR_ARG, "n_shards must be in 1..64 and n_launches >= 1");
    if (args_stride_bytes < (int64_t)BPG_ROLLOUT_ARGS_MIN_SIZE || args_stride_bytes % 8 != 0)
        return fail(BPG_ERR_ARG, "args_stride_bytes must be the size of the caller's bpg_rollout_args");
    for (int s = 0; s < n_shards; s++) {
        if (!handles[s]) return fail(BPG_ERR_ARG, "a handle is NULL");
        if (handles[s]->device != handles[0]->device) return fail(BPG_ERR_ARG, "the shards of a train live on one device");
        for (int q = 0; q < s; q++)
            if (handles[q] == handles[s] || streams[q] == streams[s])
                return fail(BPG_ERR_ARG, "every shard needs a handle and a stream of its own");
    }
    DeviceGuard guard(handles[0]->device);
    hipStream_t cs = (hipStream_t)caller_stream;
    auto event_of = [](hipEvent_t &e) -> hipError_t { return e ? hipSuccess : hipEventCreateWithFlags(&e, hipEventDisableTiming); };
    // fork: the shards' streams start behind everything enqueued on the caller's stream so far (one event, n_shards waits)
    HIP_TRY(event_of(handles[0]->ev_fork));
    HIP_TRY(hipEventRecord(handles[0]->ev_fork, cs));
    for (int s = 0; s < n_shards; s++)
        if ((hipStream_t)streams[s] != cs) HIP_TRY(hipStreamWaitEvent((hipStream_t)streams[s], handles[0]->ev_fork, 0));
    // launch-major: launch l of every shard before launch l + 1 of any, so that no stream runs dry while the host is enqueuing
    const unsigned char *base = reinterpret_cast<const unsigned char *>(args);
    for (int l = 0; l < n_launches; l++)
        for (int s = 0; s < n_shards; s++) {
            const bpg_rollout_args *a = reinterpret_cast<const bpg_rollout_args *>(base + ((int64_t)l * n_shards + s) * args_stride_bytes);
            if (int rc = bpg_rollout_v2(handles[s], a, streams[s])) return rc;
        }
    // join: the caller's stream continues behind the last launch of every shard
    for (int s = 0; s < n_shards; s++) {
        if ((hipStream_t)streams[s] == cs) continue;
        HIP_TRY(event_of(handles[s]->ev_join));
        HIP_TRY(hipEventRecord(handles[s]->ev_join, (hipStream_t)streams[s]));
        HIP_TRY(hipStreamWaitEvent(cs, handles[s]->ev_join, 0));
    }
    return BPG_OK;
}

}  // extern "C"
