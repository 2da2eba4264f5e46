// Standalone measurement harness of round 4 (not part of the product): the PER-STEP path at small and large batches.
// Includes the library source and replays, from a hipGraph, the train of T step launches a rollout worker with a policy
// in the loop issues (RolloutStudent.generate_rollouts, rollout.py:100-153: one bpg_step_record launch per time step,
// achieved_goal rows into time slot t+1 of a time-major (T+1, B, 36) slab, is_success into (T, B), actions copied),
// and says where a step's time goes:
//   * per replay: HIP events around the graph launch -> us per step;
//   * per node: entry / exit time of every block (constant-rate 100 MHz clock, lane 0 of each block; the exit is taken
//     after the block's stores have been acknowledged), from which: ramp = last entry - first entry, duration = last
//     exit - first entry, gap = first entry of node t+1 - last exit of node t;
//   * the same train with an EMPTY kernel of the same grid (what T dependent launches cost with nothing to do);
//   * variants: waves per block (fatter, fewer blocks), grid capped at the CU count, contacts from memory / in-kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tools/tune_round4 tools/tune_round4.hip
//   ./tools/tune_round4 65536 1048576
#define BPG_TIMELINE 1
#include "../blockpuzzle-gym_amd/csrc/bpgym.hip"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(unsigned long long *tl, int node)
{
    if (threadIdx.x == 0 && tl) {
        tl[((size_t)node * gridDim.x + blockIdx.x) * 2] = wall_clock64();
        tl[((size_t)node * gridDim.x + blockIdx.x) * 2 + 1] = wall_clock64();
    }
}

struct Train {
    int64_t B;
    int T;
    uint32_t *now, *ever, *masks;
    float *ag, *succ, *u_in, *u_out, *reward;
    uint8_t *flags;
    unsigned long long *tl, *cnt, *shards;
};

static StepConst mk_c()
{
    StepConst c{};
    c.valid = 0x1FFFFF; c.want_touch = 1u << 12; c.want_never = 0; c.max_steps = 50; c.sticky_on = 0;
    return c;
}
static SynthConst mk_s()
{
    SynthConst s{};
    s.seed0 = 1;
    for (int n = 0; n < 8; n++) { s.valid_by_n[n] = 0x1FFFFF; s.forced_by_n[n] = 0; }
    return s;
}

static PairTable mk_tab()
{
    bpg_variant_info info;
    bpg_get_variant_info(BPG_BLOCKS_TOUCH_VARIATION, &info);
    return pair_table(info);
}

template <int WPB, int SRC, bool NT = false>
static void launch_step(const Train &t, int step, int grid, hipStream_t st, bool count)
{
    const int64_t slot = t.B * 36;
    CountArgs ca{count ? t.cnt : nullptr, nullptr, t.shards};
    hipLaunchKernelGGL((k_step_record<6, SRC, true, WPB, NT>), dim3(grid), dim3(64 * WPB), 0, st, t.now, t.ever,
                       SRC == SRC_MASKS ? t.masks + (int64_t)step * t.B : nullptr, reinterpret_cast<uint32_t *>(t.reward), t.flags,
                       t.succ + (int64_t)step * t.B, t.ag + (int64_t)(step + 1) * slot, (int64_t)36, t.B, mk_c(), mk_s(), mk_tab(),
                       reinterpret_cast<const float4 *>(t.u_in), reinterpret_cast<float4 *>(t.u_out) + (int64_t)step * t.B, ca, (float *)nullptr);
}

struct NodeStats { double ramp, dur, gap; };

static void summarize(const char *name, const std::vector<unsigned long long> &h, int T, int grid, float us_per_replay)
{
    std::vector<double> ramp, dur, gap, period;
    unsigned long long prev_exit = 0, prev_entry = 0;
    for (int n = 0; n <= T; n++) {                         // (a train writes time slots 1 .. T: node 0 stays empty)
        unsigned long long e0 = ~0ull, e1 = 0, x1 = 0;
        for (int g = 0; g < grid; g++) {
            const unsigned long long en = h[((size_t)n * grid + g) * 2], ex = h[((size_t)n * grid + g) * 2 + 1];
            if (!en) continue;
            e0 = std::min(e0, en); e1 = std::max(e1, en); x1 = std::max(x1, ex);
        }
        if (!e1) continue;
        ramp.push_back((double)(e1 - e0) / 100.0);
        dur.push_back((double)(x1 - e0) / 100.0);
        if (prev_entry) { gap.push_back(((double)e0 - (double)prev_exit) / 100.0); period.push_back((double)(e0 - prev_entry) / 100.0); }
        prev_exit = x1; prev_entry = e0;
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
    printf("  %-58s %7.2f us/step by events | per node (median; max): period %5.2f; %5.2f  first entry -> last exit %5.2f; %5.2f  "
           "ramp %5.2f; %5.2f  gap to next node %5.2f; %5.2f\n", name, us_per_replay / T, med(period), mx(period), med(dur), mx(dur),
           med(ramp), mx(ramp), med(gap), mx(gap));
}

template <class F>
static int replay(const char *name, Train &t, int grid, F &&enqueue_train, bool timeline = true)
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t graph;
    hipGraphExec_t exec;
    // two warm-up passes outside the capture (code objects loaded, clocks up)
    enqueue_train(st); enqueue_train(st);
    CK(hipStreamSynchronize(st));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue_train(st);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *null_tl = nullptr;
    std::vector<float> times;
    // (1) without the timestamps: the figure
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_tl_step), &null_tl, sizeof(null_tl)));
    for (int r = 0; r < 30; r++) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 5) times.push_back(ms * 1e3f);
    }
    std::sort(times.begin(), times.end());
    const float us = times[times.size() / 2];
    // (2) with them: the anatomy
    std::vector<unsigned long long> h((size_t)(t.T + 1) * grid * 2, 0ull);
    if (timeline) {
        CK(hipMemset(t.tl, 0, h.size() * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_tl_step), &t.tl, sizeof(t.tl)));
        for (int r = 0; r < 3; r++) { CK(hipGraphLaunch(exec, st)); }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), t.tl, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_tl_step), &null_tl, sizeof(null_tl)));
        summarize(name, h, t.T, grid, us);
    } else {
        printf("  %-58s %7.2f us/step by events (min %.2f, max %.2f over 25 replays)\n", name, us / t.T, times.front() / t.T, times.back() / t.T);
    }
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
    return 0;
}

static int run(int64_t B)
{
    Train t{};
    t.B = B; t.T = 50;
    const int T = t.T;
    CK(hipMalloc(&t.now, B * 4)); CK(hipMalloc(&t.ever, B * 4)); CK(hipMalloc(&t.masks, (size_t)B * T * 4));
    CK(hipMalloc(&t.ag, (size_t)B * 36 * 4 * (T + 1))); CK(hipMalloc(&t.succ, (size_t)B * T * 4));
    CK(hipMalloc(&t.u_in, B * 16)); CK(hipMalloc(&t.u_out, (size_t)B * 16 * T)); CK(hipMalloc(&t.reward, B * 4)); CK(hipMalloc(&t.flags, B));
    const int max_grid = (int)((B + 63) / 64);
    CK(hipMalloc(&t.tl, (size_t)(T + 1) * max_grid * 2 * 8));
    CK(hipMalloc(&t.cnt, 16)); CK(hipMemset(t.cnt, 0, 16));
    CK(hipMalloc(&t.shards, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8)); CK(hipMemset(t.shards, 0, COUNT_SHARDS * COUNT_SHARD_STRIDE * 8));
    CK(hipMemset(t.u_in, 0, B * 16));
    hipLaunchKernelGGL(k_init, dim3(2048), dim3(256), 0, 0, t.now, t.ever, B, 6u);
    for (int s = 0; s < T; s++)
        hipLaunchKernelGGL(k_synth_masks, dim3(2048), dim3(256), 0, 0, t.ever, B, s, 1, t.masks + (int64_t)s * B, B, mk_s());
    CK(hipDeviceSynchronize());
    const float *agb = t.ag;
    const long long slot = B * 36;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_tl_ag_base), &agb, sizeof(agb)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bpgk::bpg_tl_ag_slot), &slot, sizeof(slot)));
    const double bytes = (double)B * (12 + 13 + 4 + 144 + 32);
    printf("== B = %lld envs, T = %d step launches per replay; %.1f MB per step (205 B per env-step with masks + action copy)\n",
           (long long)B, T, bytes / 1e6);

    const int tiles = (int)((B + 63) / 64);
    // what T dependent launches cost with nothing to do
    {
        const int grid = tiles;
        unsigned long long *tl = t.tl;
        if (replay("empty kernel, one 64-lane block per 64 envs", t, grid, [&](hipStream_t st) {
                for (int s = 0; s < T; s++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, st, tl, s + 1);
            })) return 1;
        if (replay("empty kernel, one block", t, 1, [&](hipStream_t st) {
                for (int s = 0; s < T; s++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, tl, s + 1);
            })) return 1;
    }
#define VARIANT_NT(label, WPB, SRC, GRID, COUNT, NT)                                                                   \
    do {                                                                                                              \
        const int grid_ = (GRID);                                                                                     \
        if (replay(label, t, grid_, [&](hipStream_t st) {                                                             \
                for (int s = 0; s < T; s++) launch_step<WPB, SRC, NT>(t, s, grid_, st, (COUNT) && s == T - 1);        \
            })) return 1;                                                                                             \
    } while (0)
#define VARIANT(label, WPB, SRC, GRID, COUNT) VARIANT_NT(label, WPB, SRC, GRID, COUNT, false)
    VARIANT_NT("masks from memory, 1 wave / block, nontemporal output stores", 1, SRC_MASKS, tiles, false, true);
    VARIANT_NT("masks from memory, 4 waves / block, nontemporal output stores", 4, SRC_MASKS, (tiles + 3) / 4, false, true);
    VARIANT_NT("in-kernel contacts, 1 wave / block, nontemporal output stores", 1, SRC_SYNTH, tiles, false, true);
    VARIANT("masks from memory, 1 wave / block (round 3's shape)", 1, SRC_MASKS, tiles, false);
    VARIANT("masks from memory, 2 waves / block", 2, SRC_MASKS, (tiles + 1) / 2, false);
    VARIANT("masks from memory, 4 waves / block", 4, SRC_MASKS, (tiles + 3) / 4, false);
    VARIANT("masks from memory, 8 waves / block", 8, SRC_MASKS, (tiles + 7) / 8, false);
    VARIANT("masks from memory, 4 waves / block, grid <= 1024", 4, SRC_MASKS, std::min((tiles + 3) / 4, 1024), false);
    VARIANT("masks from memory, 4 waves / block, grid <= 512", 4, SRC_MASKS, std::min((tiles + 3) / 4, 512), false);
    VARIANT("masks from memory, 4 waves / block, grid <= 256", 4, SRC_MASKS, std::min((tiles + 3) / 4, 256), false);
    VARIANT("masks from memory, 8 waves / block, grid <= 256", 8, SRC_MASKS, std::min((tiles + 7) / 8, 256), false);
    VARIANT("in-kernel contacts, 1 wave / block", 1, SRC_SYNTH, tiles, false);
    VARIANT("in-kernel contacts, 4 waves / block", 4, SRC_SYNTH, (tiles + 3) / 4, false);
    VARIANT("masks, 4 waves / block + success count in the last step", 4, SRC_MASKS, (tiles + 3) / 4, true);
    // the success count as a launch of its own behind the train (round 3)
    {
        const int grid = (tiles + 3) / 4;
        if (replay("masks, 4 waves / block + k_success_count launch", t, grid, [&](hipStream_t st) {
                for (int s = 0; s < T; s++) launch_step<4, SRC_MASKS>(t, s, grid, st, false);
                hipLaunchKernelGGL(k_success_count, dim3(fat_grid((t.B + 3) / 4)), dim3(1024), 0, st, t.ever, t.B, CountArgs{t.cnt, nullptr, t.shards});
            }, false)) return 1;
    }
    (void)hipFree(t.now); (void)hipFree(t.ever); (void)hipFree(t.masks); (void)hipFree(t.ag); (void)hipFree(t.succ);
    (void)hipFree(t.u_in); (void)hipFree(t.u_out); (void)hipFree(t.reward); (void)hipFree(t.flags); (void)hipFree(t.tl);
    (void)hipFree(t.cnt); (void)hipFree(t.shards);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: %s <envs> [<envs> ...]\n", argv[0]); return 1; }
    for (int i = 1; i < argc; i++)
        if (run(atoll(argv[i]))) return 1;
    return 0;
}
