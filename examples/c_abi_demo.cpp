// Plain C++ host driving the C-ABI (include/bpgym.h) with nothing but the HIP runtime:
// what a non-Python binding (cgo / JNI / N-API ...) of the env boundary would do.
//   hipcc -O2 -std=c++17 -Iinclude -o /tmp/c_abi_demo examples/c_abi_demo.cpp \
//         -Lblockpuzzle-gym_amd/lib -lbpgym -Wl,-rpath,$PWD/blockpuzzle-gym_amd/lib && /tmp/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "bpgym.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != BPG_OK) { std::printf("%s failed: %d %s\n", #x, rc_, bpg_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int64_t B = 1000;
    const int variant = bpg_variant_from_id("BlocksTouch-v0");
    bpg_variant_info info;
    CHECK(bpg_get_variant_info(variant, &info));
    std::printf("%s: num_objs %d dimg %d dimo %d\n", info.env_id, info.num_objs, info.dimg, info.dimo);

    bpg_batch *h = nullptr;
    CHECK(bpg_create(&h, variant, B, 0, info.max_episode_steps, 1234, 0));

    // id2obj as fetch_env.py:284 builds it, from the scene's geom names
    const char *geoms[] = {"floor0", nullptr, "robot0:base_link", "robot0:r_gripper_finger_link",
                           "robot0:l_gripper_finger_link", "table", "object0", "object1"};
    int8_t id2obj[8];
    for (int i = 0; i < 8; i++) id2obj[i] = (int8_t)bpg_geom2objid(geoms[i]);
    CHECK(bpg_set_id2obj(h, 0, id2obj, 8));

    // every env: object0-object1 touch, finger touches object0, floor contact ignored
    const int maxcon = 4;
    std::vector<int16_t> con((size_t)B * maxcon * 2, 0);
    std::vector<int32_t> ncon((size_t)B, 3);
    for (int64_t b = 0; b < B; b++) {
        int16_t *c = &con[(size_t)b * maxcon * 2];
        c[0] = 6; c[1] = 7;       // object0, object1   -> the goal pair (2,3)
        c[2] = 3; c[3] = 6;       // finger, object0
        c[4] = 0; c[5] = 7;       // floor (None), object1 -> ignored
    }
    int16_t *d_con; int32_t *d_ncon; float *d_reward, *d_ag; uint8_t *d_flags;
    HIP(hipMalloc(&d_con, con.size() * 2)); HIP(hipMalloc(&d_ncon, ncon.size() * 4));
    HIP(hipMalloc(&d_reward, B * 4)); HIP(hipMalloc(&d_flags, B)); HIP(hipMalloc(&d_ag, B * info.dimg * 4));
    HIP(hipMemcpy(d_con, con.data(), con.size() * 2, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_ncon, ncon.data(), ncon.size() * 4, hipMemcpyHostToDevice));

    CHECK(bpg_reset(h, nullptr, nullptr, nullptr));
    CHECK(bpg_step_contacts(h, d_con, d_ncon, maxcon, d_reward, d_flags, nullptr));
    CHECK(bpg_get_achieved_goal_f32(h, d_ag, info.dimg, nullptr));
    HIP(hipDeviceSynchronize());

    std::vector<float> reward((size_t)B), ag((size_t)B * info.dimg);
    std::vector<uint8_t> flags((size_t)B);
    HIP(hipMemcpy(reward.data(), d_reward, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(flags.data(), d_flags, B, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ag.data(), d_ag, B * info.dimg * 4, hipMemcpyDeviceToHost));
    uint32_t rbits; std::memcpy(&rbits, &reward[B - 1], 4);
    std::printf("env %lld: reward bits 0x%08X (-0.0f on success), flags 0x%02X\n", (long long)(B - 1), rbits, flags[B - 1]);
    const float *m = &ag[(size_t)(B - 1) * info.dimg];
    for (int i = 0; i < info.side; i++) {
        for (int j = 0; j < info.side; j++) std::printf("%5.1f", m[i * info.side + j]);
        std::printf("\n");
    }
    int bad = 0;
    for (int64_t b = 0; b < B; b++) {
        std::memcpy(&rbits, &reward[b], 4);
        const float *a = &ag[(size_t)b * info.dimg];
        bad += rbits != 0x80000000u || flags[b] != (BPG_FLAG_SUCCESS | BPG_FLAG_STEP_OK);
        bad += a[2 * 4 + 3] != 1.f || a[3 * 4 + 2] != 1.f || a[0 * 4 + 2] != 1.f || a[1 * 4 + 3] != -1.f;
    }
    unsigned long long *d_cnt, cnt = 0;
    HIP(hipMalloc(&d_cnt, 8)); HIP(hipMemset(d_cnt, 0, 8));
    CHECK(bpg_success_count(h, d_cnt, nullptr));
    HIP(hipMemcpy(&cnt, d_cnt, 8, hipMemcpyDeviceToHost));
    std::printf("success count %llu of %lld, mismatches %d\n", cnt, (long long)B, bad);

    // A whole T-step rollout in one launch (rollout.py:100-153) over time-major (T, ld) streams: contact
    // masks in, rewards / flags out, once with contiguous rows (ld = B) and once with the row stride the
    // library recommends for its batch size (bpg_row_stride; here an arbitrary padded one, the batch is
    // small) -- the results must be identical.  Zero-initialised args: every optional field is "default".
    const int T = info.max_episode_steps;
    const int64_t ld = B + 24;
    uint32_t *d_m0, *d_m1; float *d_r0, *d_r1; uint8_t *d_f0, *d_f1;
    HIP(hipMalloc(&d_m0, (size_t)T * B * 4)); HIP(hipMalloc(&d_m1, (size_t)T * ld * 4));
    HIP(hipMalloc(&d_r0, (size_t)T * B * 4)); HIP(hipMalloc(&d_r1, (size_t)T * ld * 4));
    HIP(hipMalloc(&d_f0, (size_t)T * B)); HIP(hipMalloc(&d_f1, (size_t)T * ld));
    CHECK(bpg_synth_masks(h, 0, T, d_m0, nullptr));
    CHECK(bpg_synth_masks_ld(h, 0, T, d_m1, ld, nullptr));
    unsigned long long *d_c2; HIP(hipMalloc(&d_c2, 16)); HIP(hipMemset(d_c2, 0, 16));
    bpg_rollout_args ra = BPG_ROLLOUT_ARGS_INIT;      // struct_size = BPG_ARGS_TAG(sizeof(bpg_rollout_args)), every other field zero
    ra.T = T; ra.reset_first = 1; ra.d_masks = d_m0; ra.d_reward_tb = d_r0; ra.d_flags_tb = d_f0; ra.d_success_count = d_c2;
    CHECK(bpg_rollout_v2(h, &ra, nullptr));
    ra.d_masks = d_m1; ra.d_reward_tb = d_r1; ra.d_flags_tb = d_f1; ra.d_success_count = d_c2 + 1;
    ra.ld_masks = ld; ra.ld_out = ld;
    CHECK(bpg_rollout_v2(h, &ra, nullptr));
    HIP(hipDeviceSynchronize());
    std::vector<float> r0((size_t)T * B), r1((size_t)T * ld);
    std::vector<uint8_t> f0((size_t)T * B), f1((size_t)T * ld);
    unsigned long long c2[2];
    HIP(hipMemcpy(r0.data(), d_r0, r0.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r1.data(), d_r1, r1.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(f0.data(), d_f0, f0.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(f1.data(), d_f1, f1.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(c2, d_c2, 16, hipMemcpyDeviceToHost));
    int bad2 = (c2[0] != c2[1]);
    for (int t = 0; t < T; t++)
        for (int64_t b = 0; b < B; b++) {
            bad2 += std::memcmp(&r0[(size_t)t * B + b], &r1[(size_t)t * ld + b], 4) != 0;
            bad2 += f0[(size_t)t * B + b] != f1[(size_t)t * ld + b];
        }
    std::printf("fused rollout, T = %d: %llu successes; padded rows (ld = %lld; bpg_row_stride(2^20) = %lld) vs contiguous: %d mismatches\n",
                T, c2[0], (long long)ld, (long long)bpg_row_stride((int64_t)1 << 20), bad2);

    // The policy-gradient return of the same rollout from the same call (policy_gradient/rollout.py:255-258):
    // d_G + gamma, no reward stream needed -- checked here against the reference's loop on the rewards of the
    // first launch (float64, gamma ** k as C pow, products and sums rounded separately).
    double *d_G; HIP(hipMalloc(&d_G, (size_t)B * T * 8));
    const double gamma = 1.0 - 1.0 / T;
    ra = BPG_ROLLOUT_ARGS_INIT;
    // ... and every env's undiscounted episode return r[0] + ... + r[T-1] (what a multi-GPU job all-gathers, 4 bytes per
    // env) from the same launch: d_episode_return
    float *d_ret; HIP(hipMalloc(&d_ret, (size_t)B * 4));
    ra.T = T; ra.reset_first = 1; ra.d_masks = d_m0; ra.d_G = d_G; ra.gamma = gamma; ra.d_episode_return = d_ret;
    CHECK(bpg_rollout_v2(h, &ra, nullptr));
    HIP(hipDeviceSynchronize());
    std::vector<double> G((size_t)B * T), ref(T);
    std::vector<float> ret((size_t)B);
    HIP(hipMemcpy(G.data(), d_G, G.size() * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ret.data(), d_ret, ret.size() * 4, hipMemcpyDeviceToHost));
    int bad3 = 0;
    for (int64_t b = 0; b < B; b++) {
        float acc = r0[(size_t)b];                                               // starts from r[0]: -0.0f stays -0.0f
        for (int t = 1; t < T; t++) acc = acc + r0[(size_t)t * B + b];
        bad3 += std::memcmp(&acc, &ret[(size_t)b], 4) != 0;
        for (int t = 0; t < T; t++) {
            const double r = (double)r0[(size_t)t * B + b];
            ref[t] = r;
            for (int t_ = 0; t_ < t; t_++) {
                volatile double p = std::pow(gamma, (double)(t - t_)) * r;       // no fused multiply-add
                ref[t_] = ref[t_] + p;
            }
        }
        bad3 += std::memcmp(ref.data(), &G[(size_t)b * T], (size_t)T * 8) != 0;
    }
    std::printf("policy-gradient return and episode return from the rollout call: %d of %lld rows differ from the reference loops\n", bad3, (long long)B);

    // The evaluator's test loop (train.py:62-64: n_test_rollouts consecutive generate_rollouts calls, only the success
    // rates kept) as ONE launch: rollouts = K, masks of all K rollouts in one (K*T, n) stream, one counter word each.
    // Two fresh handles (BlocksTouch keeps its matrix across resets, quirk Q1: both must start from the same state):
    // one runs the K rollouts in one launch, the other as K launches -- the counts must agree.
    const int K = 4;
    bpg_batch *h2 = nullptr, *h3 = nullptr;
    CHECK(bpg_create(&h2, variant, B, 0, info.max_episode_steps, 1234, 0));
    CHECK(bpg_create(&h3, variant, B, 0, info.max_episode_steps, 1234, 0));
    uint32_t *d_mk; uint8_t *d_fk;
    unsigned long long *d_ck;
    HIP(hipMalloc(&d_mk, (size_t)K * T * B * 4)); HIP(hipMalloc(&d_fk, (size_t)K * T * B));
    HIP(hipMalloc(&d_ck, 2 * K * 8)); HIP(hipMemset(d_ck, 0, 2 * K * 8));
    CHECK(bpg_synth_masks(h2, 0, K * T, d_mk, nullptr));
    ra = BPG_ROLLOUT_ARGS_INIT;
    ra.T = T; ra.reset_first = 1; ra.rollouts = K; ra.d_masks = d_mk; ra.d_flags_tb = d_fk; ra.d_success_count = d_ck;
    CHECK(bpg_rollout_v2(h2, &ra, nullptr));
    for (int k = 0; k < K; k++) {
        ra = BPG_ROLLOUT_ARGS_INIT;
        ra.T = T; ra.reset_first = 1; ra.d_masks = d_mk + (size_t)k * T * B; ra.d_success_count = d_ck + K + k;
        CHECK(bpg_rollout_v2(h3, &ra, nullptr));
    }
    HIP(hipDeviceSynchronize());
    unsigned long long ck[2 * K];
    HIP(hipMemcpy(ck, d_ck, sizeof(ck), hipMemcpyDeviceToHost));
    int bad4 = 0;
    for (int k = 0; k < K; k++) bad4 += ck[k] != ck[K + k];
    std::printf("%d rollouts in one launch: successes %llu %llu %llu %llu; %d differ from %d separate launches\n", K, ck[0], ck[1], ck[2],
                ck[3], bad4, K);
    CHECK(bpg_destroy(h2));
    CHECK(bpg_destroy(h3));

    // BASELINE configs[3] in miniature, as a host that is not PyTorch runs it: the job's env range split into contiguous shards
    // (here two VIEWS of one batch on two streams of one GPU -- the same partition that spreads a job over the GPUs of a node,
    // where every rank holds one shard as a batch of its own created with env_id_base = its first env), every shard's rollout
    // launch writes its envs' episode fail counts (one byte per env), the library's own RCCL communicator all-gathers them on
    // the launch stream (one rank here: the gather is a copy) and the bytes are expanded to the float32 returns on the device.
    const int64_t Bj = 4096;
    bpg_batch *job = nullptr, *shard[2] = {nullptr, nullptr};
    CHECK(bpg_create(&job, bpg_variant_from_id("BlocksTouchVariation-v0"), Bj, 0, 50, 99, 0));
    CHECK(bpg_create_view(&shard[0], job, 0, Bj / 2));
    CHECK(bpg_create_view(&shard[1], job, Bj / 2, Bj / 2));
    hipStream_t st[2];
    HIP(hipStreamCreate(&st[0])); HIP(hipStreamCreate(&st[1]));
    uint8_t *d_fail, *d_all; float *d_ret_g, *d_ret_direct;
    HIP(hipMalloc(&d_fail, Bj)); HIP(hipMalloc(&d_all, Bj)); HIP(hipMalloc(&d_ret_g, Bj * 4)); HIP(hipMalloc(&d_ret_direct, Bj * 4));
    static bpg_rollout_args sa = BPG_ROLLOUT_ARGS_INIT_STATIC;              // the constant initialiser (C statics)
    bpg_rollout_args train[2] = {sa, sa};
    void *shard_streams[2] = {st[0], st[1]};
    for (int r = 0; r < 2; r++) {
        train[r].T = 50; train[r].reset_first = 1;                           // contacts from the synthetic source, num_grey from the generator
        train[r].d_episode_fail_count = d_fail + r * (Bj / 2);
        train[r].d_episode_return = d_ret_direct + r * (Bj / 2);
    }
    // fork from the default stream, one launch per shard on its own stream, join: ONE call
    CHECK(bpg_rollout_train(shard, shard_streams, 2, train, (int64_t)sizeof(bpg_rollout_args), 1, nullptr));
    CHECK(bpg_advance_generator(job, 1, 0, nullptr));                        // the views drew from the generator: once, on the parent
    unsigned char id[BPG_COMM_ID_BYTES];
    bpg_comm *comm = nullptr;
    int rc_comm = bpg_comm_unique_id(id);
    int bad5 = 0;
    if (rc_comm == BPG_OK) {
        CHECK(bpg_comm_create(&comm, 0, 1, id, 0));
        CHECK(bpg_allgather_u8(comm, d_fail, d_all, Bj, nullptr));
        CHECK(bpg_episode_return_from_fail_count(d_all, d_ret_g, Bj, 0, nullptr));
        HIP(hipDeviceSynchronize());
        std::vector<float> ra2((size_t)Bj), rb2((size_t)Bj);
        HIP(hipMemcpy(ra2.data(), d_ret_g, Bj * 4, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(rb2.data(), d_ret_direct, Bj * 4, hipMemcpyDeviceToHost));
        for (int64_t e = 0; e < Bj; e++) bad5 += std::memcmp(&ra2[e], &rb2[e], 4) != 0;
        std::printf("sharded rollout + library all-gather (world %d): %d of %lld gathered returns differ from the launch's own\n",
                    bpg_comm_world_size(comm), bad5, (long long)Bj);
        CHECK(bpg_comm_destroy(comm));
    } else {
        std::printf("librccl.so not available (%s): gather skipped\n", bpg_last_error());
    }
    CHECK(bpg_destroy(shard[0])); CHECK(bpg_destroy(shard[1]));

    // the curriculum lives in the handle (fetch_env.py:623-630 for this variant): obj_range 0.08 -> 0.2 in steps of 0.025
    int level_at_top = -1;
    for (int i = 0; i < 12; i++) {
        int top = 0;
        CHECK(bpg_increase_difficulty(job, &top));
        if (top && level_at_top < 0) level_at_top = bpg_get_difficulty(job);
    }
    bpg_curriculum cur;
    CHECK(bpg_get_curriculum(job, &cur));
    std::printf("curriculum: difficulty %d at the maximum level, obj_range %.3f of %.3f\n", level_at_top, cur.obj_range, cur.max_obj_range);
    const int bad6 = !(level_at_top == 4 && cur.obj_range == cur.max_obj_range && bpg_get_difficulty(job) == 4);
    bpg_batch *gt = nullptr;
    CHECK(bpg_create(&gt, bpg_variant_from_id("GripperTouch-v0"), 4, 0, 50, 0, 0));
    const int rc_gt = bpg_increase_difficulty(gt, nullptr);                  // NotImplementedError in the reference (fetch_env.py:93-94)
    CHECK(bpg_destroy(gt));
    CHECK(bpg_destroy(job));
    CHECK(bpg_destroy(h));
    return (bad == 0 && cnt == (unsigned long long)B && bad2 == 0 && bad3 == 0 && bad4 == 0 && bad5 == 0 && bad6 == 0 &&
            rc_gt == BPG_ERR_UNSUPPORTED) ? 0 : 2;
}
