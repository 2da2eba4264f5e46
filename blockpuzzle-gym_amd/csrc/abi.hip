// abi.hip -- the handle and everything of include/bpgym.h that is not a kernel of its own: error text, variant
// table (tasks.py presets, constant colourings, goal planes), create / destroy / seed / id2obj, state pointers.
#include "state.hip.h"

namespace bpgk {

thread_local std::string g_err;

}  // namespace bpgk

int bpg_fail_(int code, const std::string &msg)
{
    bpgk::g_err = msg;
    return code;
}

namespace {

// ---------------------------------------------------------------------------
// variant table (host): tasks.py presets + constant colourings + goal planes
// ---------------------------------------------------------------------------
enum { GREY = 0, RED = 1, GREEN = 2, BLUE = 3 };   // fetch_env.py:11-14

struct VariantDef {
    const char *id;
    int num_objs;
    int colors[BPG_MAX_OBJS];
    int block_gripper;       // tasks.py
    int set_test;            // fetch_env.py:100-101 raises unless overridden
    int incr_status;         // quirk Q6
};

// the curriculum attributes each constructor sets (tasks.py passes obj_range = 0.15 to every preset; the curriculum
// presets then overwrite it)
bpg_curriculum initial_curriculum(int v)
{
    bpg_curriculum c;
    std::memset(&c, 0, sizeof(c));
    c.obj_range = 0.15;
    switch (v) {
    case BPG_BLOCKS_TOUCH:                       // fetch_env.py:346-348 (curriculum = False)
        c.max_obj_range = c.obj_range; c.obj_range_step = 0; c.has_step = 1;
        break;
    case BPG_BLOCKS_TOUCH_CURRICULUM:            // :342-345
    case BPG_BLOCKS_TOUCH_VARIATION:             // :561-563
        c.obj_range = 0.08; c.obj_range_step = 0.025; c.max_obj_range = 0.2; c.has_step = 1;
        break;
    case BPG_BLOCKS_TOUCH_CHOOSE:                // :413-415: no *_step attributes (quirk Q6)
        c.wrong_obj_range = 0; c.max_obj_range = 0.2;
        break;
    case BPG_BLOCKS_TOUCH_CHOOSE_CURRICULUM:     // :407-412
        c.obj_range = 0.08; c.obj_range_step = 0.025; c.wrong_obj_range = 0.2; c.wrong_obj_range_step = 0.02;
        c.max_obj_range = 0.3; c.has_step = 1;
        break;
    default:                                     // GripperTouch, ToppleTower: obj_range only
        break;
    }
    return c;
}

const VariantDef kVariants[BPG_NUM_VARIANTS] = {
    // fetch_env.py:323-326, tasks.py:4-20
    {"GripperTouch-v0", 3, {BLUE, GREY, GREEN, 0, 0, 0}, 0, 0, BPG_ERR_UNSUPPORTED},
    // fetch_env.py:360-363, tasks.py:22-40
    {"BlocksTouch-v0", 4, {GREY, GREY, GREEN, BLUE, 0, 0}, 0, 1, BPG_OK},
    // fetch_env.py:772-775, tasks.py:126-144
    {"ToppleTower-v0", 6, {RED, GREEN, GREY, GREY, GREY, BLUE}, 0, 0, BPG_ERR_UNSUPPORTED},
    // tasks.py:42-60 (curriculum=True, block_gripper=True)
    {"BlocksTouchCurriculum-v0", 4, {GREY, GREY, GREEN, BLUE, 0, 0}, 1, 1, BPG_OK},
    // fetch_env.py:434-441, tasks.py:62-81; curriculum=False lacks obj_range_step (:413-415)
    {"BlocksTouchChoose-v0", 5, {GREY, GREY, GREEN, BLUE, GREY, 0}, 1, 1, BPG_ERR_ATTRIBUTE},
    {"BlocksTouchChooseCurriculum-v0", 5, {GREY, GREY, GREEN, BLUE, GREY, 0}, 1, 1, BPG_OK},
    // fetch_env.py:632-639 (max_num_blocks = 4, :523), tasks.py:104-124
    {"BlocksTouchVariation-v0", 6, {GREY, GREY, GREEN, BLUE, GREY, GREY}, 1, 1, BPG_OK},
};

void fill_variant_info(int v, bpg_variant_info *o)
{
    const VariantDef &d = kVariants[v];
    std::memset(o, 0, sizeof(*o));
    o->variant = v;
    o->num_objs = d.num_objs;
    o->side = d.num_objs;
    o->dimg = o->side * o->side;
    o->is_variation = (v == BPG_BLOCKS_TOUCH_VARIATION);
    // fetch_env.py:187-228: 10 env features + 15 per block; Variation :567-621:
    // 1 + 10 + 4 * (15 + 4)
    o->dimo = o->is_variation ? 87 : 10 + 15 * (d.num_objs - 2);
    o->dimu = 4;
    o->block_gripper = d.block_gripper;
    o->max_episode_steps = 50;
    o->set_test_supported = d.set_test;
    o->increase_difficulty_status = d.incr_status;
    o->desired_goal_is_float = o->is_variation;
    std::snprintf(o->env_id, sizeof(o->env_id), "%s", d.id);
    for (int i = 0; i < BPG_MAX_OBJS; i++)
        o->obj_colors[i] = d.colors[i];
    // _sample_goal, fetch_env.py:260-273
    for (int i = 0; i < o->side; i++) {
        for (int j = 0; j < o->side; j++) {
            int ci = d.colors[i], cj = d.colors[j], g = 0;
            if ((ci == RED && cj == BLUE) || (ci == BLUE && cj == RED))
                g = -1;
            else if ((ci == GREEN && cj == BLUE) || (ci == BLUE && cj == GREEN))
                g = 1;
            o->goal[i * o->side + j] = g;
            uint32_t bit = 1u << host_pair_index(i, j);
            if (g == 1) o->want_touch |= bit;
            if (g == -1) o->want_never |= bit;
            o->valid_pairs |= bit;
        }
    }
}


}  // namespace

namespace bpgk {

__global__ void __launch_bounds__(256)
k_init(uint32_t *__restrict__ now, uint32_t *__restrict__ ever, int64_t n, uint32_t nobj)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        now[i] = 0u;                      // fetch_env.py:78 matrix all -1
        ever[i] = nobj << NOBJ_SHIFT;     // :80 has_succeeded False
    }
}

// Enqueued right after every launch that reset the batch: stream order makes all lanes of the
// reset see the old value and the next reset the new one, also under hipGraph replay.
__global__ void k_bump_counter(uint32_t *counter, uint32_t by) { *counter += by; }

int enqueue_bump_counter(uint32_t *d_counter, uint32_t by, hipStream_t st)
{
    hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, st, d_counter, by);
    return check_launch("k_bump_counter");
}

}  // namespace bpgk

extern "C" {

const char *bpg_last_error(void) { return g_err.c_str(); }
int bpg_version(void) { return BPG_LIBRARY_VERSION; }

int bpg_variant_from_id(const char *env_id)
{
    if (!env_id) return -1;
    for (int v = 0; v < BPG_NUM_VARIANTS; v++)
        if (std::strcmp(env_id, kVariants[v].id) == 0) return v;
    return -1;
}

int bpg_get_variant_info(int variant, bpg_variant_info *out)
{
    if (variant < 0 || variant >= BPG_NUM_VARIANTS || !out) return fail(BPG_ERR_ARG, "bad variant");
    fill_variant_info(variant, out);
    return BPG_OK;
}

int bpg_geom2objid(const char *name)
{
    // fetch_env.py:106-117
    if (name) {
        if (std::strstr(name, "finger")) return 0;
        if (std::strcmp(name, "table") == 0) return 1;
        if (std::strstr(name, "object")) {
            const char *p = name + (std::strlen(name) < 6 ? std::strlen(name) : 6);
            return std::atoi(p) + 2;
        }
    }
    return -1;
}

int bpg_pair_index(int i, int j)
{
    if (i < 0 || j < 0 || i >= BPG_MAX_OBJS || j >= BPG_MAX_OBJS) return -1;
    return host_pair_index(i, j);
}

int bpg_create(bpg_batch **out, int variant, int64_t n_envs, int device_id, int max_episode_steps,
               uint64_t seed, int64_t env_id_base)
{
    if (!out) return fail(BPG_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (variant < 0 || variant >= BPG_NUM_VARIANTS) return fail(BPG_ERR_ARG, "unknown variant");
    if (n_envs < 1) return fail(BPG_ERR_ARG, "n_envs must be >= 1");
    if (max_episode_steps < 1 || max_episode_steps > (int)ELAPSED_MAX)
        return fail(BPG_ERR_ARG, "max_episode_steps must be in 1..1023");
    DeviceGuard guard(device_id);
    if (guard.err != hipSuccess)
        return fail(BPG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    bpg_batch *b = new (std::nothrow) bpg_batch();
    if (!b) return fail(BPG_ERR_ARG, "out of host memory");
    fill_variant_info(variant, &b->info);
    b->n = n_envs;
    b->device = device_id;
    b->max_steps = max_episode_steps;
    b->info.max_episode_steps = max_episode_steps;
    b->seed = seed;
    b->env_base = env_id_base;
    b->d_reset_counter = nullptr;
    b->now = b->ever = nullptr;
    b->d_id2obj = nullptr;
    b->d_count_shards = nullptr;
    b->d_okbits = nullptr;
    b->cur = initial_curriculum(variant);
    b->parent = nullptr;
    b->ev_fork = b->ev_join = nullptr;
    std::memset(b->ngeom, 0, sizeof(b->ngeom));
    const size_t bytes = (size_t)((n_envs + 3) & ~(int64_t)3) * sizeof(uint32_t);
    hipError_t e1 = hipMalloc(&b->now, bytes), e2 = hipMalloc(&b->ever, bytes);
    hipError_t e3 = hipMalloc(&b->d_id2obj, BPG_MAX_MODELS * BPG_MAX_GEOMS);
    if (e3 == hipSuccess) e3 = hipMalloc(&b->d_reset_counter, 2 * sizeof(uint32_t));
    const size_t shard_bytes = (size_t)COUNT_SHARDS * COUNT_SHARD_STRIDE * sizeof(unsigned long long);
    if (e3 == hipSuccess) e3 = hipMalloc(&b->d_count_shards, shard_bytes);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        bpg_destroy(b);
        return fail(BPG_ERR_HIP, "hipMalloc failed for the packed state");
    }
    hipError_t e4 = hipMemset(b->d_id2obj, 0xFF, BPG_MAX_MODELS * BPG_MAX_GEOMS);
    if (e4 == hipSuccess) e4 = hipMemset(b->d_reset_counter, 0, 2 * sizeof(uint32_t));
    if (e4 == hipSuccess) e4 = hipMemset(b->d_count_shards, 0, shard_bytes);
    hipLaunchKernelGGL(k_init, dim3(grid_for(n_envs, 256)), dim3(256), 0, 0, b->now, b->ever, n_envs,
                       (uint32_t)b->info.num_objs);
    hipError_t e5 = hipDeviceSynchronize();
    if (e4 != hipSuccess || e5 != hipSuccess) {
        bpg_destroy(b);
        return fail(BPG_ERR_HIP, "state initialisation failed");
    }
    *out = b;
    return BPG_OK;
}

int bpg_create_view(bpg_batch **out, bpg_batch *parent, int64_t first_env, int64_t n_envs)
{
    if (!out) return fail(BPG_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!parent) return fail(BPG_ERR_ARG, "parent is NULL");
    if (parent->parent) return fail(BPG_ERR_ARG, "views of views are not supported: take the view of the parent");
    if (first_env < 0 || n_envs < 1 || first_env + n_envs > parent->n) return fail(BPG_ERR_ARG, "the view's range is not inside the parent");
    if (first_env % 4 != 0) return fail(BPG_ERR_ARG, "first_env must be a multiple of 4 (16-byte aligned state slices)");
    DeviceGuard guard(parent->device);
    bpg_batch *b = new (std::nothrow) bpg_batch(*parent);
    if (!b) return fail(BPG_ERR_ARG, "out of host memory");
    b->parent = parent;
    b->ev_fork = b->ev_join = nullptr;
    b->n = n_envs;
    b->env_base = parent->env_base + first_env;
    b->now = parent->now + first_env;
    b->ever = parent->ever + first_env;
    b->d_okbits = nullptr;
    b->d_count_shards = nullptr;
    const size_t shard_bytes = (size_t)COUNT_SHARDS * COUNT_SHARD_STRIDE * sizeof(unsigned long long);
    if (hipMalloc(&b->d_count_shards, shard_bytes) != hipSuccess || hipMemset(b->d_count_shards, 0, shard_bytes) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        if (b->d_count_shards) (void)hipFree(b->d_count_shards);
        delete b;
        return fail(BPG_ERR_HIP, "hipMalloc failed for the view's scratch");
    }
    *out = b;
    return BPG_OK;
}

int bpg_advance_generator(bpg_batch *b, uint32_t resets, uint32_t placements, void *stream)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (b->parent) return fail(BPG_ERR_ARG, "the generator's stream indices belong to the parent handle");
    DeviceGuard guard(b->device);
    if (resets)
        if (int rc = enqueue_bump_counter(b->d_reset_counter, resets, (hipStream_t)stream)) return rc;
    if (placements)
        if (int rc = enqueue_bump_counter(b->d_reset_counter + 1, placements, (hipStream_t)stream)) return rc;
    return BPG_OK;
}

int bpg_destroy(bpg_batch *b)
{
    if (!b) return BPG_OK;
    DeviceGuard guard(b->device);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->parent) {                       // a view owns its scratch only
        if (b->d_count_shards) (void)hipFree(b->d_count_shards);
        if (b->d_okbits) (void)hipFree(b->d_okbits);
        delete b;
        return BPG_OK;
    }
    if (b->now) (void)hipFree(b->now);
    if (b->ever) (void)hipFree(b->ever);
    if (b->d_id2obj) (void)hipFree(b->d_id2obj);
    if (b->d_reset_counter) (void)hipFree(b->d_reset_counter);
    if (b->d_count_shards) (void)hipFree(b->d_count_shards);
    if (b->d_okbits) (void)hipFree(b->d_okbits);
    delete b;
    return BPG_OK;
}

int64_t bpg_num_envs(const bpg_batch *b) { return b ? b->n : 0; }

int bpg_seed(bpg_batch *b, uint64_t seed)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (b->parent) return fail(BPG_ERR_ARG, "bpg_seed: a view shares its parent's seed -- seed the parent");
    DeviceGuard guard(b->device);
    // Synchronous by contract: launches in flight on ANY stream (torch side streams do not order
    // against the null stream) still read the old key and stream indices.
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(b->d_reset_counter, 0, 2 * sizeof(uint32_t)));
    HIP_TRY(hipDeviceSynchronize());
    b->seed = seed;
    return BPG_OK;
}

int bpg_set_id2obj(bpg_batch *b, int model_idx, const int8_t *id2obj, int ngeom)
{
    if (!b || !id2obj) return fail(BPG_ERR_ARG, "NULL argument");
    if (b->parent) return fail(BPG_ERR_ARG, "bpg_set_id2obj: a view shares its parent's tables -- set them on the parent");
    if (model_idx < 0 || model_idx >= BPG_MAX_MODELS) return fail(BPG_ERR_ARG, "model_idx out of range");
    if (ngeom < 1 || ngeom > BPG_MAX_GEOMS) return fail(BPG_ERR_ARG, "ngeom must be in 1..64");
    int8_t tab[BPG_MAX_GEOMS];
    std::memset(tab, 0xFF, sizeof(tab));
    for (int i = 0; i < ngeom; i++) {
        if (id2obj[i] >= b->info.num_objs) return fail(BPG_ERR_ARG, "object id >= num_objs in id2obj");
        tab[i] = id2obj[i] < 0 ? (int8_t)-1 : id2obj[i];
    }
    DeviceGuard guard(b->device);
    HIP_TRY(hipDeviceSynchronize());      // synchronous by contract: no launch may still be reading the table
    HIP_TRY(hipMemcpy(b->d_id2obj + model_idx * BPG_MAX_GEOMS, tab, BPG_MAX_GEOMS, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    b->ngeom[model_idx] = ngeom;
    return BPG_OK;
}

// Row stride for the time-major (T, ld) streams of an n-env batch.  A wave keeps 5-10 rows of the
// same columns in flight, and rows a large power of two apart put all of them on the same HBM
// channels: measured at 2^20 .. 2^23 envs, 53-68 % of peak with ld = n against 67-75 % with every row
// shifted by an odd number of KB (profiles/r02_tune_row_stride.log; a 128-byte shift is worse than none, a
// 16 KB shift as bad as none, 2 KB as bad as none on most boxes).  Among the odd shifts (1, 3, 5, 9 KB) there is
// no winner: bench.py's row_stride_probe on a dozen box sessions (profiles/r03_row_stride_probes.log) has them
// within +-5 % of each other in either direction -- as far apart as two allocations of the SAME layout in one
// process.  Below 2^20 envs the shift does not pay (neutral up to 2^18, -2 % at 2^19 and 3 * 2^18).
// ld = n below 2^20; otherwise n rounded up to 256 elements, plus 256 when that leaves an even multiple.
int64_t bpg_row_stride(int64_t n)
{
    if (n < 1) return 0;
    if (n < ((int64_t)1 << 20)) return n;
    int64_t ld = (n + 255) / 256 * 256;
    if ((ld / 256) % 2 == 0) ld += 256;
    return ld;
}

int bpg_state_ptrs(bpg_batch *b, uint32_t **d_now, uint32_t **d_ever)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (d_now) *d_now = b->now;
    if (d_ever) *d_ever = b->ever;
    return BPG_OK;
}

int bpg_counter_ptr(bpg_batch *b, uint32_t **d_counters)
{
    if (!b || !d_counters) return fail(BPG_ERR_ARG, "NULL argument");
    *d_counters = b->d_reset_counter;
    return BPG_OK;
}

int bpg_reserve_return_scratch(bpg_batch *b)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (b->d_okbits) return BPG_OK;
    DeviceGuard guard(b->device);
    if (hipMalloc(&b->d_okbits, (size_t)((b->n + 1) & ~(int64_t)1) * sizeof(uint64_t)) != hipSuccess) {
        b->d_okbits = nullptr;
        return fail(BPG_ERR_HIP, "hipMalloc failed for the step-ok words");
    }
    return BPG_OK;
}

// the entry points of these names up to round 3 had other argument lists (include/bpgym.h): a stale caller gets an error
int bpg_step_record(void)
{
    return fail(BPG_ERR_UNSUPPORTED, "bpg_step_record(handle, ten arguments) is the round-3 entry point: use bpg_step_record_v2(handle, "
                                     "const bpg_step_args *, stream) of include/bpgym.h");
}

int bpg_rollout(void)
{
    return fail(BPG_ERR_UNSUPPORTED, "bpg_rollout is the round-3 entry point (a struct without BPG_ARGS_TAG): use bpg_rollout_v2 of "
                                     "include/bpgym.h");
}

// ---- curriculum ---------------------------------------------------------------------------------------------------
int bpg_get_curriculum(const bpg_batch *b, bpg_curriculum *out)
{
    if (!b || !out) return fail(BPG_ERR_ARG, "NULL argument");
    *out = root_of(b)->cur;
    return BPG_OK;
}

int bpg_set_curriculum(bpg_batch *b, const bpg_curriculum *in)
{
    if (!b || !in) return fail(BPG_ERR_ARG, "NULL argument");
    root_of(b)->cur = *in;
    return BPG_OK;
}

int bpg_get_difficulty(const bpg_batch *b) { return b ? root_of(b)->cur.difficulty : 0; }   // fetch_env.py:96-97

int bpg_increase_difficulty(bpg_batch *b, int *at_max)
{
    if (!b) return fail(BPG_ERR_ARG, "handle is NULL");
    if (at_max) *at_max = 0;
    const int v = b->info.variant;
    if (b->info.increase_difficulty_status == BPG_ERR_UNSUPPORTED)
        return fail(BPG_ERR_UNSUPPORTED, "increase_difficulty: NotImplementedError (fetch_env.py:93-94)");
    bpg_curriculum &c = root_of(b)->cur;
    if (!c.has_step)                                   // `self.obj_range += self.obj_range_step` raises before anything changes
        return fail(BPG_ERR_ATTRIBUTE, "increase_difficulty: the preset has no obj_range_step (AttributeError, fetch_env.py:413-420)");
    bool top = false;
    if (v == BPG_BLOCKS_TOUCH_CHOOSE || v == BPG_BLOCKS_TOUCH_CHOOSE_CURRICULUM) {     // fetch_env.py:419-432
        c.obj_range += c.obj_range_step;
        c.wrong_obj_range -= c.wrong_obj_range_step;
        if (c.obj_range > c.max_obj_range) {
            c.obj_range = c.max_obj_range;
            if (c.wrong_obj_range < 0) {
                c.wrong_obj_range = 0;
                top = true;
            }
        } else if (c.wrong_obj_range < 0) {
            c.wrong_obj_range = 0;
        }
        if (!top) c.difficulty += 1;
    } else {                                           // fetch_env.py:351-358 (BlocksTouch*), :623-630 (Variation)
        c.obj_range += c.obj_range_step;
        if (c.obj_range > c.max_obj_range) {
            c.obj_range = c.max_obj_range;
            top = true;
        } else {
            c.difficulty += 1;
        }
    }
    if (at_max) *at_max = top ? 1 : 0;
    return BPG_OK;
}

int bpg_placement_from_curriculum(const bpg_batch *b, int test, bpg_placement_args *out)
{
    if (!b || !out) return fail(BPG_ERR_ARG, "NULL argument");
    if (test && !b->info.set_test_supported)
        return fail(BPG_ERR_UNSUPPORTED, "set_test: NotImplementedError (fetch_env.py:100-101)");
    out->obj_range = root_of(b)->cur.obj_range;
    out->max_obj_range = root_of(b)->cur.max_obj_range;
    out->wrong_obj_range = root_of(b)->cur.wrong_obj_range;
    out->test = test ? 1 : 0;
    out->challenge = root_of(b)->cur.challenge;
    return BPG_OK;
}

}  // extern "C"
