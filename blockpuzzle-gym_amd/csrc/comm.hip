// comm.hip -- the one exchange of a multi-GPU job: RCCL collectives issued by the library on the caller's
// stream.  Mirrors what the reference does across its MPI ranks (train.py:21-26 mpi_average, :113-119 mpi_fork) and per
// rollout (rollout.py:163-167 `success_rate = np.mean(successful)`): an all-gather of a per-env statistic of the rollout,
// or an all-reduce of the [successes, envs] rows.  librccl.so is loaded on first use and never linked: a single-GPU host
// does not need it.  xGMI is point-to-point, so a ring collective is bound per link: the byte form of the episode return
// (bpg_rollout_args.d_episode_fail_count) moves a quarter of the float32 form's bytes.
#include "state.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only; every function is resolved with dlsym

struct bpg_comm {
    ncclComm_t comm;
    int rank, world, device;
    bool owned;
};

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

// loaded once per process; a process that already has an RCCL (PyTorch's) gets that very library back from dlopen
Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        const char *e = dlerror();
        r.why = std::string("librccl.so could not be loaded: ") + (e ? e : "?");
        return &r;
    }
    auto sym = [&](const char *nm) {
        void *p = dlsym(r.lib, nm);
        if (!p && r.why.empty()) r.why = std::string("librccl.so has no symbol ") + nm;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return &r;
}

int need_rccl(Rccl *&out)
{
    out = rccl();
    if (!out->why.empty()) return fail(BPG_ERR_UNSUPPORTED, out->why);
    return BPG_OK;
}

int nccl_fail(Rccl *r, const char *what, ncclResult_t e)
{
    return fail(BPG_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

}  // namespace

extern "C" {

int bpg_comm_unique_id(void *id_bytes)
{
    static_assert(BPG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "BPG_COMM_ID_BYTES");
    if (!id_bytes) return fail(BPG_ERR_ARG, "NULL argument");
    Rccl *r;
    if (int rc = need_rccl(r)) return rc;
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
    std::memcpy(id_bytes, id.internal, NCCL_UNIQUE_ID_BYTES);
    return BPG_OK;
}

int bpg_comm_create(bpg_comm **out, int rank, int world_size, const void *id_bytes, int device_id)
{
    if (!out || !id_bytes) return fail(BPG_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(BPG_ERR_ARG, "rank / world_size out of range");
    Rccl *r;
    if (int rc = need_rccl(r)) return rc;
    DeviceGuard guard(device_id);
    if (guard.err != hipSuccess) return fail(BPG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    ncclUniqueId id;
    std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->CommInitRank(&comm, world_size, id, rank);        // collective over the ranks: blocks until all arrive
    if (e != ncclSuccess) return nccl_fail(r, "ncclCommInitRank", e);
    bpg_comm *c = new (std::nothrow) bpg_comm{comm, rank, world_size, device_id, true};
    if (!c) {
        (void)r->CommDestroy(comm);
        return fail(BPG_ERR_ARG, "out of host memory");
    }
    *out = c;
    return BPG_OK;
}

int bpg_comm_adopt(bpg_comm **out, void *nccl_comm, int rank, int world_size, int device_id)
{
    if (!out || !nccl_comm) return fail(BPG_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(BPG_ERR_ARG, "rank / world_size out of range");
    Rccl *r;
    if (int rc = need_rccl(r)) return rc;
    bpg_comm *c = new (std::nothrow) bpg_comm{static_cast<ncclComm_t>(nccl_comm), rank, world_size, device_id, false};
    if (!c) return fail(BPG_ERR_ARG, "out of host memory");
    *out = c;
    return BPG_OK;
}

int bpg_comm_destroy(bpg_comm *c)
{
    if (!c) return BPG_OK;
    int rc = BPG_OK;
    if (c->owned) {
        Rccl *r = rccl();
        DeviceGuard guard(c->device);
        const ncclResult_t e = r->CommDestroy(c->comm);
        if (e != ncclSuccess) rc = nccl_fail(r, "ncclCommDestroy", e);
    }
    delete c;
    return rc;
}

int bpg_comm_rank(const bpg_comm *c) { return c ? c->rank : -1; }
int bpg_comm_world_size(const bpg_comm *c) { return c ? c->world : 0; }

static int allgather(bpg_comm *c, const void *d_local, void *d_out, int64_t n_local, ncclDataType_t dt, void *stream)
{
    if (!c || !d_local || !d_out) return fail(BPG_ERR_ARG, "NULL argument");
    if (n_local < 0) return fail(BPG_ERR_ARG, "n_local < 0");
    if (n_local == 0) return BPG_OK;
    Rccl *r = rccl();
    DeviceGuard guard(c->device);
    const ncclResult_t e = r->AllGather(d_local, d_out, (size_t)n_local, dt, c->comm, (hipStream_t)stream);
    if (e != ncclSuccess) return nccl_fail(r, "ncclAllGather", e);
    return BPG_OK;
}

int bpg_allgather_u8(bpg_comm *c, const uint8_t *d_local, uint8_t *d_out, int64_t n_local, void *stream)
{
    return allgather(c, d_local, d_out, n_local, ncclUint8, stream);
}

int bpg_allgather_f32(bpg_comm *c, const float *d_local, float *d_out, int64_t n_local, void *stream)
{
    return allgather(c, d_local, d_out, n_local, ncclFloat32, stream);
}

int bpg_allreduce_sum_u64(bpg_comm *c, unsigned long long *d_inout, int64_t count, void *stream)
{
    if (!c || !d_inout) return fail(BPG_ERR_ARG, "NULL argument");
    if (count < 0) return fail(BPG_ERR_ARG, "count < 0");
    if (count == 0) return BPG_OK;
    Rccl *r = rccl();
    DeviceGuard guard(c->device);
    const ncclResult_t e = r->AllReduce(d_inout, d_inout, (size_t)count, ncclUint64, ncclSum, c->comm, (hipStream_t)stream);
    if (e != ncclSuccess) return nccl_fail(r, "ncclAllReduce", e);
    return BPG_OK;
}

}  // extern "C"
