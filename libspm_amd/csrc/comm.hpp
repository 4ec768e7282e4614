// comm.hpp -- the multi-GPU exchange step behind the C ABI: a gatherv of hit records to one rank over RCCL
// (the north_star's "RCCL gatherv of hit records over xGMI").
//
// The path shards by text position and needs no data-path collective; the one exchange is this gatherv.  RCCL has no
// native gatherv:  (1) ncclAllGather of one uint64 hit count per rank,  (2) ncclGroupStart; every non-root rank
// ncclSend()s its n * record_bytes bytes to the root, the root ncclRecv()s each at its offset; ncclGroupEnd.
// librccl.so is opened with dlopen when the first communicator is made: libspm_hip.so has no link-time dependency on
// it, and a single-GPU user never loads it.  The Python driver (libspm_amd/dist.py) does the same exchange through
// torch.distributed for bench.py; this is the entry a C++ host (the drop-in boundary) uses.
#pragma once

#include <dlfcn.h>

#include "common.hpp"

namespace spm_hip
{

// the part of rccl.h this file uses (declared here so that building does not need the RCCL headers either)
struct rccl_unique_id
{
    char internal[128];
};
typedef void *rccl_comm_t;
enum
{
    kNcclSuccess = 0,
    kNcclUint8 = 1,
    kNcclUint64 = 5
};

struct rccl_api
{
    void *lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

inline rccl_api *rccl()
{
    static rccl_api api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib)
                break;
        }
        if (api.lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
            api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
            api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
            api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Send || !api.Recv ||
                !api.GroupStart || !api.GroupEnd) {
                dlclose(api.lib);
                api.lib = nullptr;
            }
        }
    }
    return api.lib ? &api : nullptr;
}

} // namespace spm_hip

struct spm_comm
{
    spm_ctx *ctx = nullptr;
    spm_hip::rccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    unsigned long long *d_counts = nullptr; // [world + 1]: slot world = this rank's count (send buffer)
    unsigned long long *h_counts = nullptr; // pinned
    void *d_recv = nullptr;                 // root: gathered records (grown on demand)
    uint64_t recv_bytes = 0;
};

#define SPM_RCCL_CHECK(ctx, call)                                                                                      \
    do {                                                                                                               \
        int _r = (call);                                                                                               \
        if (_r != spm_hip::kNcclSuccess) {                                                                             \
            spm_hip::rccl_api *_a = spm_hip::rccl();                                                                   \
            SPM_SET_ERR(ctx, "%s failed: %s", #call, _a && _a->GetErrorString ? _a->GetErrorString(_r) : "rccl error"); \
            return SPM_E_HIP;                                                                                          \
        }                                                                                                              \
    } while (0)

// Offsets of the gatherv: rank r's records land at byte offsets[r] of the root's buffer; offsets[world] = total bytes.
extern "C" int spm_hip_gatherv_plan(const uint64_t *counts, uint32_t world, uint32_t record_bytes, uint64_t *offsets)
{
    if (!counts || !offsets || world == 0 || record_bytes == 0)
        return SPM_E_INVALID;
    uint64_t at = 0;
    for (uint32_t r = 0; r < world; ++r) {
        offsets[r] = at;
        if (counts[r] > (~0ull - at) / record_bytes)
            return SPM_E_OVERFLOW;
        at += counts[r] * record_bytes;
    }
    offsets[world] = at;
    return SPM_OK;
}

extern "C" int spm_hip_comm_unique_id(void *id128)
{
    spm_hip::rccl_api *A = spm_hip::rccl();
    if (!id128 || !A)
        return SPM_E_UNSUPPORTED;
    return A->GetUniqueId(static_cast<spm_hip::rccl_unique_id *>(id128)) == spm_hip::kNcclSuccess ? SPM_OK : SPM_E_HIP;
}

extern "C" int spm_hip_comm_init(spm_ctx *ctx, const void *unique_id128, int rank, int world, spm_comm **out)
{
    if (!ctx || !unique_id128 || !out || world < 1 || rank < 0 || rank >= world) {
        SPM_SET_ERR(ctx, "spm_hip_comm_init: invalid argument");
        return SPM_E_INVALID;
    }
    spm_hip::rccl_api *A = spm_hip::rccl();
    if (!A) {
        SPM_SET_ERR(ctx, "spm_hip_comm_init: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
        return SPM_E_UNSUPPORTED;
    }
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<spm_comm> C(new spm_comm);
    C->ctx = ctx;
    C->rank = rank;
    C->world = world;
    spm_hip::rccl_unique_id id;
    memcpy(&id, unique_id128, sizeof(id));
    SPM_RCCL_CHECK(ctx, A->CommInitRank(&C->comm, world, id, rank));
    SPM_HIP_CHECK(ctx, hipMalloc(&C->d_counts, (size_t)(world + 1) * sizeof(unsigned long long)));
    SPM_HIP_CHECK(ctx, hipHostMalloc(&C->h_counts, (size_t)(world + 1) * sizeof(unsigned long long), hipHostMallocDefault));
    *out = C.release();
    return SPM_OK;
}

extern "C" void spm_hip_comm_destroy(spm_comm *c)
{
    if (!c)
        return;
    spm_hip::rccl_api *A = spm_hip::rccl();
    if (c->ctx)
        hipStreamSynchronize(c->ctx->stream);
    if (A && c->comm)
        A->CommDestroy(c->comm);
    hipFree(c->d_counts);
    hipFree(c->d_recv);
    if (c->h_counts)
        hipHostFree(c->h_counts);
    delete c;
}

// Gather `n_local` records of `record_bytes` bytes (device memory) from every rank to `root`, rank order = shard order.
// On the root: *device_records = the gathered records (owned by the communicator, valid until the next call),
// counts[0..world) = records per rank (may be NULL), *n_total = their sum.  Elsewhere *device_records = NULL, *n_total = 0.
// Runs on the context's stream; returns when the records are there (one host synchronisation: the root has to size its
// receives).
static int gatherv_device(spm_comm *c, const void *d_local, uint64_t n_local, uint32_t record_bytes, int root,
                          const void **device_records, uint64_t *n_total, uint64_t *counts)
{
    spm_ctx *ctx = c->ctx;
    spm_hip::rccl_api *A = spm_hip::rccl();
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int W = c->world;
    c->h_counts[W] = n_local;
    SPM_HIP_CHECK(ctx, hipMemcpyAsync(c->d_counts + W, c->h_counts + W, sizeof(unsigned long long), hipMemcpyHostToDevice,
                                      ctx->stream));
    SPM_RCCL_CHECK(ctx, A->AllGather(c->d_counts + W, c->d_counts, 1, spm_hip::kNcclUint64, c->comm, ctx->stream));
    *device_records = nullptr;
    *n_total = 0;
    if (c->rank != root) {
        if (n_local) {
            SPM_RCCL_CHECK(ctx, A->GroupStart());
            SPM_RCCL_CHECK(ctx, A->Send(d_local, n_local * record_bytes, spm_hip::kNcclUint8, root, c->comm, ctx->stream));
            SPM_RCCL_CHECK(ctx, A->GroupEnd());
        }
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return SPM_OK;
    }
    SPM_HIP_CHECK(ctx, hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                      ctx->stream));
    SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> cnt(c->h_counts, c->h_counts + W), off((size_t)W + 1);
    int rc = spm_hip_gatherv_plan(cnt.data(), (uint32_t)W, record_bytes, off.data());
    if (rc != SPM_OK)
        return rc;
    if (off[W] > c->recv_bytes) {
        hipFree(c->d_recv);
        c->d_recv = nullptr;
        c->recv_bytes = 0;
        SPM_HIP_CHECK(ctx, hipMalloc(&c->d_recv, off[W] + off[W] / 4 + 4096));
        c->recv_bytes = off[W] + off[W] / 4 + 4096;
    }
    if (cnt[root])
        SPM_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t *)c->d_recv + off[root], d_local, cnt[root] * record_bytes,
                                          hipMemcpyDeviceToDevice, ctx->stream));
    bool any = false;
    for (int r = 0; r < W; ++r)
        any = any || (r != root && cnt[r]);
    if (any) {
        SPM_RCCL_CHECK(ctx, A->GroupStart());
        for (int r = 0; r < W; ++r)
            if (r != root && cnt[r])
                SPM_RCCL_CHECK(ctx, A->Recv((uint8_t *)c->d_recv + off[r], cnt[r] * record_bytes, spm_hip::kNcclUint8, r,
                                            c->comm, ctx->stream));
        SPM_RCCL_CHECK(ctx, A->GroupEnd());
    }
    SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *device_records = c->d_recv;
    *n_total = off[W] / record_bytes;
    if (counts)
        for (int r = 0; r < W; ++r)
            counts[r] = cnt[r];
    return SPM_OK;
}

extern "C" int spm_hip_gatherv_hits(spm_comm *c, spm_hits *local, int root, const void **device_records,
                                    uint64_t *n_total, uint64_t *counts)
{
    if (!c || !local || !device_records || !n_total || root < 0 || root >= c->world || local->ctx != c->ctx) {
        if (c)
            SPM_SET_ERR(c->ctx, "spm_hip_gatherv_hits: invalid argument (the hits must come from the communicator's context)");
        return SPM_E_INVALID;
    }
    const void *d = nullptr;
    uint64_t n = 0;
    int rc = spm_hip_hits_device(local, &d, &n);
    if (rc != SPM_OK)
        return rc;
    return gatherv_device(c, d, n, (uint32_t)sizeof(spm_hit), root, device_records, n_total, counts);
}

extern "C" int spm_hip_gatherv_jst_hits(spm_comm *c, spm_jst_hits *local, int root, const void **device_records,
                                        uint64_t *n_total, uint64_t *counts)
{
    if (!c || !local || !device_records || !n_total || root < 0 || root >= c->world) {
        if (c)
            SPM_SET_ERR(c->ctx, "spm_hip_gatherv_jst_hits: invalid argument");
        return SPM_E_INVALID;
    }
    const void *d = nullptr;
    uint64_t n = 0;
    int rc = spm_hip_jst_hits_device(local, &d, &n);
    if (rc != SPM_OK)
        return rc;
    return gatherv_device(c, d, n, (uint32_t)sizeof(spm_jst_hit), root, device_records, n_total, counts);
}
