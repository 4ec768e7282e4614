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

#include <mutex>

#include "comm_protocol.hpp"
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

inline std::string &rccl_load_error()
{
    static std::string e;
    return e;
}

inline rccl_api *rccl()
{
    static rccl_api api;
    static std::once_flag once; // (two threads making their first communicators at the same time load the table once)
    std::call_once(once, []() {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib)
                break;
            const char *e = dlerror(); // (read once: the call clears it)
            if (e)
                rccl_load_error() = e;
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
                rccl_load_error() = "symbols missing";
            }
        }
    });
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
    return spm_hip::gatherv_plan(counts, world, record_bytes, offsets);
}

extern "C" int spm_hip_comm_selftest(int world, int root, int scenario, int victim, uint32_t record_bytes, uint64_t seed,
                                     int *detail)
{
    return spm_hip::comm_selftest(world, root, scenario, victim, record_bytes, seed, detail);
}

extern "C" int spm_hip_comm_unique_id(void *id128)
{
    spm_hip::rccl_api *A = spm_hip::rccl();
    if (!id128 || !A)
        return SPM_E_UNSUPPORTED;
    return A->GetUniqueId(static_cast<spm_hip::rccl_unique_id *>(id128)) == spm_hip::kNcclSuccess ? SPM_OK : SPM_E_HIP;
}

extern "C" void spm_hip_comm_destroy(spm_comm *c);

extern "C" int spm_hip_comm_init(spm_ctx *ctx, const void *unique_id128, int rank, int world, spm_comm **out)
{
    if (!ctx || !unique_id128 || !out || world < 1 || rank < 0 || rank >= world) {
        SPM_SET_ERR(ctx, "spm_hip_comm_init: invalid argument");
        return SPM_E_INVALID;
    }
    spm_hip::rccl_api *A = spm_hip::rccl();
    if (!A) {
        SPM_SET_ERR(ctx, "spm_hip_comm_init: librccl.so could not be loaded (%s)", spm_hip::rccl_load_error().c_str());
        return SPM_E_UNSUPPORTED;
    }
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // (the error paths below release the RCCL communicator and the buffers too)
    std::unique_ptr<spm_comm, void (*)(spm_comm *)> C(new spm_comm, spm_hip_comm_destroy);
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
    if (c->ctx) {
        hipSetDevice(c->ctx->device);
        hipStreamSynchronize(c->ctx->stream);
    }
    if (A && c->comm)
        A->CommDestroy(c->comm);
    hipFree(c->d_counts);
    hipFree(c->d_recv);
    if (c->h_counts)
        hipHostFree(c->h_counts);
    delete c;
}

namespace
{
// the transport of comm_protocol.hpp over RCCL + HIP, on the context's stream
struct rccl_transport : spm_hip::gatherv_transport
{
    spm_comm *c;
    spm_hip::rccl_api *A;
    explicit rccl_transport(spm_comm *comm) : c(comm), A(spm_hip::rccl()) {}
    int exchange_words(uint64_t mine, uint64_t *all) override
    {
        spm_ctx *ctx = c->ctx;
        const int W = c->world;
        c->h_counts[W] = mine;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(c->d_counts + W, c->h_counts + W, sizeof(unsigned long long), hipMemcpyHostToDevice,
                                          ctx->stream));
        SPM_RCCL_CHECK(ctx, A->AllGather(c->d_counts + W, c->d_counts, 1, spm_hip::kNcclUint64, c->comm, ctx->stream));
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                          ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < W; ++r)
            all[r] = c->h_counts[r];
        return SPM_OK;
    }
    int reserve(uint64_t bytes, void **buffer) override
    {
        spm_ctx *ctx = c->ctx;
        if (bytes > c->recv_bytes) {
            hipFree(c->d_recv);
            c->d_recv = nullptr;
            c->recv_bytes = 0;
            const uint64_t want = bytes + bytes / 4 + 4096;
            if (hipMalloc(&c->d_recv, want) != hipSuccess) {
                c->d_recv = nullptr;
                SPM_SET_ERR(ctx, "gatherv: the root cannot allocate %llu bytes for the gathered records", (unsigned long long)want);
                return SPM_E_NOMEM;
            }
            c->recv_bytes = want;
        }
        *buffer = c->d_recv;
        return SPM_OK;
    }
    int copy_own(void *dst, const void *src, uint64_t bytes) override
    {
        SPM_HIP_CHECK(c->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
        return SPM_OK;
    }
    int group_begin() override
    {
        SPM_RCCL_CHECK(c->ctx, A->GroupStart());
        return SPM_OK;
    }
    int send(const void *src, uint64_t bytes, int peer) override
    {
        SPM_RCCL_CHECK(c->ctx, A->Send(src, bytes, spm_hip::kNcclUint8, peer, c->comm, c->ctx->stream));
        return SPM_OK;
    }
    int recv(void *dst, uint64_t bytes, int peer) override
    {
        SPM_RCCL_CHECK(c->ctx, A->Recv(dst, bytes, spm_hip::kNcclUint8, peer, c->comm, c->ctx->stream));
        return SPM_OK;
    }
    int group_end() override
    {
        SPM_RCCL_CHECK(c->ctx, A->GroupEnd());
        return SPM_OK;
    }
    int finish() override
    {
        SPM_HIP_CHECK(c->ctx, hipStreamSynchronize(c->ctx->stream));
        return SPM_OK;
    }
};
} // namespace

// Gather `n_local` records of `record_bytes` bytes (device memory) from every rank to `root`, rank order = shard order.
// On the root: *device_records = the gathered records (owned by the communicator, valid until the next call),
// counts[0..world) = records per rank (may be NULL), *n_total = their sum.  Elsewhere *device_records = NULL, *n_total = 0.
// local_error: this rank's result is unusable -- it still takes part, and every rank returns an error (comm_protocol.hpp).
// Runs on the context's stream; returns when the records are there.
static int gatherv_device(spm_comm *c, int local_error, const void *d_local, uint64_t n_local, uint32_t record_bytes, int root,
                          const void **device_records, uint64_t *n_total, uint64_t *counts)
{
    SPM_HIP_CHECK(c->ctx, hipSetDevice(c->ctx->device));
    rccl_transport T(c);
    const int rc = spm_hip::gatherv_protocol(T, c->rank, c->world, root, local_error, d_local, n_local, record_bytes,
                                             device_records, n_total, counts);
    if (rc == SPM_E_PEER)
        SPM_SET_ERR(c->ctx, "gatherv: another rank reported an error; nothing was exchanged");
    return rc;
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
    const int rc = spm_hip_hits_device(local, &d, &n); // (e.g. SPM_E_OVERFLOW: more hits than max_hits -- told to every rank)
    return gatherv_device(c, rc, d, rc == SPM_OK ? n : 0, (uint32_t)sizeof(spm_hit), root, device_records, n_total, counts);
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
    const int rc = spm_hip_jst_hits_device(local, &d, &n);
    return gatherv_device(c, rc, d, rc == SPM_OK ? n : 0, (uint32_t)sizeof(spm_jst_hit), root, device_records, n_total, counts);
}
