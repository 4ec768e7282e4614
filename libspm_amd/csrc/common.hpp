// common.hpp -- internal types of libspm_hip.so (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/spm_hip.h"

namespace spm_hip
{

constexpr int kWave = 64; // CDNA wavefront

struct error_sink
{
    std::string msg;
};

extern thread_local std::string g_init_error;

} // namespace spm_hip

struct hits_block // device buffers + events of one scan result, recycled through the context
{
    spm_hit *d_hits = nullptr;
    unsigned long long *d_count = nullptr;
    uint64_t cap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool zeroed = false; // the counters were cleared when the block went back to the pool
    unsigned long long *h_c = nullptr; // pinned: where a deferred scan's counters land (allocated on first use, recycled)
    hipEvent_t ev_done = nullptr;
};

struct spm_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    std::string err;
    // scratch reused across scans
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    std::vector<hits_block> pool;
    std::vector<std::pair<void *, uint64_t>> jst_pool; // record buffers of journaled-sequence searches (pointer, capacity)
    // band table of the filter engine: empty between scans (see run_filter)
    ulonglong2 *d_band_tab = nullptr; // {key, value} per slot (filter.hpp: band_value)
    uint64_t band_slots = 0;
    bool band_dirty = false;
    uint32_t *d_table_poison = nullptr; // device flag: the table holds slots a scan left behind (filter.hpp: resolve_params)
    // pinned staging of needle-set uploads: two halves, so the host fills one while the other travels (patterns.hip)
    uint8_t *h_stage = nullptr;
    size_t stage_half = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    unsigned long long *h_counters = nullptr; // pinned: the per-scan counter read-back lands here (a pageable target costs
                                              // an extra staging hop on every scan)
};

struct spm_text
{
    spm_ctx *ctx = nullptr;
    uint8_t *d = nullptr;
    uint64_t n = 0;
    uint64_t alloc = 0; // bytes readable from d (>= n)
    uint32_t sigma = 4;
    bool owned = false;
    uint32_t *d_packed = nullptr; // optional 2-bit shadow (spm_hip_text_pack), zero-padded to whole 4096-symbol chunks
    uint64_t packed_words = 0;
};

struct spm_hits
{
    spm_ctx *ctx = nullptr;
    spm_hit *d_hits = nullptr;
    unsigned long long *d_count = nullptr; // [0] hits, [1] candidates, [2] overflow flags
    uint64_t cap = 0;
    uint64_t cand_cap = 0; // survivor list capacity of the filter run
    uint64_t band_cap = 0;
    uint64_t n = 0;
    bool counted = false;
    bool hook_final = false;          // the caller's after-launch work (scan_impl) ran on the final hit list ...
    unsigned long long fan_count = 0; // ... and counted this much into counter slot 12
    bool sorted_host = false;
    std::vector<spm_hit> host;
    spm_scan_stats stats{};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
    void *d_aux[2] = {nullptr, nullptr}; // segmented scans: tile table, segment offsets (freed with the hits)
    // deferred completion (SPM_SCAN_DEFER): the counters are on their way to h_c behind ev_done; what the scan was, in case
    // it has to be repeated
    bool pending = false;
    bool c_on_the_way = false;  // ... the copy of the counters into h_c has been enqueued (by the scan, or by the device-side
                                // fused copy, whose kernel writes them itself: one launch less in a C2 step)
    bool d_count_cleared = false; // the fused-copy kernel consumed the device counters and cleared them for the next scan
    unsigned long long *h_c = nullptr;
    hipEvent_t ev_done = nullptr;
    const spm_text *d_text = nullptr;
    const struct spm_patterns *d_patterns = nullptr;
    uint64_t d_begin = 0, d_end = 0;
    spm_scan_opts d_opts{};
};

struct pass_entry // key directory of one pass of the seed filter, in L2 (filter.hpp: resolve_kernel)
{
    const uint4 *ht; // {key, first entry, entries, -}, open addressing; an empty slot has .z == 0
    uint32_t ht_mask;
    uint32_t pad;
};

// scope guard for temporary device buffers: freed on every return path
struct dev_scratch
{
    std::vector<void *> owned;
    template <typename T>
    hipError_t alloc(T **p, size_t bytes)
    {
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
        if (e == hipSuccess)
            owned.push_back(*p);
        return e;
    }
    ~dev_scratch()
    {
        for (void *p : owned)
            hipFree(p);
    }
};

#define SPM_SET_ERR(ctx, ...)                                                                                          \
    do {                                                                                                               \
        char _b[512];                                                                                                  \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                                                                         \
        if (ctx)                                                                                                       \
            (ctx)->err = _b;                                                                                           \
        else                                                                                                           \
            spm_hip::g_init_error = _b;                                                                                \
    } while (0)

#define SPM_HIP_CHECK(ctx, call)                                                                                       \
    do {                                                                                                               \
        hipError_t _e = (call);                                                                                        \
        if (_e != hipSuccess) {                                                                                        \
            SPM_SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__);               \
            return SPM_E_HIP;                                                                                          \
        }                                                                                                              \
    } while (0)
