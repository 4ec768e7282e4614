// hits.hip -- results of a scan behind the C ABI: views, copies, statistics.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "hd.hpp"

// ----------------------------------------------------------------------------------------------------
// hits
// ----------------------------------------------------------------------------------------------------
static int hits_count(spm_hits *h)
{
    spm_ctx *ctx = h->ctx;
    if (h->pending) {
        const int rc = spm_complete_deferred(h);
        if (rc != SPM_OK)
            return rc;
    }
    if (!h->counted) {
        unsigned long long *c = ctx->h_counters;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(c, h->d_count, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        h->n = c[0];
        h->counted = true;
    }
    if (h->n > h->cap) {
        SPM_SET_ERR(ctx, "scan produced %llu hits but the buffer holds %llu; raise spm_scan_opts.max_hits",
                    (unsigned long long)h->n, (unsigned long long)h->cap);
        return SPM_E_OVERFLOW;
    }
    return SPM_OK;
}

extern "C" int spm_hip_hits_view(spm_hits *h, const spm_hit **records, uint64_t *n)
{
    if (!h || !records || !n)
        return SPM_E_INVALID;
    int rc = hits_count(h);
    if (rc != SPM_OK)
        return rc;
    if (!h->sorted_host) {
        h->host.resize(h->n);
        if (h->n) {
            SPM_HIP_CHECK(h->ctx, hipMemcpyAsync(h->host.data(), h->d_hits, h->n * sizeof(spm_hit),
                                                 hipMemcpyDeviceToHost, h->ctx->stream));
            SPM_HIP_CHECK(h->ctx, hipStreamSynchronize(h->ctx->stream));
        }
        std::sort(h->host.begin(), h->host.end(), [](const spm_hit &a, const spm_hit &b) {
            // positions compare as signed: a restored Shift-Or state can complete an occurrence that began before
            // this chunk, whose begin position is "negative" (wrapped) relative to the chunk
            return a.pattern != b.pattern ? a.pattern < b.pattern : (int64_t)a.pos < (int64_t)b.pos;
        });
        h->sorted_host = true;
    }
    *records = h->host.data();
    *n = h->n;
    return SPM_OK;
}

extern "C" int spm_hip_hits_device(spm_hits *h, const void **device_records, uint64_t *n)
{
    if (!h || !device_records || !n)
        return SPM_E_INVALID;
    int rc = hits_count(h);
    if (rc != SPM_OK)
        return rc;
    *device_records = h->d_hits;
    *n = h->n;
    return SPM_OK;
}

extern "C" int spm_hip_hits_copy_device(spm_hits *h, void *device_dst, uint64_t cap, uint64_t *n)
{
    if (!h || !n || (cap && !device_dst))
        return SPM_E_INVALID;
    int rc = hits_count(h);
    if (rc != SPM_OK)
        return rc;
    *n = h->n;
    const uint64_t c = std::min(h->n, cap);
    if (c)
        SPM_HIP_CHECK(h->ctx, hipMemcpyAsync(device_dst, h->d_hits, c * sizeof(spm_hit), hipMemcpyDeviceToDevice,
                                             h->ctx->stream));
    return SPM_OK;
}

// [count | records] in one launch: lane 0 writes the 16-byte header {n, 0}, the grid copies the records
__global__ __launch_bounds__(256) void hits_fused_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src,
                                                                unsigned long long n, unsigned long long n_copy)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        dst[0] = make_uint4((uint32_t)n, (uint32_t)(n >> 32), 0u, 0u);
    for (uint64_t r = i; r < n_copy; r += (uint64_t)gridDim.x * blockDim.x)
        dst[1 + r] = src[r];
}

extern "C" int spm_hip_hits_copy_fused(spm_hits *h, void *device_dst, uint64_t cap, uint64_t *n)
{
    if (!h || !n || !device_dst)
        return SPM_E_INVALID;
    int rc = hits_count(h);
    if (rc != SPM_OK)
        return rc;
    *n = h->n;
    if ((uintptr_t)device_dst & 15) {
        SPM_SET_ERR(h->ctx, "spm_hip_hits_copy_fused: the destination must be 16-byte aligned");
        return SPM_E_INVALID;
    }
    // one kernel: the count travels as a kernel argument (no host buffer that would have to outlive the call), the records
    // behind it
    const uint64_t c = std::min(h->n, cap);
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((c + 255) / 256, (uint64_t)h->ctx->n_cu * 4));
    hipLaunchKernelGGL(hits_fused_copy_kernel, dim3(grid), dim3(256), 0, h->ctx->stream, static_cast<uint4 *>(device_dst),
                       reinterpret_cast<const uint4 *>(h->d_hits), (unsigned long long)h->n, (unsigned long long)c);
    SPM_HIP_CHECK(h->ctx, hipGetLastError());
    return SPM_OK;
}

// [count | status | records] without the host: the counters are read on the device.  status != 0: a list overflowed or
// spans gave up -- the host has to complete the scan (scan.hip: spm_complete_deferred).  The kernel also delivers the
// counters to the result's pinned block (`host_c`, device-visible host memory) and, once every workgroup has read them
// (a ticket in the spare slot 15), clears them for the scan that will reuse the block: a deferred C2 step is three
// launches -- streaming, resolve, this -- instead of five.
__global__ __launch_bounds__(256) void hits_fused_copy_device_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src,
                                                                       unsigned long long *__restrict__ counters,
                                                                       unsigned long long *__restrict__ host_c,
                                                                       unsigned long long cap, unsigned long long hit_cap,
                                                                       unsigned long long cand_cap)
{
    const unsigned long long n = counters[0];
    const unsigned long long status = (counters[2] != 0 || counters[6] != 0 || counters[1] > cand_cap || n > hit_cap) ? 1ull : 0ull;
    const unsigned long long n_copy = n < cap ? (n < hit_cap ? n : hit_cap) : cap;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        dst[0] = make_uint4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)status, 0u);
    if (blockIdx.x == 0 && threadIdx.x < 13 && host_c)
        host_c[threadIdx.x] = counters[threadIdx.x];
    for (uint64_t r = i; r < n_copy; r += (uint64_t)gridDim.x * blockDim.x)
        dst[1 + r] = src[r];
    if (host_c) {
        __syncthreads(); // (every thread of this workgroup has read what it needs of the counters)
        __shared__ unsigned int last;
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(&counters[15], 1ull) == (unsigned long long)gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last && threadIdx.x < 16)
            counters[threadIdx.x] = 0;
    }
}

extern "C" int spm_hip_hits_copy_fused_device(spm_hits *h, void *device_dst, uint64_t cap)
{
    if (!h || !device_dst || ((uintptr_t)device_dst & 15))
        return SPM_E_INVALID;
    if (h->pending && h->c_on_the_way) {
        // a second device-side copy of the same deferred result: the counters have gone to the host (and the device's
        // copy may be cleared): complete the scan there first
        const int rc = hits_count(h);
        if (rc != SPM_OK && rc != SPM_E_OVERFLOW)
            return rc;
    }
    if (!h->pending && h->counted) {
        // the host has completed this scan (its fallbacks included; the device counters may still show what it handled):
        // the count it knows, status 0
        const uint64_t c = std::min(h->n, std::min<uint64_t>(cap, h->cap));
        const unsigned g1 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((c + 255) / 256, (uint64_t)h->ctx->n_cu * 4));
        hipLaunchKernelGGL(hits_fused_copy_kernel, dim3(g1), dim3(256), 0, h->ctx->stream, static_cast<uint4 *>(device_dst),
                           reinterpret_cast<const uint4 *>(h->d_hits), (unsigned long long)h->n, (unsigned long long)c);
        SPM_HIP_CHECK(h->ctx, hipGetLastError());
        return SPM_OK;
    }
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((std::min<uint64_t>(cap, h->cap) + 255) / 256, (uint64_t)h->ctx->n_cu));
    unsigned long long *host_c = h->pending ? h->h_c : nullptr; // (a deferred scan: its counters travel with this launch)
    hipLaunchKernelGGL(hits_fused_copy_device_kernel, dim3(grid), dim3(256), 0, h->ctx->stream, static_cast<uint4 *>(device_dst),
                       reinterpret_cast<const uint4 *>(h->d_hits), h->d_count, host_c, (unsigned long long)cap,
                       (unsigned long long)h->cap, (unsigned long long)(h->cand_cap ? h->cand_cap : ~0ull));
    SPM_HIP_CHECK(h->ctx, hipGetLastError());
    if (host_c) {
        SPM_HIP_CHECK(h->ctx, hipEventRecord(h->ev_done, h->ctx->stream));
        h->c_on_the_way = true;
        h->d_count_cleared = true;
    }
    return SPM_OK;
}

extern "C" int spm_hip_hits_stats(const spm_hits *hc, spm_scan_stats *out)
{
    if (!hc || !out)
        return SPM_E_INVALID;
    spm_hits *h = const_cast<spm_hits *>(hc);
    int rc = hits_count(h);
    if (rc != SPM_OK && rc != SPM_E_OVERFLOW)
        return rc;
    if (h->timed) {
        SPM_HIP_CHECK(h->ctx, hipEventSynchronize(h->ev[3]));
        hipEventElapsedTime(&h->stats.ms_total, h->ev[0], h->ev[3]);
        hipEventElapsedTime(&h->stats.ms_main, h->ev[1], h->ev[2]);
        hipEventElapsedTime(&h->stats.ms_verify, h->ev[2], h->ev[3]);
    }
    h->stats.n_hits = h->n;
    *out = h->stats;
    return SPM_OK;
}

extern "C" uint64_t spm_hip_hits_checksum(spm_hits *h)
{
    const spm_hit *r = nullptr;
    uint64_t n = 0;
    if (spm_hip_hits_view(h, &r, &n) != SPM_OK)
        return 0;
    uint64_t s = 0;
    for (uint64_t i = 0; i < n; ++i)
        s += mix64(r[i].pos ^ ((uint64_t)r[i].pattern << 40) ^ ((uint64_t)(uint32_t)r[i].score << 58));
    return s;
}

extern "C" void spm_hip_hits_destroy(spm_hits *h)
{
    if (!h)
        return;
    if (h->pending)
        (void)spm_complete_deferred(h); // (the needle set's hints; a scan that needed attention is simply dropped here)
    if (h->d_aux[0] || h->d_aux[1]) {
        if (h->ctx)
            hipStreamSynchronize(h->ctx->stream);
        hipFree(h->d_aux[0]);
        hipFree(h->d_aux[1]);
    }
    if (h->ctx && h->d_hits && h->d_count && h->ev[3] && h->ctx->pool.size() < 8) {
        hits_block b;
        b.d_hits = h->d_hits;
        b.d_count = h->d_count;
        b.cap = h->cap;
        for (int e = 0; e < 4; ++e)
            b.ev[e] = h->ev[e];
        b.h_c = h->h_c;
        b.ev_done = h->ev_done;
        // the next scan's counters: cleared now, behind this scan's last read of them (stream order), not in front of that scan
        // (a deferred scan whose counters went out through the device-side fused copy has cleared them there)
        b.zeroed = h->d_count_cleared || hipMemsetAsync(b.d_count, 0, 16 * sizeof(unsigned long long), h->ctx->stream) == hipSuccess;
        h->ctx->pool.push_back(b); // stream order makes reuse by the next scan safe
    } else {
        if (h->ctx)
            hipStreamSynchronize(h->ctx->stream);
        hipFree(h->d_hits);
        hipFree(h->d_count);
        for (int i = 0; i < 6; ++i)
            if (h->ev[i])
                hipEventDestroy(h->ev[i]);
        if (h->h_c)
            hipHostFree(h->h_c);
        if (h->ev_done)
            hipEventDestroy(h->ev_done);
    }
    delete h;
}


void spm_warm_hits_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)hits_fused_copy_kernel);
}
