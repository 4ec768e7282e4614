// text.hip -- context and haystacks behind the C ABI; the synthetic inputs of the benchmark configs.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "synth_kernels.hpp"
#include "text_kernels.hpp"

namespace spm_hip
{
thread_local std::string g_init_error;
}

// ----------------------------------------------------------------------------------------------------
// context / text
// ----------------------------------------------------------------------------------------------------
extern "C" int spm_hip_init(int device, void *stream, spm_ctx **out)
{
    if (!out) {
        SPM_SET_ERR((spm_ctx *)nullptr, "spm_hip_init: out == NULL");
        return SPM_E_INVALID;
    }
    spm_ctx *none = nullptr;
    int n_dev = 0;
    SPM_HIP_CHECK(none, hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) {
        SPM_SET_ERR(none, "spm_hip_init: device %d not available (%d HIP devices)", device, n_dev);
        return SPM_E_INVALID;
    }
    SPM_HIP_CHECK(none, hipSetDevice(device));
    hipDeviceProp_t prop;
    SPM_HIP_CHECK(none, hipGetDeviceProperties(&prop, device));
    std::unique_ptr<spm_ctx> ctx(new spm_ctx);
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        SPM_HIP_CHECK(none, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    SPM_HIP_CHECK(none, hipHostMalloc(&ctx->h_counters, 16 * sizeof(unsigned long long), hipHostMallocDefault));
    // needle sets go up through pinned memory in one stream of chunks (a pageable hipMemcpy per table costs a staging hop
    // and a synchronisation each: 12 of the 16 ms of creating 1024 needles)
    ctx->stage_half = (size_t)8 << 20;
    SPM_HIP_CHECK(none, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), 2 * ctx->stage_half, hipHostMallocDefault));
    for (int e = 0; e < 2; ++e)
        SPM_HIP_CHECK(none, hipEventCreateWithFlags(&ctx->stage_ev[e], hipEventDisableTiming));
    // the first host-to-device copy of a process sets up the copy engine's queue (7 of the 8 ms of uploading the 2 MB of a
    // 1024-needle set): done here, with the first device-to-host one
    if (hipMalloc(&ctx->d_scratch, (size_t)128 << 20) == hipSuccess) { // (the first scan of 1 GiB against 1 024 needles asks for 91 MB)
        ctx->scratch_bytes = (size_t)128 << 20;
        memset(ctx->h_stage, 0, ctx->stage_half); // (a copy of the size the uploads use: small ones take another path)
        (void)hipMemcpyAsync(ctx->d_scratch, ctx->h_stage, ctx->stage_half, hipMemcpyHostToDevice, ctx->stream);
        (void)hipEventRecord(ctx->stage_ev[0], ctx->stream);
        (void)hipMemcpyAsync(ctx->h_counters, ctx->d_scratch, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    } else {
        ctx->d_scratch = nullptr;
    }
    SPM_HIP_CHECK(none, hipMalloc(reinterpret_cast<void **>(&ctx->d_table_poison), 16));
    SPM_HIP_CHECK(none, hipMemsetAsync(ctx->d_table_poison, 0, 16, ctx->stream));
    // what the first scan would otherwise allocate in front of its kernels (~0.1 ms per hipMalloc / event): a first piece of
    // scratch (survivor / band lists of a text up to ~2 GiB) and one recycled hit block of the default capacity
    {
        hits_block b;
        b.cap = 1ull << 20;
        bool ok = hipMalloc(&b.d_hits, b.cap * sizeof(spm_hit)) == hipSuccess &&
                  hipMalloc(&b.d_count, 16 * sizeof(unsigned long long)) == hipSuccess;
        for (int e = 0; e < 4 && ok; ++e)
            ok = hipEventCreate(&b.ev[e]) == hipSuccess;
        // (what a deferred scan adds: its own pinned counter block -- a pinned allocation costs ~0.1 ms -- and an event)
        if (ok && hipHostMalloc(reinterpret_cast<void **>(&b.h_c), 16 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
            b.h_c = nullptr;
        if (ok && hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming) != hipSuccess)
            b.ev_done = nullptr;
        if (ok && hipMemsetAsync(b.d_count, 0, 16 * sizeof(unsigned long long), ctx->stream) == hipSuccess) {
            b.zeroed = true;
            ctx->pool.push_back(b);
        } else {
            hipFree(b.d_hits);
            hipFree(b.d_count);
            for (int e = 0; e < 4; ++e)
                if (b.ev[e])
                    hipEventDestroy(b.ev[e]);
            if (b.h_c)
                hipHostFree(b.h_c);
            if (b.ev_done)
                hipEventDestroy(b.ev_done);
        }
    }
    spm_warm_text_kernels();
    spm_warm_brute_kernels();
    spm_warm_filter_kernels();
    spm_warm_hits_kernels();
    spm_warm_jst_kernels();
    *out = ctx.release();
    return SPM_OK;
}

extern "C" void spm_hip_destroy(spm_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (hits_block &b : ctx->pool) {
        hipFree(b.d_hits);
        hipFree(b.d_count);
        for (int e = 0; e < 4; ++e)
            if (b.ev[e])
                hipEventDestroy(b.ev[e]);
        if (b.h_c)
            hipHostFree(b.h_c);
        if (b.ev_done)
            hipEventDestroy(b.ev_done);
    }
    for (auto &b : ctx->jst_pool)
        hipFree(b.first);
    hipFree(ctx->d_table_poison);
    if (ctx->h_counters)
        hipHostFree(ctx->h_counters);
    if (ctx->h_stage)
        hipHostFree(ctx->h_stage);
    for (int e = 0; e < 2; ++e)
        if (ctx->stage_ev[e])
            hipEventDestroy(ctx->stage_ev[e]);
    if (ctx->own_stream)
        hipStreamDestroy(ctx->stream);
    hipFree(ctx->d_scratch);
    hipFree(ctx->d_band_tab);
    delete ctx;
}

extern "C" const char *spm_hip_last_error(const spm_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : spm_hip::g_init_error.c_str();
}

extern "C" int spm_hip_synchronize(spm_ctx *ctx)
{
    if (!ctx)
        return SPM_E_INVALID;
    SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SPM_OK;
}

int text_alloc(spm_ctx *ctx, uint64_t n, uint32_t sigma, spm_text **out)
{
    std::unique_ptr<spm_text> t(new spm_text);
    t->ctx = ctx;
    t->n = n;
    t->sigma = sigma;
    t->alloc = ((n + 1023) & ~1023ull) + 1024; // the kernels never read past `n`, padding is slack only
    t->owned = true;
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SPM_HIP_CHECK(ctx, hipMalloc(&t->d, t->alloc));
    *out = t.release();
    return SPM_OK;
}

extern "C" int spm_hip_text_upload(spm_ctx *ctx, const uint8_t *ranks, uint64_t n, uint32_t sigma, spm_text **out)
{
    if (!ctx || !out || (n && !ranks) || sigma < 2 || sigma > 255) {
        SPM_SET_ERR(ctx, "spm_hip_text_upload: invalid argument");
        return SPM_E_INVALID;
    }
    for (uint64_t i = 0; i < n; ++i)
        if (ranks[i] >= sigma) {
            SPM_SET_ERR(ctx, "spm_hip_text_upload: symbol %u at %llu is not a rank < sigma=%u", ranks[i],
                        (unsigned long long)i, sigma);
            return SPM_E_INVALID;
        }
    spm_text *t = nullptr;
    int rc = text_alloc(ctx, n, sigma, &t);
    if (rc != SPM_OK)
        return rc;
    if (n) {
        hipError_t e = hipMemcpyAsync(t->d, ranks, n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            SPM_SET_ERR(ctx, "text upload failed: %s", hipGetErrorString(e));
            spm_hip_text_destroy(t);
            return SPM_E_HIP;
        }
    }
    *out = t;
    return SPM_OK;
}

extern "C" int spm_hip_text_wrap(spm_ctx *ctx, const void *device_ranks, uint64_t n, uint32_t sigma, spm_text **out)
{
    if (!ctx || !out || (n && !device_ranks) || ((uintptr_t)device_ranks & 15) || sigma < 2 || sigma > 255) {
        SPM_SET_ERR(ctx, "spm_hip_text_wrap: invalid argument (pointer must be 16-byte aligned)");
        return SPM_E_INVALID;
    }
    if (n) { // one pass over the borrowed buffer: every symbol must be a rank < sigma (as spm_hip_text_upload checks)
        SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        unsigned int *d_bad = nullptr;
        dev_scratch tmp;
        SPM_HIP_CHECK(ctx, tmp.alloc(&d_bad, sizeof(unsigned int)));
        SPM_HIP_CHECK(ctx, hipMemsetAsync(d_bad, 0, sizeof(unsigned int), ctx->stream));
        const uint64_t n_q = (n + 15) / 16;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_q + 255) / 256, (uint64_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(text_validate_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t *)device_ranks, n,
                           sigma, d_bad);
        SPM_HIP_CHECK(ctx, hipGetLastError());
        unsigned int bad = 0;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) {
            SPM_SET_ERR(ctx, "spm_hip_text_wrap: the buffer holds symbols that are not ranks < sigma=%u", sigma);
            return SPM_E_INVALID;
        }
    }
    spm_text *t = new spm_text;
    t->ctx = ctx;
    t->d = (uint8_t *)device_ranks;
    t->n = n;
    t->alloc = n;
    t->sigma = sigma;
    t->owned = false;
    *out = t;
    return SPM_OK;
}

extern "C" int spm_hip_text_generate(spm_ctx *ctx, uint64_t seed, uint64_t global_begin, uint64_t n, spm_text **out)
{
    if (!ctx || !out || (global_begin & 31)) {
        SPM_SET_ERR(ctx, "spm_hip_text_generate: global_begin must be a multiple of 32");
        return SPM_E_INVALID;
    }
    spm_text *t = nullptr;
    int rc = text_alloc(ctx, n, 4, &t);
    if (rc != SPM_OK)
        return rc;
    if (n) {
        const uint64_t n_words = (n + 31) / 32;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_words + 255) / 256, (uint64_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(synth_text_kernel, dim3(grid), dim3(256), 0, ctx->stream, t->d, seed, global_begin, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            SPM_SET_ERR(ctx, "text generate failed: %s", hipGetErrorString(e));
            spm_hip_text_destroy(t);
            return SPM_E_HIP;
        }
    }
    *out = t;
    return SPM_OK;
}

extern "C" int spm_hip_text_generate_repeats(spm_ctx *ctx, uint64_t seed, uint64_t global_begin, uint64_t n,
                                             uint32_t repeat_ppm, spm_text **out)
{
    if (!ctx || !out || (global_begin & 31) || repeat_ppm > 130000) {
        SPM_SET_ERR(ctx, "spm_hip_text_generate_repeats: global_begin must be a multiple of 32, repeat_ppm <= 130000");
        return SPM_E_INVALID;
    }
    spm_text *t = nullptr;
    int rc = text_alloc(ctx, n, 4, &t);
    if (rc != SPM_OK)
        return rc;
    if (n) {
        const uint64_t n_q = (n + 15) / 16;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_q + 255) / 256, (uint64_t)ctx->n_cu * 32);
        hipLaunchKernelGGL(synth_repeat_text_kernel, dim3(grid), dim3(256), 0, ctx->stream, t->d, seed, repeat_ppm,
                           global_begin, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            SPM_SET_ERR(ctx, "text generate failed: %s", hipGetErrorString(e));
            spm_hip_text_destroy(t);
            return SPM_E_HIP;
        }
    }
    *out = t;
    return SPM_OK;
}

extern "C" int spm_hip_text_pack(spm_ctx *ctx, spm_text *text)
{
    if (!ctx || !text) {
        SPM_SET_ERR(ctx, "spm_hip_text_pack: invalid argument");
        return SPM_E_INVALID;
    }
    if (text->sigma != 4) {
        SPM_SET_ERR(ctx, "spm_hip_text_pack: only dna4 haystacks have a 2-bit encoding (sigma = %u)", text->sigma);
        return SPM_E_INVALID;
    }
    if (text->d_packed)
        return SPM_OK;
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t n_words = (text->n + 15) / 16;
    const uint64_t padded = ((text->n + 4095) / 4096) * 256 + 2048; // whole p-chunks + one group of slack
    uint32_t *d = nullptr;
    unsigned int *d_bad = nullptr;
    SPM_HIP_CHECK(ctx, hipMalloc(&d, padded * sizeof(uint32_t)));
    SPM_HIP_CHECK(ctx, hipMalloc(&d_bad, sizeof(unsigned int)));
    SPM_HIP_CHECK(ctx, hipMemsetAsync(d, 0, padded * sizeof(uint32_t), ctx->stream));
    SPM_HIP_CHECK(ctx, hipMemsetAsync(d_bad, 0, sizeof(unsigned int), ctx->stream));
    if (n_words) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_words + 255) / 256, (uint64_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(text_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, text->d,
                           text->owned ? std::min(text->alloc, text->n) : text->n, d, n_words, d_bad);
        SPM_HIP_CHECK(ctx, hipGetLastError());
    }
    unsigned int bad = 0;
    SPM_HIP_CHECK(ctx, hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
    SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(d_bad);
    if (bad) {
        hipFree(d);
        SPM_SET_ERR(ctx, "spm_hip_text_pack: the text holds symbols that are not dna4 ranks (>= 4)");
        return SPM_E_INVALID;
    }
    text->d_packed = d;
    text->packed_words = padded;
    return SPM_OK;
}

extern "C" int spm_hip_text_is_packed(const spm_text *text) { return text && text->d_packed ? 1 : 0; }

extern "C" int spm_hip_text_download(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t n, uint8_t *dst)
{
    if (!ctx || !text || begin + n > text->n || (n && !dst)) {
        SPM_SET_ERR(ctx, "spm_hip_text_download: invalid argument");
        return SPM_E_INVALID;
    }
    if (n) {
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(dst, text->d + begin, n, hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SPM_OK;
}

extern "C" uint64_t spm_hip_text_length(const spm_text *t) { return t ? t->n : 0; }
extern "C" const void *spm_hip_text_device_ptr(const spm_text *t) { return t ? t->d : nullptr; }

extern "C" void spm_hip_text_destroy(spm_text *t)
{
    if (!t)
        return;
    if (t->owned)
        hipFree(t->d);
    hipFree(t->d_packed);
    delete t;
}

extern "C" uint64_t spm_hip_synth_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p,
                                          uint32_t L, uint32_t kmax, uint8_t *out)
{
    return synth_pattern(seed_text, seed_pat, n_total, p, L, kmax, out);
}

extern "C" uint64_t spm_hip_synth_repeat_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p,
                                                 uint32_t L, uint32_t kmax, uint32_t repeat_ppm, uint32_t across_every,
                                                 uint8_t *out)
{
    return synth_repeat_pattern(seed_text, seed_pat, n_total, p, L, kmax, repeat_ppm, across_every, out);
}

extern "C" void spm_hip_synth_repeat_text(uint64_t seed, uint32_t repeat_ppm, uint64_t begin, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i)
        out[i] = repeat_base(seed, repeat_ppm, begin + i);
}

extern "C" uint64_t spm_hip_mix64(uint64_t z) { return mix64(z); }

extern "C" const char *spm_hip_version(void) { return "libspm_hip 0.2 (gfx950)"; }

void spm_warm_text_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)synth_text_kernel);
}
