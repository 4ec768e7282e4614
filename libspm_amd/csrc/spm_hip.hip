// spm_hip.hip -- C ABI of libspm_hip.so (see include/spm_hip.h) and the host side of the scan engines.
//
// Host responsibilities (what the reference does in its matcher constructors and in
// seqan_pattern_base::operator(), /root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-71):
//   * build the per-needle bit-mask tables once (patterns_create)  -- [upstream] _patternFirstInit
//   * choose engine, tile the haystack, launch, collect hits       -- the find loop
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "brute.hpp"
#include "common.hpp"
#include "filter.hpp"
#include "index_build.hpp"
#include "synth.hpp"
#include "tables_build.hpp"

namespace spm_hip
{
thread_local std::string g_init_error;
}

using namespace spm_hip;

// ----------------------------------------------------------------------------------------------------
// pattern set
// ----------------------------------------------------------------------------------------------------
struct spm_patterns : spm_hip::seed_index // (the seed index: passes, entries, seed layout -- index_build.hpp)
{
    spm_ctx *ctx = nullptr;
    int algo = 0;
    uint32_t n = 0;
    uint32_t sigma = 4;
    std::vector<uint8_t> ranks;
    std::vector<uint32_t> offsets;
    std::vector<int32_t> m, k; // padded to n_groups*64
    uint32_t n_groups = 0;
    uint32_t max_m = 0;
    uint32_t max_window = 0;
    uint32_t max_k = 0;
    uint32_t NW = 1;   // 32-bit words per needle in the brute kernels (power of two)
    bool is_myers() const { return algo == SPM_ALGO_MYERS || algo == SPM_ALGO_MYERS_PREFIX; }
    // device
    uint32_t *d_peq = nullptr; // [group][sigma+1][NW][64], needles top-aligned
    uint32_t *d_peq_bot = nullptr; // Myers only: same shape, needles bottom-aligned (cut-off kernel)
    uint32_t *d_peq_verify = nullptr; // exact matchers: Myers-style match masks (top-aligned) for the verify kernel;
                                      // for Myers sets verification reads d_peq itself
    uint32_t *d_hp0 = nullptr; // prefix: [group][NW][64]
    int32_t *d_m = nullptr;
    int32_t *d_k = nullptr;
    mutable uint64_t cand_hint = 0; // most candidates a filter scan of this set has produced so far
    uint8_t *d_surplus = nullptr;   // per needle: seeds - k (candidate merging); nullptr = no needle has k >= kMergeMinK
    uint8_t *d_ranks = nullptr;     // filterable sets: the needles' symbols, back to back (whole-seed check of a candidate)
    uint32_t *d_offsets = nullptr;  // ... and where each needle starts
    uint32_t *d_needle_pk = nullptr;  // dna4 sets: the needles 2 bits per symbol, 16 per word (piece count of a candidate)
    uint32_t *d_pk_offsets = nullptr; // ... and the first word of each
    uint16_t *d_seed_q = nullptr;
    pass_entry *d_pass_tab = nullptr; // the passes' key directories, for resolve_kernel
    uint4 *d_entries = nullptr;
    mutable uint64_t hit_hint = 0;    // most hits a filter scan of this set has reported so far (sizes the dedupe set)
    mutable bool scanned = false;     // the hints come from at least one completed filter scan
    mutable int exact_whole = -1;     // 1: k = 0 and every needle is its own single seed (decided at the first scan)
    mutable uint64_t band_hint = 0;   // ... and band-list slots it drew (sizes the verification grid)
    spm_build_stats build{};          // what spm_hip_patterns_create spent where
};

using clk = std::chrono::steady_clock;
static float ms_since(clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); }
// SPM_HIP_TRACE=1: one stderr line per C-ABI call that does work, with its timings
static bool spm_trace_on()
{
    const char *v = getenv("SPM_HIP_TRACE");
    return v && *v && *v != '0';
}

extern "C" int spm_hip_patterns_create(spm_ctx *ctx, int algo, const uint8_t *ranks_concat, const uint32_t *offsets,
                                       uint32_t n_patterns, const uint16_t *k, uint32_t sigma, spm_patterns **out)
{
    if (!ctx || !out || (n_patterns && (!offsets || !ranks_concat)) || sigma < 2 || sigma > 255 || algo < 0 ||
        algo > SPM_ALGO_HORSPOOL) {
        SPM_SET_ERR(ctx, "spm_hip_patterns_create: invalid argument");
        return SPM_E_INVALID;
    }
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const auto t_begin = clk::now();
    std::unique_ptr<spm_patterns, void (*)(spm_patterns *)> ps(new spm_patterns, spm_hip_patterns_destroy); // (error paths free the device side too)
    ps->ctx = ctx;
    ps->algo = algo;
    ps->n = n_patterns;
    ps->sigma = sigma;
    ps->offsets.assign(offsets, offsets + n_patterns + 1);
    ps->ranks.assign(ranks_concat, ranks_concat + (n_patterns ? offsets[n_patterns] : 0));
    ps->n_groups = std::max(1u, (n_patterns + 63) / 64);
    ps->m.assign((size_t)ps->n_groups * 64, 0);
    ps->k.assign((size_t)ps->n_groups * 64, -1);
    for (uint32_t p = 0; p < n_patterns; ++p) {
        if (offsets[p + 1] < offsets[p]) {
            SPM_SET_ERR(ctx, "spm_hip_patterns_create: offsets not ascending");
            return SPM_E_INVALID;
        }
        const uint32_t m = offsets[p + 1] - offsets[p];
        if (m > SPM_MAX_NEEDLE) {
            SPM_SET_ERR(ctx, "needle %u has %u symbols; limit is %u", p, m, SPM_MAX_NEEDLE);
            return SPM_E_UNSUPPORTED;
        }
        const uint32_t kk = (ps->is_myers() && k) ? k[p] : 0;
        ps->m[p] = (int32_t)m;
        ps->k[p] = (int32_t)kk;
        ps->max_m = std::max(ps->max_m, m);
        ps->max_k = std::max(ps->max_k, kk);
        ps->max_window = std::max(ps->max_window, m + kk);
    }
    ps->NW = next_pow2(std::max(1u, (ps->max_m + 31) / 32));
    const uint32_t NW = ps->NW;
    const index_tuning tune = index_tuning::from_env();
    ps->build.threads = tune.n_threads();
    needle_view nv;
    nv.algo = algo;
    nv.n = n_patterns;
    nv.sigma = sigma;
    nv.ranks = ps->ranks.data();
    nv.offsets = ps->offsets.data();
    nv.m = ps->m.data();
    nv.k = ps->k.data();
    nv.max_k = ps->max_k;
    auto upload = [&](auto **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(bytes, 16));
        if (e == hipSuccess && bytes)
            e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    double ms_upload = 0;
    auto t0 = clk::now();

    // ---- match-mask tables: [group][row][word][lane], needles top-aligned (see brute.hpp) ----
    {
        brute_tables bt;
        build_brute_tables(nv, ps->n_groups, NW, sigma <= 5 || sigma == 15, tune.n_threads(), bt);
        ps->build.ms_tables = ms_since(t0);
        const auto tu = clk::now();
        SPM_HIP_CHECK(ctx, upload(&ps->d_peq, bt.peq.data(), bt.peq.size() * sizeof(uint32_t)));
        if (!bt.verify.empty()) // the filter engine verifies exact matchers with the Myers recurrence at k = 0
            SPM_HIP_CHECK(ctx, upload(&ps->d_peq_verify, bt.verify.data(), bt.verify.size() * sizeof(uint32_t)));
        if (!bt.bot.empty())
            SPM_HIP_CHECK(ctx, upload(&ps->d_peq_bot, bt.bot.data(), bt.bot.size() * sizeof(uint32_t)));
        if (!bt.hp0.empty())
            SPM_HIP_CHECK(ctx, upload(&ps->d_hp0, bt.hp0.data(), bt.hp0.size() * sizeof(uint32_t)));
        SPM_HIP_CHECK(ctx, upload(&ps->d_m, ps->m.data(), ps->m.size() * sizeof(int32_t)));
        SPM_HIP_CHECK(ctx, upload(&ps->d_k, ps->k.data(), ps->k.size() * sizeof(int32_t)));
        ms_upload += ms_since(tu);
    }

    // ---- filter engine tables (verification reads the brute table) ----
    if ((sigma == 4 || sigma == 5 || sigma == 15) && algo != SPM_ALGO_MYERS_PREFIX && n_patterns > 0) {
        const auto ti = clk::now();
        int rc = build_filter_index(nv, tune, *ps);
        if (rc != SPM_OK)
            return rc;
        ps->build.ms_index = ms_since(ti);
        const auto tu = clk::now();
        if (!ps->fidx.empty()) {
            std::vector<pass_entry> pt;
            for (filter_index &F : ps->fidx) {
                SPM_HIP_CHECK(ctx, upload(&F.d_bitmap, F.h_image.data(), F.h_image.size() * sizeof(uint32_t)));
                SPM_HIP_CHECK(ctx, upload(&F.d_ht, F.h_ht.data(), F.h_ht.size() * sizeof(u32x4)));
                if (F.dense)
                    SPM_HIP_CHECK(ctx, upload(&F.d_buckets, F.h_buckets.data(), F.h_buckets.size() * sizeof(uint16_t)));
                ps->build.keys += F.n_keys;
                for (uint32_t d = 0; d < 16; ++d)
                    ps->build.anchor_sixteenths += (F.dimer_set >> d) & 1u;
                F.h_image = std::vector<uint32_t>();
                F.h_ht = std::vector<u32x4>();
                F.h_buckets = std::vector<uint16_t>();
                pt.push_back(pass_entry{reinterpret_cast<const uint4 *>(F.d_ht), F.ht_mask, 0});
            }
            SPM_HIP_CHECK(ctx, upload(&ps->d_pass_tab, pt.data(), pt.size() * sizeof(pass_entry)));
            SPM_HIP_CHECK(ctx, upload(&ps->d_entries, ps->h_entries.data(), ps->h_entries.size() * sizeof(u32x4)));
            ps->h_entries = std::vector<u32x4>();
            const size_t nr = ps->ranks.size() + 64; // padded: resolve_kernel reads whole dwords around a seed
            SPM_HIP_CHECK(ctx, hipMalloc(&ps->d_ranks, nr));
            SPM_HIP_CHECK(ctx, hipMemset(ps->d_ranks, 0, nr));
            if (!ps->ranks.empty())
                SPM_HIP_CHECK(ctx, hipMemcpy(ps->d_ranks, ps->ranks.data(), ps->ranks.size(), hipMemcpyHostToDevice));
            SPM_HIP_CHECK(ctx, upload(&ps->d_offsets, ps->offsets.data(), ps->offsets.size() * sizeof(uint32_t)));
            if (ps->sigma == 4) { // the same symbols 2 bits each, 16 per word, every needle from a word of its own
                std::vector<uint32_t> pk, pk_off;
                pack_needles(nv, pk, pk_off);
                SPM_HIP_CHECK(ctx, upload(&ps->d_needle_pk, pk.data(), pk.size() * sizeof(uint32_t)));
                SPM_HIP_CHECK(ctx, upload(&ps->d_pk_offsets, pk_off.data(), pk_off.size() * sizeof(uint32_t)));
            }
            SPM_HIP_CHECK(ctx, upload(&ps->d_seed_q, ps->seed_q.data(), ps->seed_q.size() * sizeof(uint16_t)));
        }
        if (!ps->fidx.empty() && ps->max_k >= kMergeMinK && ps->max_k <= 1000) {
            std::vector<uint8_t> surplus(ps->m.size(), 1);
            for (uint32_t p = 0; p < ps->n; ++p)
                surplus[p] = (uint8_t)(ps->seed_n[p] - (uint32_t)ps->k[p]);
            SPM_HIP_CHECK(ctx, upload(&ps->d_surplus, surplus.data(), surplus.size()));
        }
        ms_upload += ms_since(tu);
    }
    ps->build.ms_upload = (float)ms_upload;
    ps->build.ms_total = ms_since(t_begin);
    ps->build.passes = (uint32_t)ps->fidx.size();
    ps->build.dense = ps->filter_dense ? 1u : 0u;
    ps->build.stride = ps->filter_stride;
    ps->build.key_len = ps->filter_key_len;
    if (spm_trace_on())
        fprintf(stderr, "[spm_hip] patterns_create: %u needles, algo %d, sigma %u -> %u pass(es)%s, %llu keys, %u/16 of the dimers; "
                        "%.2f ms (tables %.2f, index %.2f, upload %.2f; %u threads)\n",
                n_patterns, algo, sigma, ps->build.passes, ps->build.dense ? " dense" : "", (unsigned long long)ps->build.keys,
                ps->build.anchor_sixteenths, ps->build.ms_total, ps->build.ms_tables, ps->build.ms_index, ps->build.ms_upload,
                ps->build.threads);
    *out = ps.release();
    return SPM_OK;
}

extern "C" int spm_hip_patterns_build_stats(const spm_patterns *p, spm_build_stats *out)
{
    if (!p || !out)
        return SPM_E_INVALID;
    *out = p->build;
    return SPM_OK;
}

extern "C" void spm_hip_patterns_destroy(spm_patterns *p)
{
    if (!p)
        return;
    hipFree(p->d_peq);
    hipFree(p->d_peq_bot);
    hipFree(p->d_peq_verify);
    hipFree(p->d_hp0);
    hipFree(p->d_m);
    hipFree(p->d_k);
    hipFree(p->d_surplus);
    hipFree(p->d_ranks);
    hipFree(p->d_offsets);
    hipFree(p->d_needle_pk);
    hipFree(p->d_pk_offsets);
    hipFree(p->d_pass_tab);
    hipFree(p->d_entries);
    hipFree(p->d_seed_q);
    for (filter_index &F : p->fidx) {
        hipFree(F.d_bitmap);
        hipFree(F.d_ht);
        hipFree(F.d_buckets);
    }
    delete p;
}

extern "C" uint64_t spm_hip_patterns_window_size(const spm_patterns *p, uint32_t pattern)
{
    if (!p || pattern >= p->n || p->m[pattern] == 0)
        return 0;
    return (uint64_t)p->m[pattern] + (p->is_myers() ? (uint64_t)p->k[pattern] : 0);
}

extern "C" int spm_hip_patterns_filterable(const spm_patterns *p) { return p && !p->fidx.empty() ? 1 : 0; }

// ---- state blobs -------------------------------------------------------------------------------------
static uint32_t abi_words(const spm_patterns *p)
{
    return p->is_myers() ? std::max(1u, (p->max_m + 63) / 64) : std::max(1u, (p->max_m + 31) / 32);
}

extern "C" size_t spm_hip_patterns_state_stride(const spm_patterns *p)
{
    if (!p)
        return 0;
    const uint32_t nw = abi_words(p);
    return p->is_myers() ? 8 + (size_t)16 * nw : 8 + (size_t)4 * ((nw + 1) & ~1u);
}

static void set_bits(std::vector<uint32_t> &v, uint32_t lo, uint32_t hi)
{
    for (uint32_t b = lo; b < hi; ++b)
        v[b / 32] |= 1u << (b % 32);
}

extern "C" int spm_hip_patterns_state_init(const spm_patterns *p, void *state)
{
    if (!p || !state)
        return SPM_E_INVALID;
    const size_t stride = spm_hip_patterns_state_stride(p);
    const uint32_t nw = abi_words(p);
    memset(state, 0, stride * p->n);
    for (uint32_t i = 0; i < p->n; ++i) {
        uint8_t *rec = (uint8_t *)state + stride * i;
        const uint32_t m = (uint32_t)p->m[i];
        if (p->is_myers()) {
            int32_t score = (int32_t)m;
            memcpy(rec, &score, 4);
            memcpy(rec + 4, &nw, 4);
            uint64_t *vp = (uint64_t *)(rec + 8);
            for (uint32_t b = 0; b < m; ++b)
                vp[b / 64] |= 1ull << (b % 64);
        } else {
            memcpy(rec, &nw, 4);
            uint32_t *r = (uint32_t *)(rec + 8);
            for (uint32_t w = 0; w < nw; ++w)
                r[w] = 0xFFFFFFFFu;
        }
    }
    return SPM_OK;
}

// ABI state -> internal [group][rows][64] layout (top-aligned) and back
static void state_to_internal(const spm_patterns *p, const void *state, std::vector<uint32_t> &out)
{
    const uint32_t NW = p->NW, nw = abi_words(p);
    const size_t stride = spm_hip_patterns_state_stride(p);
    const uint32_t rows = p->is_myers() ? 2 * NW + 1 : NW;
    out.assign((size_t)p->n_groups * rows * 64, 0);
    for (uint32_t i = 0; i < p->n_groups * 64; ++i) {
        const uint32_t g = i / 64, l = i % 64;
        const uint32_t m = i < p->n ? (uint32_t)p->m[i] : 0;
        const uint32_t off = NW * 32 - m;
        auto at = [&](uint32_t row) -> uint32_t & { return out[((size_t)g * rows + row) * 64 + l]; };
        if (i >= p->n || m == 0) {
            if (p->is_myers())
                at(2 * NW) = 0x3FFFFFFF; // score that never reaches k
            else
                for (uint32_t w = 0; w < NW; ++w)
                    at(w) = 0xFFFFFFFFu;
            continue;
        }
        const uint8_t *rec = (const uint8_t *)state + stride * i;
        if (p->is_myers()) {
            int32_t score;
            memcpy(&score, rec, 4);
            const uint64_t *vp = (const uint64_t *)(rec + 8);
            const uint64_t *vn = vp + nw;
            for (uint32_t j = 0; j < m; ++j) {
                const uint32_t b = off + j;
                if ((vp[j / 64] >> (j % 64)) & 1)
                    at(b / 32) |= 1u << (b % 32);
                if ((vn[j / 64] >> (j % 64)) & 1)
                    at(NW + b / 32) |= 1u << (b % 32);
            }
            at(2 * NW) = (uint32_t)score;
        } else {
            const uint32_t *r = (const uint32_t *)(rec + 8);
            for (uint32_t w = 0; w < NW; ++w)
                at(w) = 0;
            for (uint32_t j = 0; j < m; ++j) {
                const uint32_t b = off + j;
                if ((r[j / 32] >> (j % 32)) & 1)
                    at(b / 32) |= 1u << (b % 32);
            }
        }
    }
}

static void state_from_internal(const spm_patterns *p, const std::vector<uint32_t> &in, void *state)
{
    const uint32_t NW = p->NW, nw = abi_words(p);
    const size_t stride = spm_hip_patterns_state_stride(p);
    const uint32_t rows = p->is_myers() ? 2 * NW + 1 : NW;
    memset(state, 0, stride * p->n);
    for (uint32_t i = 0; i < p->n; ++i) {
        const uint32_t g = i / 64, l = i % 64;
        const uint32_t m = (uint32_t)p->m[i];
        const uint32_t off = NW * 32 - m;
        auto at = [&](uint32_t row) -> uint32_t { return in[((size_t)g * rows + row) * 64 + l]; };
        uint8_t *rec = (uint8_t *)state + stride * i;
        if (p->is_myers()) {
            int32_t score = m ? (int32_t)at(2 * NW) : 0;
            memcpy(rec, &score, 4);
            memcpy(rec + 4, &nw, 4);
            uint64_t *vp = (uint64_t *)(rec + 8);
            uint64_t *vn = vp + nw;
            for (uint32_t j = 0; j < m; ++j) {
                const uint32_t b = off + j;
                if ((at(b / 32) >> (b % 32)) & 1)
                    vp[j / 64] |= 1ull << (j % 64);
                if ((at(NW + b / 32) >> (b % 32)) & 1)
                    vn[j / 64] |= 1ull << (j % 64);
            }
        } else {
            memcpy(rec, &nw, 4);
            uint32_t *r = (uint32_t *)(rec + 8);
            for (uint32_t w = 0; w < nw; ++w)
                r[w] = 0xFFFFFFFFu; // bits >= |P| stay set, as in SeqAn's masks
            for (uint32_t j = 0; j < m; ++j) {
                const uint32_t b = off + j;
                if (!((at(b / 32) >> (b % 32)) & 1))
                    r[j / 32] &= ~(1u << (j % 32));
            }
        }
    }
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
    }
    for (auto &b : ctx->jst_pool)
        hipFree(b.first);
    if (ctx->h_counters)
        hipHostFree(ctx->h_counters);
    if (ctx->fused_hdr_ev)
        hipEventDestroy(ctx->fused_hdr_ev);
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

static int text_alloc(spm_ctx *ctx, uint64_t n, uint32_t sigma, spm_text **out)
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

// ----------------------------------------------------------------------------------------------------
// scan
// ----------------------------------------------------------------------------------------------------
namespace
{

struct scan_args
{
    spm_ctx *ctx;
    const spm_text *text;
    uint64_t begin, end, ctx_begin;
    const spm_patterns *ps;
    spm_scan_opts opts;
    const void *state_in;
    void *state_out;
    spm_hits *hits;
    const uint64_t *seg_offsets = nullptr; // host; n_segments + 1 entries
    uint64_t n_segments = 0;
    const uint64_t *d_seg_offsets = nullptr; // the same table already resident on the device (journaled-sequence index)
    const uint32_t *d_seg_owned = nullptr;   // optional per-segment offset of the first wanted end symbol (filter engine)
    uint64_t cand_cap_override = 0;          // retry after a survivor overflow: the count the first attempt needed
    uint64_t band_scale = 0;                 // retry after a band list / table overflow: that much more room
    bool seen_full = false;                  // retry after a dedupe-set overflow: size it for the caller's hit buffer
    std::vector<uint64_t> seg_host;          // host copy fetched on demand when only the device table was given
    // span-local fallback: the filter run leaves these for the brute-force re-scan of the spans that gave up
    unsigned long long *d_seen = nullptr;
    uint32_t seen_mask = 0;
    uint64_t *d_ovf = nullptr;               // overflow list in the scratch buffer: {begin, symbols} per span
    const std::vector<uint64_t> *tiles = nullptr; // brute pass over an explicit tile table {scan_lo, own_lo, own_hi}
};

constexpr uint64_t kOvfCap = 1ull << 17; // spans the overflow list holds (2 MiB); beyond: whole-scan fallback

template <int NW>
void launch_brute_nw(const spm_patterns *ps, const brute_params &P, dim3 grid, dim3 block, size_t lds,
                     hipStream_t stream, bool cutoff)
{
    if (cutoff) {
        hipFuncSetAttribute((const void *)myers_cutoff_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_cutoff_kernel<NW>), grid, block, lds, stream, P);
    } else if (ps->algo == SPM_ALGO_MYERS) {
        hipFuncSetAttribute((const void *)myers_brute_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_brute_kernel<NW, false>), grid, block, lds, stream, P);
    } else if (ps->algo == SPM_ALGO_MYERS_PREFIX) {
        hipFuncSetAttribute((const void *)myers_brute_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_brute_kernel<NW, true>), grid, block, lds, stream, P);
    } else {
        hipFuncSetAttribute((const void *)shiftor_brute_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((shiftor_brute_kernel<NW>), grid, block, lds, stream, P);
    }
}

void launch_brute(const spm_patterns *ps, const brute_params &P, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                  bool cutoff)
{
    switch (ps->NW) {
    case 1: launch_brute_nw<1>(ps, P, grid, block, lds, s, cutoff); break;
    case 2: launch_brute_nw<2>(ps, P, grid, block, lds, s, cutoff); break;
    case 4: launch_brute_nw<4>(ps, P, grid, block, lds, s, cutoff); break;
    case 8: launch_brute_nw<8>(ps, P, grid, block, lds, s, cutoff); break;
    case 16: launch_brute_nw<16>(ps, P, grid, block, lds, s, cutoff); break;
    case 32: launch_brute_nw<32>(ps, P, grid, block, lds, s, cutoff); break;
    default: launch_brute_nw<64>(ps, P, grid, block, lds, s, cutoff); break;
    }
}

template <int NWN>
void launch_verify_nw(const verify_params &V, dim3 grid, hipStream_t s)
{
    // LDS holds (sigma+1)*NWN words per thread; keep the block within ~128 KiB
    uint32_t threads = 256;
    const size_t per_thread = (size_t)(V.sigma + 1) * NWN * 4 + 2 * (2 * (size_t)V.max_k + 1 + V.max_span);
    while (threads > 64 && per_thread * threads > 128 * 1024)
        threads >>= 1;
    const size_t lds = per_thread * threads;
    hipFuncSetAttribute((const void *)verify_kernel<NWN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((verify_kernel<NWN>), grid, dim3(threads), lds, s, V);
}

template <int G, int NB>
void launch_verify_wave_g(verify_params V, const uint32_t *peq_bot, uint32_t max_m, dim3 grid, hipStream_t s)
{
    // One wave per workgroup: a scan leaves a few thousand long bands, i.e. far fewer busy waves than the GPU has SIMDs,
    // and each is a serial chain ~1500 steps long.  With four-wave workgroups filled in order, the dispatcher packed the
    // busy waves four to a SIMD on a third of the CUs and left the rest idle.
    const uint32_t threads = (uint32_t)std::max(64, std::min(256, env_int("SPM_HIP_VERIFY_WAVE_THREADS", 64)));
    grid.x *= 256 / threads;
    const uint32_t n_slots = 2 * V.max_k + 1 + V.max_span;
    // text window of one candidate: cold start |P| + k symbols before the first end position, then the end positions
    V.wave_text = ((max_m + V.max_k + n_slots + 16 + 15) & ~15u) + 16;
    const size_t per_group = ((n_slots * 2 + 15) & ~15u) + V.wave_text;
    size_t lds = (size_t)(threads / 64) * (64 / G) * per_group;
    // (diagnostics: a larger LDS claim per workgroup caps how many of them a CU takes at once)
    lds = std::max<size_t>(lds, (size_t)std::max(0, std::min(160, env_int("SPM_HIP_VERIFY_WAVE_LDS_KB", 0))) * 1024);
    hipFuncSetAttribute((const void *)verify_wave_kernel<G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((verify_wave_kernel<G, NB>), grid, dim3(threads), lds, s, V, peq_bot);
}

// long needles: NB 32-row blocks per lane, G = lanes per band >= blocks of the longest needle / NB.  SPM_HIP_VERIFY_WAVE_NB=2:
// two blocks per lane from 17 blocks on (four |P| = 1024 bands share a wave instead of two).
void launch_verify_wave(uint32_t n_blocks, const verify_params &V, const uint32_t *peq_bot, uint32_t max_m, dim3 grid,
                        hipStream_t s)
{
    // (measured on C5, 2 798 bands of |P| = 1024: one block per lane 0.300 ms, two 0.358 -- a step is a chain of dependent
    // instructions, its latency and not its issue slots set the pace, and the second block lengthens the chain.)
    const bool two = env_int("SPM_HIP_VERIFY_WAVE_NB", 1) >= 2;
    if (n_blocks <= 8)
        launch_verify_wave_g<8, 1>(V, peq_bot, max_m, grid, s);
    else if (n_blocks <= 16)
        launch_verify_wave_g<16, 1>(V, peq_bot, max_m, grid, s);
    else if (n_blocks <= 32) {
        if (two)
            launch_verify_wave_g<16, 2>(V, peq_bot, max_m, grid, s);
        else
            launch_verify_wave_g<32, 1>(V, peq_bot, max_m, grid, s);
    } else {
        if (two)
            launch_verify_wave_g<32, 2>(V, peq_bot, max_m, grid, s);
        else
            launch_verify_wave_g<64, 1>(V, peq_bot, max_m, grid, s);
    }
}

// nwn = 32-bit words that can hold needle rows = ceil(max |P| / 32), rounded up to an instantiated width
void launch_verify(uint32_t nwn, const verify_params &V, dim3 grid, hipStream_t s)
{
    switch (nwn) {
    case 1: launch_verify_nw<1>(V, grid, s); break;
    case 2: launch_verify_nw<2>(V, grid, s); break;
    case 3: launch_verify_nw<3>(V, grid, s); break;
    case 4: launch_verify_nw<4>(V, grid, s); break;
    case 5: launch_verify_nw<5>(V, grid, s); break;
    case 6: launch_verify_nw<6>(V, grid, s); break;
    case 7: launch_verify_nw<7>(V, grid, s); break;
    case 8: launch_verify_nw<8>(V, grid, s); break;
    case 16: launch_verify_nw<16>(V, grid, s); break;
    case 32: launch_verify_nw<32>(V, grid, s); break;
    default: launch_verify_nw<64>(V, grid, s); break;
    }
}

int ensure_scratch(spm_ctx *ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes)
        return SPM_OK;
    if (ctx->d_scratch) {
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        SPM_HIP_CHECK(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    SPM_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return SPM_OK;
}

// one brute-force pass; `report` = false suppresses hits (state-only pass)
int run_brute(const scan_args &A, uint64_t begin, uint64_t end, uint64_t ctx_begin, const uint32_t *d_state_in,
              uint32_t *d_state_out, bool report, bool single_tile)
{
    spm_ctx *ctx = A.ctx;
    const spm_patterns *ps = A.ps;
    brute_params P{};
    P.text = A.text->d;
    P.text_alloc = A.text->owned ? A.text->alloc : A.text->n;
    P.scan_begin = begin;
    P.scan_end = end;
    P.ctx_begin = ctx_begin;
    P.pos_offset = A.opts.pos_offset;
    P.n_groups = ps->n_groups;
    P.warm = ps->max_window > 0 ? ps->max_window - 1 : 0;
    P.sigma = ps->sigma;
    P.has_state = d_state_in ? 1 : 0;
    P.peq = ps->d_peq;
    P.hp0 = ps->d_hp0;
    P.m = ps->d_m;
    P.k = ps->d_k;
    P.state_in = d_state_in;
    P.state_out = d_state_out;
    P.hits = A.hits->d_hits;
    P.counters = A.hits->d_count + (report ? 0 : 3); // a state-only pass counts into a dummy slot
    P.hit_cap = report ? A.hits->cap : 0;
    if (A.tiles) { // re-scan of the filter's overflowed spans: report only what its verification has not reported
        P.seen = A.d_seen;
        P.seen_mask = A.seen_mask;
        P.overflow = A.hits->d_count + 2;
    }

    const size_t lds_per_wave = (size_t)(ps->sigma + 1) * ps->NW * 64 * sizeof(uint32_t);
    uint32_t wpw = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / lds_per_wave));
    const size_t lds = lds_per_wave * wpw;
    if (lds > 160 * 1024) {
        SPM_SET_ERR(ctx, "needle set needs %zu bytes of LDS per wave (sigma=%u, %u words); limit 160 KiB", lds,
                    ps->sigma, ps->NW);
        return SPM_E_UNSUPPORTED;
    }
    const uint64_t range = end - begin;
    uint32_t grid = ps->n_groups * std::max(1u, (uint32_t)(ctx->n_cu * 8) / (ps->n_groups * 1));
    grid = std::max(grid, ps->n_groups);
    // grid counts workgroups; keep (grid * wpw) % n_groups == 0 so that a wave keeps its needle group in LDS
    const uint64_t n_waves = (uint64_t)grid * wpw;
    uint64_t tile;
    if (single_tile || ps->algo == SPM_ALGO_MYERS_PREFIX) {
        tile = (range + 3) & ~3ull;
    } else {
        const uint64_t min_tile = std::max<uint64_t>(2048, (((uint64_t)P.warm * 16) + 255) & ~255ull);
        tile = (range * ps->n_groups) / (n_waves * 4) + 1;
        tile = (tile + 255) & ~255ull;
        tile = std::max(tile, min_tile);
        tile = std::min<uint64_t>(tile, 1u << 20);
    }
    if (tile == 0)
        tile = 4;
    P.tile = (uint32_t)std::min<uint64_t>(tile, 0xFFFFFF00u);
    P.n_tiles = (uint32_t)std::max<uint64_t>(1, (range + P.tile - 1) / P.tile);
    if (A.tiles) {
        P.n_tiles = (uint32_t)(A.tiles->size() / 3);
        uint64_t *d_tab = nullptr;
        SPM_HIP_CHECK(ctx, hipMalloc(&d_tab, A.tiles->size() * sizeof(uint64_t)));
        hipFree(A.hits->d_aux[0]);
        A.hits->d_aux[0] = d_tab;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_tab, A.tiles->data(), A.tiles->size() * sizeof(uint64_t),
                                          hipMemcpyHostToDevice, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        P.tile_tab = d_tab;
    } else if (!A.seg_offsets && A.d_seg_offsets) {
        // brute-force run over a device-resident segment table (fallback of the journaled-sequence search)
        scan_args &W = const_cast<scan_args &>(A);
        W.seg_host.resize(A.n_segments + 1);
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(W.seg_host.data(), A.d_seg_offsets, (A.n_segments + 1) * sizeof(uint64_t),
                                          hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        W.seg_offsets = W.seg_host.data();
    }
    if (A.seg_offsets && !A.tiles) {
        // every segment is its own haystack: tiles never cross a segment, warm-up stays inside it
        std::vector<uint64_t> tab;
        for (uint64_t s = 0; s < A.n_segments; ++s) {
            const uint64_t sb = A.seg_offsets[s], se = A.seg_offsets[s + 1];
            for (uint64_t lo = sb; lo < se; lo += P.tile) {
                const uint64_t hi = std::min<uint64_t>(lo + P.tile, se);
                tab.push_back(lo >= sb + P.warm ? lo - P.warm : sb);
                tab.push_back(lo);
                tab.push_back(hi);
            }
        }
        if (tab.empty()) { // only empty segments
            tab = {begin, begin, begin};
        }
        if (tab.size() / 3 > 0xFFFFFFFFull) {
            SPM_SET_ERR(ctx, "segmented scan: too many tiles");
            return SPM_E_UNSUPPORTED;
        }
        P.n_tiles = (uint32_t)(tab.size() / 3);
        uint64_t *d_tab = nullptr;
        SPM_HIP_CHECK(ctx, hipMalloc(&d_tab, tab.size() * sizeof(uint64_t)));
        hipFree(A.hits->d_aux[0]);
        A.hits->d_aux[0] = d_tab;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice,
                                          ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // `tab` is a host temporary
        P.tile_tab = d_tab;
    }
    if (single_tile && P.n_tiles != 1) {
        SPM_SET_ERR(ctx, "internal: single-tile pass over %llu symbols", (unsigned long long)range);
        return SPM_E_INVALID;
    }
    const uint64_t n_items = (uint64_t)P.n_tiles * P.n_groups;
    const uint32_t need_wg = (uint32_t)std::min<uint64_t>((n_items + wpw - 1) / wpw, grid);
    // shrinking the grid must keep the group<->wave affinity: round up to a multiple of n_groups when possible
    uint32_t launch_grid = need_wg;
    if (launch_grid < grid) {
        const uint32_t q = (launch_grid + ps->n_groups - 1) / ps->n_groups * ps->n_groups;
        launch_grid = std::min(grid, std::max(q, 1u));
    }
    // Ukkonen cut-off kernel for stateless Myers scans (SPM_HIP_BRUTE_CUTOFF=0 selects the full-width kernel)
    const bool cutoff = ps->algo == SPM_ALGO_MYERS && !d_state_in && !d_state_out && ps->d_peq_bot &&
                        env_int("SPM_HIP_BRUTE_CUTOFF", 1) != 0;
    if (cutoff)
        P.peq = ps->d_peq_bot;
    launch_brute(ps, P, dim3(launch_grid), dim3(64 * wpw), lds, ctx->stream, cutoff);
    SPM_HIP_CHECK(ctx, hipGetLastError());
    A.hits->stats.main_launches++;
    return SPM_OK;
}

// persistent band table of the context: empty between scans (the verification gives every slot back), so a scan pays
// for the bands it has, not for a memset of the table
int ensure_band_table(spm_ctx *ctx, uint64_t slots)
{
    if (ctx->band_slots >= slots && !ctx->band_dirty)
        return SPM_OK;
    if (ctx->band_slots < slots) {
        if (ctx->d_band_tab) {
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->d_band_tab);
            ctx->d_band_tab = nullptr;
            ctx->band_slots = 0;
        }
        SPM_HIP_CHECK(ctx, hipMalloc(&ctx->d_band_tab, slots * sizeof(ulonglong2)));
        ctx->band_slots = slots;
    }
    SPM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_band_tab, 0xFF, ctx->band_slots * sizeof(ulonglong2), ctx->stream)); // all free
    ctx->band_dirty = false;
    return SPM_OK;
}

int run_filter(const scan_args &A)
{
    spm_ctx *ctx = A.ctx;
    const spm_patterns *ps = A.ps;
    spm_hits *H = A.hits;
    const uint64_t kmax = ps->max_k;
    // ---- sizes: survivor list, band list, band table, dedupe set ----
    constexpr uint64_t kSurvMax = 1ull << 27; // 2 GiB of survivors: beyond that spans give up (brute-force re-scan)
    uint64_t est = std::max<uint64_t>(4096, 8ull * ps->n * (kmax + 1)); // a handful of true seed hits per needle
    // real texts are not uniform: room for the seed hits of repeat stretches (one survivor per 512 symbols)
    est = std::max<uint64_t>(est, (A.end - A.begin) / 512);
    // chance hits of short keys: windows looked at x keys / 4^key_len, per pass (negligible for 16-symbol keys)
    double chance = 0;
    for (const filter_index &F : ps->fidx)
        chance += (double)(A.end - A.begin) / std::max(1u, F.stride) * (double)F.n_keys / std::pow(4.0, (double)F.key_len);
    est = std::max<uint64_t>(est, (uint64_t)(2.0 * chance));
    est = std::max<uint64_t>(est, ps->cand_hint + ps->cand_hint / 4);
    // slots are drawn in growing chunks per wave (unused tails stay invalid): twice the estimate + the first chunks
    uint64_t surv_cap = std::min(2 * est + (uint64_t)ctx->n_cu * 16 * kChunkMin, kSurvMax);
    if (A.cand_cap_override)
        surv_cap = A.cand_cap_override;
    const int cc = env_int("SPM_HIP_FILTER_CAND_CAP", 0);
    if (cc > 0)
        surv_cap = (uint64_t)cc;
    // (+ the first chunk of every wave of resolve_kernel: n_cu x 8 workgroups of 4 waves)
    const uint64_t band_cap = std::max<uint64_t>(surv_cap, 4096) * (A.band_scale ? A.band_scale : 1) +
                              (uint64_t)ctx->n_cu * 32 * kChunkMin;
    uint64_t band_slots = 1u << 12;
    while (band_slots < 2 * band_cap)
        band_slots <<= 1;
    // bands: Bw diagonals each.  Sets with surplus seeds: (k+1) x factor, overlapping by k + 1 (wider bands mean fewer
    // occurrences whose seeds straddle two of them at the price of more end positions per verification; 4(k+1) measured
    // best for |P| = 1024, k = 64; the lane-per-band kernel keeps its end-position slots per thread: narrow bands there).
    // Other sets: 64 diagonals, no overlap -- every band with a seed hit is verified.
    uint32_t nwn = std::max(1u, (ps->max_m + 31) / 32);
    const int wave_min = env_int("SPM_HIP_VERIFY_WAVE_MIN_WORDS", 8); // 0 = never use the wave-per-band kernel
    // (the wave-per-band kernel keeps the match masks of <= 5 symbols in registers: dna15 sets use the lane-per-band one)
    const bool use_wave = ps->d_peq_bot && wave_min > 0 && nwn >= (uint32_t)wave_min && ps->sigma <= 5;
    const bool overlap = ps->d_surplus != nullptr;
    uint32_t Bw;
    if (overlap) {
        const uint32_t bw_factor = use_wave ? (uint32_t)std::max(1, env_int("SPM_HIP_FILTER_BAND_FACTOR", 4)) : 1u;
        Bw = (ps->max_k + 1) * bw_factor;
        if (Bw + ps->max_k > 2047)
            Bw = ps->max_k + 1;
    } else {
        Bw = (uint32_t)std::max(8, std::min(64, env_int("SPM_HIP_FILTER_BAND", 32))); // (one mask bit per diagonal)
    }
    const uint32_t max_span = Bw - 1 + (overlap ? ps->max_k + 1 : 0);
    // dedupe set: one key per reported hit, so twice the hit capacity is room enough; a caller with a huge hit buffer
    // (repeat-rich texts) pays for what earlier scans of this needle set actually reported
    uint64_t want_seen = std::min<uint64_t>(band_cap * (2 * kmax + 1 + max_span), std::max<uint64_t>(H->cap, 1));
    if (!A.seen_full)
        // (the first scan of a needle set knows nothing yet: room for 4 M hits -- a 64 MiB memset, 10 us -- rather than a
        // set that a repeat-rich text fills up, which costs a second run of the whole scan)
        want_seen = std::min<uint64_t>(want_seen, ps->scanned ? std::max<uint64_t>(1u << 18, 4 * ps->hit_hint) : (1ull << 22));
    uint64_t seen_slots = 1u << 16;
    while (seen_slots < 2 * want_seen)
        seen_slots <<= 1;
    const size_t surv_bytes = surv_cap * sizeof(survivor);
    const size_t seen_bytes = seen_slots * sizeof(unsigned long long);
    const size_t band_bytes = band_cap * sizeof(band_rec) * (overlap ? 2 : 1); // (+ the selected bands of overlapping sets)
    const size_t ovf_bytes = kOvfCap * 2 * sizeof(uint64_t);
    int rc = ensure_scratch(ctx, surv_bytes + seen_bytes + band_bytes + ovf_bytes);
    if (rc != SPM_OK)
        return rc;
    // Exact sets whose needles are their own single seed (k = 0, no `N`, e.g. Shift-Or / Horspool sets): the whole-seed check
    // of the resolve kernel is the whole comparison, so it reports the hits itself -- no band table, no verification launch.
    // (Not for needles that are repeats: their merged index entries skip the per-offset check.)
    bool exact_hits = ps->max_k == 0 && !overlap && ps->filter_max_range == 0 && ps->d_ranks &&
                      env_int("SPM_HIP_VERIFY_SEED_CHECK", 1) != 0 && env_int("SPM_HIP_EXACT_FROM_RESOLVE", 1) != 0;
    if (exact_hits && ps->exact_whole < 0) {
        bool whole = true;
        for (uint32_t p = 0; p < ps->n && whole; ++p)
            whole = ps->seed_n[p] == 1 && ps->seed_q[p] == (uint32_t)ps->m[p];
        ps->exact_whole = whole ? 1 : 0;
    }
    exact_hits = exact_hits && ps->exact_whole == 1;
    if (!exact_hits) {
        rc = ensure_band_table(ctx, band_slots);
        if (rc != SPM_OK)
            return rc;
    }
    survivor *d_surv = (survivor *)ctx->d_scratch;
    unsigned long long *d_seen = (unsigned long long *)((uint8_t *)ctx->d_scratch + surv_bytes);
    band_rec *d_bands = (band_rec *)((uint8_t *)d_seen + seen_bytes);
    uint64_t *d_ovf = (uint64_t *)((uint8_t *)d_bands + band_bytes);
    {
        scan_args &W = const_cast<scan_args &>(A);
        W.d_seen = d_seen;
        W.seen_mask = (uint32_t)(seen_slots - 1);
        W.d_ovf = d_ovf;
    }
    SPM_HIP_CHECK(ctx, hipMemsetAsync(d_seen, 0xFF, seen_bytes, ctx->stream));
    if (!exact_hits)
        ctx->band_dirty = true; // until the verification has consumed every band of this scan

    filter_params P{};
    P.text = A.text->d;
    P.text_alloc = A.text->owned ? A.text->alloc : A.text->n;
    // windows that can belong to an occurrence whose last symbol is owned
    const uint64_t reach = ps->max_window;
    P.lo = A.begin >= A.ctx_begin + reach ? A.begin - reach : A.ctx_begin;
    P.hi = A.end;
    P.surv = d_surv;
    P.counters = H->d_count;
    P.surv_cap = surv_cap;
    P.ovf_spans = d_ovf;
    P.ovf_cap = kOvfCap;
    SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[1], ctx->stream));
    for (size_t fi = 0; fi < ps->fidx.size(); ++fi) {
    const filter_index &F = ps->fidx[fi];
    P.stride = F.stride;
    P.key_len = F.key_len;
    P.key_mask = F.key_len >= 16 ? 0xFFFFFFFFu : ((1u << (2 * F.key_len)) - 1);
    P.bitmap_words = F.hash_variant == 2 ? 1024 : F.bitmap_words;
    P.lds_words = F.lds_words;
    P.chd_slot_mask = F.chd_slot_mask;
    P.chd_bucket_shift = F.chd_bucket_shift;
    P.chd_disp_off = F.chd_disp_off;
    P.n_probes = F.n_probes;
    P.bitmap = F.d_bitmap;
    P.pass = (uint32_t)fi;
    P.anchor_c = F.anchor_c;
    P.anchor_cm = F.anchor_cm;
    P.n_pat = F.n_pat;
    for (uint32_t i = 0; i < kDensePatterns; ++i) {
        P.pat_c[i] = F.pat_c[i];
        P.pat_cm[i] = F.pat_cm[i];
    }
    P.bucket_shift = F.bucket_shift;
    P.dense_debug = (uint32_t)env_int("SPM_HIP_DENSE_DEBUG", 0);
    P.buckets = reinterpret_cast<const uint4 *>(F.d_buckets);
    if (fi > 0) // each pass draws its spans from a fresh head
        SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count + 4, 0, sizeof(unsigned long long), ctx->stream));
    const bool use_packed = !F.dense && A.text->d_packed && ps->sigma == 4 && F.hash_variant == 2 && F.stride >= 2 &&
                            !(A.opts.flags & SPM_SCAN_IGNORE_PACKED);
    // measured best: 8 waves per CU on the 1-byte text when HBM binds, 16 on the 2-bit shadow and at stride 1 with
    // 16-symbol keys (LDS-bound: C4 25.8 vs 29.1 ms; the other stride-1/2 variants need more than 128 VGPRs)
    // stride 2: two chunks per group (16 windows per lane) need < 128 VGPRs, so 16 waves per CU hide the LDS round trips
    // (C5: 0.53 -> 0.46 ms; four chunks per group hold 167 VGPRs at 8 waves)
    const bool narrow2 = F.stride == 2 && !use_packed && env_int("SPM_HIP_FILTER_S2_U", 2) == 2;
    const bool wide_ok = use_packed || narrow2 || (F.stride == 1 && F.key_len >= 16 && ps->sigma == 4 &&
                                                   !env_int("SPM_HIP_FILTER_FORCE_MASKED", 0));
    const uint32_t threads = F.dense ? 1024u : (uint32_t)std::max(
        64, std::min(wide_ok ? 1024 : 512, env_int("SPM_HIP_FILTER_THREADS", wide_ok ? 1024 : 512)));
    // + the workgroup's span-dequeue slot (4 words) + one survivor-chunk record per wave (+ dense: one queue per wave)
    const size_t lds = (size_t)F.lds_words * 4 + 16 + 16 * kCandRec * 4 + (F.dense ? 16 * sizeof(dense_queue) : 0);
    const uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 2048 / threads));
    const uint32_t grid = ctx->n_cu * wg_per_cu;
    const uint64_t n_waves = (uint64_t)grid * (threads / 64);
    const uint64_t n_chunks = (P.hi - (P.lo & ~1023ull) + 1023) / 1024;
    uint64_t span = n_chunks / (n_waves * (uint64_t)std::max(1, env_int("SPM_HIP_FILTER_SPANS_PER_WAVE", 32))) + 1;
    // small texts: at least 64 KiB per dequeue as long as every wave still gets ~4 spans (a 1 GiB text ran 14 % faster
    // with 64-chunk spans than with the 24 the rule above gives: fewer dequeue rounds, each a workgroup barrier)
    if (span < 64)
        span = std::max<uint64_t>(span, std::min<uint64_t>(64, n_chunks / (n_waves * 4) + 1));
    span = std::min<uint64_t>(std::max<uint64_t>(span, 8), 4096);
    const int fs = env_int("SPM_HIP_FILTER_SPAN", 0);
    if (fs > 0)
        span = (uint64_t)fs;
    span = (span + 7) & ~7ull; // whole groups of chunks
    P.span_chunks = (uint32_t)span;
    P.span_unit = 1024;
    // candidates a span may produce before it gives up and is re-scanned by the brute-force kernel: one per 4 symbols
    // costs the verification about what the re-scan would
    {
        const int sb = env_int("SPM_HIP_FILTER_SPAN_BUDGET", 0);
        P.span_budget = sb > 0 ? (uint32_t)sb : (uint32_t)std::max<uint64_t>(256, span * 1024 / 4);
    }
    // span dequeue: per wave while the dequeue rate stays far below what one atomic word sustains (~88/us, i.e.
    // spans >= 192 KiB at 7 TB/s), per workgroup otherwise (measured: C3 2.52 vs 2.59 ms, C2 0.88 vs 0.20 ms)
    const int dyn = env_int("SPM_HIP_FILTER_DYN", -1);
    P.dynamic = dyn >= 0 ? (uint32_t)std::min(2, dyn) : (span >= 192 ? 1u : 2u);

    const int U = env_int("SPM_HIP_FILTER_U", 8) >= 8 ? 8 : 4;
    const bool NT = env_int("SPM_HIP_FILTER_NT", 1) != 0;
    P.hash_variant = F.hash_variant;
    const bool short_keys = F.key_len < 16 || env_int("SPM_HIP_FILTER_FORCE_MASKED", 0) != 0; // (the env: diagnostics)
#define LAUNCH_FILTER4(S, UU, NTT, HV, SG, KM)                                                                         \
    do {                                                                                                               \
        hipFuncSetAttribute((const void *)seed_filter_kernel<S, UU, NTT, HV, SG, KM>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
        hipLaunchKernelGGL((seed_filter_kernel<S, UU, NTT, HV, SG, KM>), dim3(grid), dim3(threads), lds, ctx->stream,  \
                           P);                                                                                         \
    } while (0)
    // keys shorter than 16 symbols only occur with strides 1 and 2: the masked variants exist for those alone
#define LAUNCH_FILTER3(S, UU, NTT, HV)                                                                                 \
    do {                                                                                                               \
        constexpr bool km = (S) <= 2;                                                                                  \
        if (ps->sigma == 5) {                                                                                          \
            if (short_keys)                                                                                            \
                LAUNCH_FILTER4(S, UU, true, 2, 5, km);                                                                 \
            else                                                                                                       \
                LAUNCH_FILTER4(S, UU, true, 2, 5, false);                                                              \
        } else if (ps->sigma == 15) {                                                                                  \
            if (short_keys)                                                                                            \
                LAUNCH_FILTER4(S, UU, true, 2, 15, km);                                                                \
            else                                                                                                       \
                LAUNCH_FILTER4(S, UU, true, 2, 15, false);                                                             \
        } else {                                                                                                       \
            if (short_keys)                                                                                            \
                LAUNCH_FILTER4(S, UU, NTT, HV, 4, km);                                                                 \
            else                                                                                                       \
                LAUNCH_FILTER4(S, UU, NTT, HV, 4, false);                                                              \
        }                                                                                                              \
    } while (0)
#define LAUNCH_FILTER2(S, UU)                                                                                          \
    do {                                                                                                               \
        if (NT) {                                                                                                      \
            if (F.hash_variant == 2)                                                                                   \
                LAUNCH_FILTER3(S, UU, true, 2);                                                                        \
            else if (F.hash_variant == 1)                                                                              \
                LAUNCH_FILTER3(S, UU, true, 1);                                                                        \
            else                                                                                                       \
                LAUNCH_FILTER3(S, UU, true, 0);                                                                        \
        } else {                                                                                                       \
            if (F.hash_variant == 2)                                                                                   \
                LAUNCH_FILTER3(S, UU, false, 2);                                                                       \
            else if (F.hash_variant == 1)                                                                              \
                LAUNCH_FILTER3(S, UU, false, 1);                                                                       \
            else                                                                                                       \
                LAUNCH_FILTER3(S, UU, false, 0);                                                                       \
        }                                                                                                              \
    } while (0)
#define LAUNCH_FILTER(S, UMAX)                                                                                         \
    do {                                                                                                               \
        if (U >= 8 && UMAX >= 8)                                                                                       \
            LAUNCH_FILTER2(S, (UMAX >= 8 ? 8 : UMAX));                                                                 \
        else                                                                                                           \
            LAUNCH_FILTER2(S, (UMAX >= 4 ? 4 : UMAX));                                                                 \
    } while (0)
    if (F.dense) {
#define LAUNCH_DENSE(NP)                                                                                               \
    do {                                                                                                               \
        hipFuncSetAttribute((const void *)seed_filter_dense_kernel<4, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                                 \
        hipLaunchKernelGGL((seed_filter_dense_kernel<4, NP>), dim3(grid), dim3(threads), lds, ctx->stream, P);         \
    } while (0)
        if (F.n_pat <= 1)
            LAUNCH_DENSE(1);
        else if (F.n_pat == 2)
            LAUNCH_DENSE(2);
        else
            LAUNCH_DENSE(3);
#undef LAUNCH_DENSE
    } else if (use_packed) {
        // p-chunks of 4096 symbols: recompute the span geometry in those units
        filter_params Q = P;
        const uint64_t n_pchunks = (Q.hi - (Q.lo & ~4095ull) + 4095) / 4096;
        uint64_t pspan = n_pchunks / (n_waves * (uint64_t)std::max(1, env_int("SPM_HIP_FILTER_SPANS_PER_WAVE", 8))) + 1;
        pspan = std::min<uint64_t>(std::max<uint64_t>(pspan, 4), 4096);
        pspan = (pspan + 3) & ~3ull;
        Q.span_chunks = (uint32_t)pspan;
        Q.span_unit = 4096;
        if (env_int("SPM_HIP_FILTER_SPAN_BUDGET", 0) <= 0)
            Q.span_budget = (uint32_t)std::max<uint64_t>(256, pspan * 4096 / 4);
        Q.dynamic = dyn >= 0 ? (uint32_t)std::min(2, dyn) : (pspan >= 48 ? 1u : 2u);
        const uint4 *shadow = reinterpret_cast<const uint4 *>(A.text->d_packed);
#define LAUNCH_PACKED2(S, U2, KM)                                                                                      \
    do {                                                                                                               \
        hipFuncSetAttribute((const void *)seed_filter_packed_kernel<S, U2, 2, KM>,                                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
        hipLaunchKernelGGL((seed_filter_packed_kernel<S, U2, 2, KM>), dim3(grid), dim3(threads), lds, ctx->stream, Q,  \
                           shadow);                                                                                    \
    } while (0)
#define LAUNCH_PACKED(S, U2)                                                                                           \
    do {                                                                                                               \
        constexpr bool km = (S) <= 2;                                                                                  \
        if (short_keys)                                                                                                \
            LAUNCH_PACKED2(S, U2, km);                                                                                 \
        else                                                                                                           \
            LAUNCH_PACKED2(S, U2, false);                                                                              \
    } while (0)
        switch (F.stride) {
        case 16: LAUNCH_PACKED(16, 4); break;
        case 8: LAUNCH_PACKED(8, 4); break;
        case 4: LAUNCH_PACKED(4, 2); break;
        case 2: LAUNCH_PACKED(2, 1); break;
        default:
            // stride 1 has 16 windows per word: 4 words already fill the 32-bit survivor mask twice over
            SPM_SET_ERR(ctx, "internal: packed filter with stride 1");
            return SPM_E_UNSUPPORTED;
        }
#undef LAUNCH_PACKED
#undef LAUNCH_PACKED2
    } else
    switch (F.stride) {
    case 16: LAUNCH_FILTER(16, 8); break;
    case 8: LAUNCH_FILTER(8, 8); break;
    case 4: LAUNCH_FILTER(4, 8); break;
    case 2:
        if (narrow2)
            LAUNCH_FILTER2(2, 2);
        else
            LAUNCH_FILTER(2, 4);
        break;
    default:
        if (F.anchor_cm != 0 && F.hash_variant == 2 && ps->sigma == 4 && !short_keys) {
            // anchored pass: few windows per lane are looked up, so a lane can hold more text
            const int au = env_int("SPM_HIP_FILTER_ANCHOR_U", 4);
#define LAUNCH_ANCHORED(UU)                                                                                            \
    do {                                                                                                               \
        hipFuncSetAttribute((const void *)seed_filter_kernel<1, UU, true, 2, 4, false, true>,                          \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
        hipLaunchKernelGGL((seed_filter_kernel<1, UU, true, 2, 4, false, true>), dim3(grid), dim3(threads), lds,       \
                           ctx->stream, P);                                                                            \
    } while (0)
            if (au >= 8)
                LAUNCH_ANCHORED(8);
            else if (au >= 4)
                LAUNCH_ANCHORED(4);
            else
                LAUNCH_ANCHORED(2);
#undef LAUNCH_ANCHORED
        } else {
            LAUNCH_FILTER(1, 2);
        }
        break;
    }
#undef LAUNCH_FILTER2
#undef LAUNCH_FILTER3
#undef LAUNCH_FILTER4
#undef LAUNCH_FILTER
    SPM_HIP_CHECK(ctx, hipGetLastError());
    H->stats.main_launches++;
    }
    SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[2], ctx->stream));

    // ---- resolve: survivors -> needles -> whole-seed check -> diagonal bands (one launch for all passes) ----
    const uint64_t *d_seg = nullptr;
    uint64_t n_seg = 0;
    if (A.d_seg_offsets) {
        d_seg = A.d_seg_offsets;
        n_seg = A.n_segments;
    } else if (A.seg_offsets) {
        uint64_t *d = nullptr;
        SPM_HIP_CHECK(ctx, hipMalloc(&d, (A.n_segments + 1) * sizeof(uint64_t)));
        hipFree(H->d_aux[1]);
        H->d_aux[1] = d;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d, A.seg_offsets, (A.n_segments + 1) * sizeof(uint64_t),
                                          hipMemcpyHostToDevice, ctx->stream));
        d_seg = d;
        n_seg = A.n_segments;
    }
    uint32_t seg_bits = 0;
    while (n_seg && (1ull << seg_bits) < n_seg + 1)
        ++seg_bits;
    resolve_params R{};
    R.surv = d_surv;
    R.counters = H->d_count;
    R.surv_cap = surv_cap;
    R.passes = ps->d_pass_tab;
    R.entries = ps->d_entries;
    R.key_len = ps->filter_key_len;
    R.text = A.text->d;
    R.text_alloc = A.text->owned ? A.text->alloc : A.text->n;
    R.needle_ranks = env_int("SPM_HIP_VERIFY_SEED_CHECK", 1) ? ps->d_ranks : nullptr;
    R.needle_offsets = ps->d_offsets;
    R.seed_q = ps->d_seed_q;
    R.flank_check = (ps->sigma == 4 && !overlap && R.needle_ranks && env_int("SPM_HIP_FLANK_CHECK", 1)) ? 1u : 0u;
    R.pieces_check = env_int("SPM_HIP_PIECES_CHECK", 1) ? R.flank_check : 0u;
    R.m = ps->d_m;
    R.k = ps->d_k;
    R.hay_begin = A.ctx_begin;
    R.hay_end = A.end;
    R.seg_offsets = d_seg;
    R.n_segments = n_seg;
    R.Bw = Bw;
    R.overlap = overlap ? 1u : 0u;
    R.max_m = ps->max_m;
    R.band_bits = 43 - seg_bits;
    R.band_tab = ctx->d_band_tab;
    R.needle_pk = ps->d_needle_pk;
    R.pk_offsets = ps->d_pk_offsets;
    R.exact_hits = exact_hits ? 1u : 0u;
    R.report_begin = ps->is_myers() ? 0 : 1;
    R.scan_begin = A.begin;
    R.scan_end = A.end;
    R.pos_offset = A.opts.pos_offset;
    R.seg_owned = A.d_seg_owned;
    R.seen = d_seen;
    R.seen_mask = (uint32_t)(seen_slots - 1);
    R.hits = H->d_hits;
    R.hit_counter = H->d_count;
    R.overflow = H->d_count + 2;
    R.hit_cap = H->cap;
    R.table_mask = (uint32_t)(band_slots - 1);
    R.bands = d_bands;
    R.band_cap = band_cap;
    // grid: what earlier scans of this needle set produced (a full grid of idle workgroups costs ~30 us on a 2.6 ms scan);
    // a scan that produces more simply loops
    const uint64_t surv_expect = ps->cand_hint ? 2 * ps->cand_hint : surv_cap;
    // (5 workgroups per CU are resident at once -- LDS queues, 84 VGPRs --: a larger grid only adds a second, partly filled
    // round.  A lane takes ~4 survivors in turn: measured on C5, whose survivors are few and cheap, 0.137 -> 0.10 ms;
    // c3r 1.33 -> 1.25 ms with the cap alone.)
    const uint64_t rmax = (uint64_t)ctx->n_cu * (uint64_t)std::max(1, env_int("SPM_HIP_RESOLVE_WGS_PER_CU", 5));
    // (a short survivor list: one survivor per lane, its latency is the kernel's; a long one: four per lane)
    const uint64_t per_wg = (surv_expect + 255) / 256 <= rmax ? 256 : (uint64_t)std::max(256, env_int("SPM_HIP_RESOLVE_SURV_PER_WG", 1024));
    const uint32_t rgrid = (uint32_t)std::min<uint64_t>(rmax, std::max<uint64_t>(ctx->n_cu / 2, (surv_expect + per_wg - 1) / per_wg));
    hipLaunchKernelGGL(resolve_kernel, dim3(rgrid), dim3(256), 0, ctx->stream, R);
    SPM_HIP_CHECK(ctx, hipGetLastError());

    // ---- verification: one band = one verification ----
    verify_params V{};
    V.text = A.text->d;
    V.text_alloc = A.text->owned ? A.text->alloc : A.text->n;
    V.ctx_begin = A.ctx_begin;
    V.scan_begin = A.begin;
    V.scan_end = A.end;
    V.pos_offset = A.opts.pos_offset;
    V.bands = d_bands;
    V.counters = H->d_count;
    V.band_cap = band_cap;
    V.band_tab = ctx->d_band_tab;
    V.surplus = ps->d_surplus;
    V.Bw = Bw;
    V.overlap = overlap ? 1u : 0u;
    V.max_m = ps->max_m;
    V.peq32 = ps->d_peq_verify ? ps->d_peq_verify : ps->d_peq;
    V.sigma = ps->sigma;
    V.nw_table = ps->NW;
    V.max_k = ps->max_k;
    V.max_span = max_span;
    V.m = ps->d_m;
    V.k = ps->d_k;
    V.report_begin = ps->is_myers() ? 0 : 1;
    V.seen = d_seen;
    V.seen_mask = (uint32_t)(seen_slots - 1);
    V.hits = H->d_hits;
    V.hit_counter = H->d_count;
    V.hit_cap = H->cap;
    V.overflow = H->d_count + 2;
    V.seg_offsets = d_seg;
    V.n_segments = n_seg;
    V.seg_owned = A.d_seg_owned;
    V.band_counter = 3;
    if (overlap) {
        hipLaunchKernelGGL(band_select_kernel, dim3(ctx->n_cu * 2), dim3(256), 0, ctx->stream, V, d_bands + band_cap,
                           H->d_count + 10);
        SPM_HIP_CHECK(ctx, hipGetLastError());
        V.bands = d_bands + band_cap;
        V.band_counter = 10;
        V.preselected = 1;
    }
    if (exact_hits) {
        // (the resolve kernel reported the hits)
    } else if (use_wave) {
        // one verification is a ~1400-step serial chain: enough waves that every band gets its own right away
        launch_verify_wave(nwn, V, ps->d_peq_bot, ps->max_m, dim3(ctx->n_cu * 16), ctx->stream);
    } else {
        if (nwn > 8)
            nwn = ps->NW; // power of two beyond 8 words
        const uint64_t band_expect = ps->band_hint ? 2 * ps->band_hint : band_cap;
        const uint32_t vgrid = (uint32_t)std::min<uint64_t>((uint64_t)ctx->n_cu * 4, std::max<uint64_t>(ctx->n_cu / 2, (band_expect + 255) / 256));
        launch_verify(nwn, V, dim3(vgrid), ctx->stream);
    }
    SPM_HIP_CHECK(ctx, hipGetLastError());
    SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream));
    H->cand_cap = surv_cap;
    H->band_cap = band_cap;
    return SPM_OK;
}

} // namespace

// after_launch: work the caller wants on the stream right behind the filter scan's kernels, before the host reads the
// counters (the pan-genome search's fan-out); spm_hits::hook_final tells whether it saw the final hit list.
static int scan_impl(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end, const spm_patterns *patterns,
                     const spm_scan_opts *opts_in, const void *state_in, void *state_out, const uint64_t *seg_offsets,
                     uint64_t n_segments, spm_hits **out, const uint64_t *d_seg_offsets = nullptr,
                     const uint32_t *d_seg_owned = nullptr, const std::function<int(spm_hits *)> *after_launch = nullptr);

extern "C" int spm_hip_scan(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end,
                            const spm_patterns *patterns, const spm_scan_opts *opts_in, const void *state_in,
                            void *state_out, spm_hits **out)
{
    return scan_impl(ctx, text, begin, end, patterns, opts_in, state_in, state_out, nullptr, 0, out);
}

extern "C" int spm_hip_scan_segments(spm_ctx *ctx, const spm_text *text, const uint64_t *seg_offsets,
                                     uint64_t n_segments, const spm_patterns *patterns, const spm_scan_opts *opts_in,
                                     spm_hits **out)
{
    if (!ctx || !text || !seg_offsets || n_segments == 0) {
        SPM_SET_ERR(ctx, "spm_hip_scan_segments: invalid argument");
        return SPM_E_INVALID;
    }
    for (uint64_t s = 0; s < n_segments; ++s)
        if (seg_offsets[s + 1] < seg_offsets[s] || seg_offsets[s + 1] > text->n) {
            SPM_SET_ERR(ctx, "spm_hip_scan_segments: offsets must ascend and stay inside the text");
            return SPM_E_INVALID;
        }
    spm_scan_opts o{};
    if (opts_in)
        o = *opts_in;
    o.left_context = 0;
    return scan_impl(ctx, text, seg_offsets[0], seg_offsets[n_segments], patterns, &o, nullptr, nullptr, seg_offsets,
                     n_segments, out);
}

static int scan_impl(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end, const spm_patterns *patterns,
                     const spm_scan_opts *opts_in, const void *state_in, void *state_out, const uint64_t *seg_offsets,
                     uint64_t n_segments, spm_hits **out, const uint64_t *d_seg_offsets, const uint32_t *d_seg_owned,
                     const std::function<int(spm_hits *)> *after_launch)
{
    if (!ctx || !text || !patterns || !out || begin > end || end > text->n) {
        SPM_SET_ERR(ctx, "spm_hip_scan: invalid argument");
        return SPM_E_INVALID;
    }
    if (patterns->sigma != text->sigma) {
        SPM_SET_ERR(ctx, "spm_hip_scan: text sigma %u != pattern sigma %u", text->sigma, patterns->sigma);
        return SPM_E_INVALID;
    }
    spm_scan_opts opts{};
    if (opts_in)
        opts = *opts_in;
    const auto t_call = clk::now();
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));

    std::unique_ptr<spm_hits, void (*)(spm_hits *)> H(new spm_hits, spm_hip_hits_destroy);
    H->ctx = ctx;
    H->cap = opts.max_hits ? opts.max_hits : (1ull << 20);
    {
        // recycle the buffers of an earlier scan (hipMalloc/hipEventCreate per scan cost ~0.2 ms)
        bool reused = false;
        for (size_t i = 0; i < ctx->pool.size(); ++i)
            if (ctx->pool[i].cap == H->cap) {
                const hits_block b = ctx->pool[i];
                ctx->pool.erase(ctx->pool.begin() + i);
                H->d_hits = b.d_hits;
                H->d_count = b.d_count;
                for (int e = 0; e < 4; ++e)
                    H->ev[e] = b.ev[e];
                reused = true;
                break;
            }
        if (!reused) {
            SPM_HIP_CHECK(ctx, hipMalloc(&H->d_hits, std::max<uint64_t>(H->cap, 1) * sizeof(spm_hit)));
            SPM_HIP_CHECK(ctx, hipMalloc(&H->d_count, 16 * sizeof(unsigned long long)));
            for (int i = 0; i < 4; ++i)
                SPM_HIP_CHECK(ctx, hipEventCreate(&H->ev[i]));
        }
    }
    SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));

    scan_args A{ctx, text, begin, end, opts.left_context ? 0 : begin, patterns, opts, state_in, state_out, H.get()};
    A.seg_offsets = seg_offsets;
    A.n_segments = n_segments;
    A.d_seg_offsets = d_seg_offsets;
    A.d_seg_owned = d_seg_owned;

    const bool has_state = state_in != nullptr;
    const bool want_filter = opts.engine == SPM_ENGINE_FILTER || (opts.engine == SPM_ENGINE_AUTO && !patterns->fidx.empty());
    if (opts.engine == SPM_ENGINE_FILTER && patterns->fidx.empty()) {
        SPM_SET_ERR(ctx, "spm_hip_scan: the seed filter does not apply to this needle set");
        return SPM_E_UNSUPPORTED;
    }
    // Restorable scans (myers_matcher_restorable.hpp:72-82: the chunk continues from the restored state).  Only the
    // first window_size - 1 symbols of a chunk can complete an occurrence that began before it: those are scanned by the
    // brute-force kernel from the state; from there on every occurrence lies inside the chunk, so the seed filter takes
    // the rest with the chunk as its haystack.  The state after the last symbol comes from the last 2 max|P| symbols.
    // Short chunks stay with the brute-force kernel (unless the caller asks for the filter).
    const bool stateful = has_state || state_out != nullptr;
    const uint64_t state_prefix = has_state && patterns->max_window > 0 ? patterns->max_window - 1 : 0;
    bool use_filter = want_filter && patterns->n > 0 && end > begin;
    if (stateful && use_filter &&
        (seg_offsets || d_seg_offsets || end - begin <= state_prefix ||
         (opts.engine != SPM_ENGINE_FILTER && end - begin < (1u << 18))))
        use_filter = false;
    if (opts.engine == SPM_ENGINE_FILTER && !use_filter && patterns->n > 0 && end > begin) {
        SPM_SET_ERR(ctx, "spm_hip_scan: the seed filter does not apply to this stateful scan (chunk shorter than a window)");
        return SPM_E_UNSUPPORTED;
    }
    if (use_filter && has_state) {
        // the filter's part of a chunk: hits whose last symbol lies at or behind begin + window - 1, haystack = the chunk
        A.begin = begin + state_prefix;
        A.ctx_begin = begin;
    }

    SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[0], ctx->stream));
    H->timed = true;
    if (patterns->n == 0 || end == begin) {
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[1], ctx->stream));
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[2], ctx->stream));
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream));
        H->stats.engine_used = SPM_ENGINE_BRUTE;
        if (state_in && state_out && state_out != state_in)
            memcpy(state_out, state_in, spm_hip_patterns_state_stride(patterns) * patterns->n);
        *out = H.release();
        return SPM_OK;
    }

    if (use_filter) {
        H->stats.engine_used = SPM_ENGINE_FILTER;
        // [0] hits, [1] survivor slots drawn, [2] hard overflow (band list / band table / dedupe set / overflow list),
        // [3] band slots drawn, [5] candidates, [6] spans that gave up, [7] bands verified
        unsigned long long *c = ctx->h_counters;
        const bool segmented = seg_offsets || d_seg_offsets;
        int rc = SPM_OK;
        for (int outer = 0; outer < 2; ++outer) { // (second round: the dedupe set was too small for the re-scan's hits)
        bool again = false;
        for (int attempt = 0;; ++attempt) {
            rc = run_filter(A);
            if (rc != SPM_OK)
                return rc;
            if (after_launch && !stateful) {
                rc = (*after_launch)(H.get());
                if (rc != SPM_OK)
                    return rc;
            }
            // the overflow checks need the counters: one small D2H copy
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(c, H->d_count, 13 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (c[2] == 0)
                ctx->band_dirty = false; // every band was consumed: the table is empty again
            if (c[0] > H->cap || attempt == 2)
                break;
            // Start over with more room when the lists were too small for this text (the first attempt counted the
            // demand): survivor buffer full -- spans gave up for that reason, not for their own budget --, or band list /
            // band table / dedupe set full.
            const bool more_surv = c[1] > H->cand_cap && H->cand_cap < (1ull << 27) && !env_int("SPM_HIP_FILTER_CAND_CAP", 0);
            const bool more_bands = c[2] != 0 && c[3] > H->band_cap && H->band_cap < (1ull << 28);
            const bool more_seen = c[2] != 0 && !A.seen_full && !more_bands && c[3] <= H->band_cap;
            if (!more_surv && !more_bands && !more_seen)
                break;
            if (more_surv)
                A.cand_cap_override = std::min<uint64_t>(1ull << 27, std::max<uint64_t>(c[1] + c[1] / 8 + 4096, 4 * H->cand_cap));
            if (more_bands)
                A.band_scale = std::max<uint64_t>(1, A.band_scale) * std::max<uint64_t>(2, (c[3] + H->band_cap - 1) / H->band_cap + 1);
            if (more_seen)
                A.seen_full = true;
            SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));
            H->stats.main_launches = 0;
        }
        H->stats.n_candidates = c[5];
        H->stats.n_bands = (uint32_t)std::min<unsigned long long>(c[7], 0xFFFFFFFFull);
        if (c[2] == 0 && c[6] == 0 && c[1] <= H->cand_cap) {
            patterns->cand_hint = std::max<uint64_t>(patterns->cand_hint, c[1]);
            patterns->hit_hint = std::max<uint64_t>(patterns->hit_hint, c[0]);
            patterns->band_hint = std::max<uint64_t>(patterns->band_hint, c[3]);
            patterns->scanned = true;
        }
        if (c[0] > H->cap) {
            // more hits than the caller's buffer takes: that is the caller's overflow (SPM_E_OVERFLOW from the views,
            // the count so far in stats.n_hits), not a reason to scan again
            H->n = c[0];
            H->counted = true;
        } else if (c[2] != 0) {
            // lists still too small: the whole range again, brute force
            H->stats.fell_back = 1;
            use_filter = false;
            SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));
        } else if (c[6] != 0) {
            // ---- span-local fallback: only the spans that gave up are scanned again, by the brute-force kernel ----
            const uint64_t n_ovf = c[6];
            std::vector<uint64_t> ov(2 * n_ovf);
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(ov.data(), A.d_ovf, ov.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            // a window that starts in span [b, b + len) belongs to occurrences whose last symbol lies in
            // [b - 16, b + len + max_window): those are scanned again (clipped to the owned range), merged where they touch
            std::vector<std::pair<uint64_t, uint64_t>> rg;
            rg.reserve(n_ovf);
            for (uint64_t i = 0; i < n_ovf; ++i) {
                const uint64_t b = ov[2 * i], len = ov[2 * i + 1];
                const uint64_t lo = std::max<uint64_t>(A.begin, b >= 16 ? b - 16 : 0);
                const uint64_t hi = std::min<uint64_t>(end, b + len + patterns->max_window);
                if (lo < hi)
                    rg.emplace_back(lo, hi);
            }
            std::sort(rg.begin(), rg.end());
            std::vector<std::pair<uint64_t, uint64_t>> mg;
            for (const auto &r : rg) {
                if (!mg.empty() && r.first <= mg.back().second)
                    mg.back().second = std::max(mg.back().second, r.second);
                else
                    mg.push_back(r);
            }
            uint64_t total = 0;
            for (const auto &r : mg)
                total += r.second - r.first;
            const uint64_t warm = patterns->max_window > 0 ? patterns->max_window - 1 : 0;
            // tile length: enough tiles to fill the machine, long enough that the warm-up stays a small share
            const uint64_t want_tiles = (uint64_t)ctx->n_cu * 32 / std::max(1u, patterns->n_groups) + 1;
            uint64_t tile = std::max<uint64_t>(std::max<uint64_t>(1024, (warm * 8 + 255) & ~255ull), (total / want_tiles + 255) & ~255ull);
            tile = std::min<uint64_t>(tile, 1u << 20);
            std::vector<uint64_t> tab;
            const uint64_t *segs = nullptr;
            if (segmented) { // every segment is a haystack of its own: the tiles follow the segment table
                if (A.seg_offsets) {
                    segs = A.seg_offsets;
                } else {
                    A.seg_host.resize(A.n_segments + 1);
                    SPM_HIP_CHECK(ctx, hipMemcpyAsync(A.seg_host.data(), A.d_seg_offsets, (A.n_segments + 1) * sizeof(uint64_t),
                                                      hipMemcpyDeviceToHost, ctx->stream));
                    SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    segs = A.seg_host.data();
                }
            }
            for (const auto &r : mg) {
                if (!segs) {
                    for (uint64_t lo = r.first; lo < r.second; lo += tile) {
                        const uint64_t hi = std::min(lo + tile, r.second);
                        tab.push_back(lo >= A.ctx_begin + warm ? lo - warm : A.ctx_begin);
                        tab.push_back(lo);
                        tab.push_back(hi);
                    }
                    continue;
                }
                // the segments that meet [r.first, r.second): the one holding r.first, then on
                uint64_t sidx = (uint64_t)(std::upper_bound(segs, segs + A.n_segments + 1, r.first) - segs);
                sidx = sidx ? sidx - 1 : 0;
                for (; sidx < A.n_segments && segs[sidx] < r.second; ++sidx) {
                    const uint64_t sb = segs[sidx], se = segs[sidx + 1];
                    const uint64_t o_lo = std::max(r.first, sb), o_hi = std::min(r.second, se);
                    for (uint64_t lo = o_lo; lo < o_hi; lo += tile) {
                        const uint64_t hi = std::min(lo + tile, o_hi);
                        tab.push_back(lo >= sb + warm ? lo - warm : sb); // (cold start inside the segment)
                        tab.push_back(lo);
                        tab.push_back(hi);
                    }
                }
            }
            H->stats.fallback_spans = (uint32_t)std::min<uint64_t>(n_ovf, 0xFFFFFFFFu);
            if (!tab.empty()) {
                if (tab.size() / 3 > 0xFFFFFFFFull) {
                    SPM_SET_ERR(ctx, "span-local fallback: too many tiles");
                    return SPM_E_UNSUPPORTED;
                }
                A.tiles = &tab;
                rc = run_brute(A, begin, end, A.ctx_begin, nullptr, nullptr, true, false);
                A.tiles = nullptr;
                if (rc != SPM_OK)
                    return rc;
                H->stats.main_launches--; // (run_brute counts itself as a main launch: ms_main stays the filter's)
                SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream)); // the re-scan counts as verification time
                SPM_HIP_CHECK(ctx, hipMemcpyAsync(c, H->d_count, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
                SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            }
            H->stats.fallback_symbols = total;
            if (c[2] != 0 && c[0] <= H->cap && !A.seen_full) {
                // the dedupe set ran out during the re-scan (it was sized for what earlier scans reported): once more,
                // sized for the caller's hit buffer
                A.seen_full = true;
                again = true;
                H->stats.main_launches = 0;
                SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));
            } else if (c[2] != 0 && c[0] <= H->cap) {
                // still not enough: start over with the brute-force engine
                H->stats.fell_back = 1;
                use_filter = false;
                SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));
            } else {
                H->n = c[0];
                H->counted = true;
            }
        } else {
            H->n = c[0];
            H->counted = true;
            if (after_launch && !stateful) { // nothing was added to the hit list after the caller's work ran on it
                H->hook_final = true;
                H->fan_count = c[12];
            }
        }
        if (!again)
            break;
        }
    }
    if (use_filter && stateful) {
        // ---- the brute-force kernel's share of a filtered chunk: its first window - 1 symbols, and the exit state ----
        A.begin = begin;
        uint32_t *d_in = nullptr, *d_out = nullptr;
        std::vector<uint32_t> h_in;
        const uint32_t rows = patterns->is_myers() ? 2 * patterns->NW + 1 : patterns->NW;
        const size_t st_words = (size_t)patterns->n_groups * rows * 64;
        dev_scratch tmp;
        SPM_HIP_CHECK(ctx, tmp.alloc(&d_in, st_words * 4 * 2));
        d_out = d_in + st_words;
        if (has_state) {
            state_to_internal(patterns, state_in, h_in);
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h_in.data(), st_words * 4, hipMemcpyHostToDevice, ctx->stream));
            int rc = run_brute(A, begin, begin + state_prefix, begin, d_in, nullptr, true, true);
            if (rc != SPM_OK)
                return rc;
            H->stats.main_launches--; // (ms_main stays the filter's)
            H->counted = false;       // more hits may have arrived
        }
        if (state_out) {
            const uint64_t range = end - begin;
            const uint64_t tail = std::min<uint64_t>(range, 2ull * patterns->max_m + 4);
            const uint64_t tb = end - tail;
            const bool from_state = has_state && tb == begin;
            int rc = run_brute(A, tb, end, tb, from_state ? d_in : nullptr, d_out, false, true);
            if (rc != SPM_OK)
                return rc;
            H->stats.main_launches--;
        }
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream));
        std::vector<uint32_t> h_out(st_words);
        if (state_out)
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(h_out.data(), d_out, st_words * 4, hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // (h_in is a host temporary; d_in is freed on return)
        if (state_out)
            state_from_internal(patterns, h_out, state_out);
    }
    if (!use_filter) {
        H->stats.engine_used = SPM_ENGINE_BRUTE;
        // state plumbing
        uint32_t *d_in = nullptr, *d_out = nullptr;
        std::vector<uint32_t> h_in;
        const uint32_t rows = patterns->is_myers() ? 2 * patterns->NW + 1 : patterns->NW;
        const size_t st_words = (size_t)patterns->n_groups * rows * 64;
        if (has_state || state_out) {
            SPM_HIP_CHECK(ctx, hipMalloc(&d_in, st_words * 4 * 2));
            d_out = d_in + st_words;
        }
        if (has_state) {
            state_to_internal(patterns, state_in, h_in);
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h_in.data(), st_words * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[1], ctx->stream));
        const uint64_t range = end - begin;
        // a scan that must hand back an exact state and fits one tile does both in one pass
        const bool one_pass_state = state_out && (range <= (1u << 16) || patterns->algo == SPM_ALGO_MYERS_PREFIX);
        int rc = run_brute(A, begin, end, A.ctx_begin, has_state ? d_in : nullptr, one_pass_state ? d_out : nullptr,
                           true, one_pass_state);
        if (rc != SPM_OK) {
            hipFree(d_in);
            return rc;
        }
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[2], ctx->stream));
        if (state_out && !one_pass_state) {
            // State after the last symbol: sequentially exact from a cold start 2*max|P| symbols earlier
            // (every DP cell D[i][j] <= i has an optimal alignment spanning <= 2i symbols).
            const uint64_t tail = std::min<uint64_t>(range, 2ull * patterns->max_m + 4);
            const uint64_t tb = end - tail;
            const bool from_state = has_state && tb == begin;
            rc = run_brute(A, tb, end, tb, from_state ? d_in : nullptr, d_out, false, true);
            if (rc != SPM_OK) {
                hipFree(d_in);
                return rc;
            }
        }
        SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream));
        if (state_out) {
            std::vector<uint32_t> h_out(st_words);
            SPM_HIP_CHECK(ctx, hipMemcpyAsync(h_out.data(), d_out, st_words * 4, hipMemcpyDeviceToHost, ctx->stream));
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            state_from_internal(patterns, h_out, state_out);
        }
        if (d_in) {
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(d_in);
        }
    }
    if (spm_trace_on()) { // (costs one event synchronisation: diagnostics only)
        spm_scan_stats st{};
        spm_hip_hits_stats(H.get(), &st);
        fprintf(stderr, "[spm_hip] scan [%llu, %llu)%s: engine %s%s, %u main launch(es); %.3f ms (main %.3f, verification %.3f); "
                        "%llu seed-checked pairs, %u bands, %llu hits%s; host %.3f ms\n",
                (unsigned long long)begin, (unsigned long long)end, seg_offsets || d_seg_offsets ? " segmented" : "",
                st.engine_used == SPM_ENGINE_FILTER ? (patterns->filter_dense ? "filter (dense pass)" : "filter") : "brute",
                st.fell_back ? " after a whole-scan fallback" : "", st.main_launches, st.ms_total, st.ms_main, st.ms_verify,
                (unsigned long long)st.n_candidates, st.n_bands, (unsigned long long)st.n_hits,
                st.fallback_spans ? " (spans re-scanned by the brute-force kernel)" : "", ms_since(t_call));
    }
    *out = H.release();
    return SPM_OK;
}

// ----------------------------------------------------------------------------------------------------
// hits
// ----------------------------------------------------------------------------------------------------
static int hits_count(spm_hits *h)
{
    spm_ctx *ctx = h->ctx;
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

extern "C" int spm_hip_hits_copy_fused(spm_hits *h, void *device_dst, uint64_t cap, uint64_t *n)
{
    if (!h || !n || !device_dst)
        return SPM_E_INVALID;
    int rc = hits_count(h);
    if (rc != SPM_OK)
        return rc;
    *n = h->n;
    // header: the count, from memory that outlives the call (the context's pinned block: slots 13 / 14 belong to nobody
    // else; an asynchronous copy from a stack array reads a dead frame if the runtime does not stage it first).  The
    // stream orders this copy behind the previous call's, so the slots are free to overwrite once that copy was issued
    // -- which a host that reuses them needs a fence for: wait for the last header copy before writing the next.
    spm_ctx *ctx = h->ctx;
    if (ctx->fused_hdr_ev) {
        if (ctx->fused_hdr_pending)
            SPM_HIP_CHECK(ctx, hipEventSynchronize(ctx->fused_hdr_ev));
    } else {
        SPM_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->fused_hdr_ev, hipEventDisableTiming));
    }
    ctx->h_counters[13] = h->n;
    ctx->h_counters[14] = 0;
    SPM_HIP_CHECK(ctx, hipMemcpyAsync(device_dst, ctx->h_counters + 13, 16, hipMemcpyHostToDevice, ctx->stream));
    SPM_HIP_CHECK(ctx, hipEventRecord(ctx->fused_hdr_ev, ctx->stream));
    ctx->fused_hdr_pending = true;
    const uint64_t c = std::min(h->n, cap);
    if (c)
        SPM_HIP_CHECK(h->ctx, hipMemcpyAsync(static_cast<uint8_t *>(device_dst) + sizeof(spm_hit), h->d_hits, c * sizeof(spm_hit),
                                             hipMemcpyDeviceToDevice, h->ctx->stream));
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
        h->ctx->pool.push_back(b); // stream order makes reuse by the next scan safe
    } else {
        if (h->ctx)
            hipStreamSynchronize(h->ctx->stream);
        hipFree(h->d_hits);
        hipFree(h->d_count);
        for (int i = 0; i < 6; ++i)
            if (h->ev[i])
                hipEventDestroy(h->ev[i]);
    }
    delete h;
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

// Host-only self-check of the seed index (no device, no context): index_build.hpp
extern "C" int spm_hip_host_selftest(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                                     const uint16_t *k, uint32_t sigma, uint64_t *stats)
{
    return spm_hip::host_selftest(algo, ranks_concat, offsets, n_patterns, k, sigma, stats);
}

extern "C" const char *spm_hip_version(void) { return "libspm_hip 0.1 (gfx950)"; }

#include "jst.hpp"
#include "comm.hpp"
