// patterns.hip -- needle sets behind the C ABI (include/spm_hip.h): what the reference does in its matcher constructors
// (/root/reference/libspm/libspm/matcher/myers_matcher.hpp:40-43) -- match-mask tables, seed index, state blobs.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "tables_build.hpp"

// Every table of a needle set lives in ONE device allocation and travels in one stream of pinned chunks: the sources are laid
// end to end (256-byte aligned), the context's two staging halves are filled by turns (a thread team copies when the set is
// large) and each half goes up with one asynchronous copy while the other is being filled.  Twenty hipMalloc + pageable
// hipMemcpy calls cost 12 ms for 1024 needles and 55 ms for 100 000; this costs what the bytes cost.
namespace
{
struct needle_upload
{
    struct item
    {
        void **dst;
        const void *src;
        size_t bytes, off;
    };
    std::vector<item> items;
    size_t total = 0;
    size_t add(void **dst, const void *src, size_t bytes, size_t zero_tail = 0)
    {
        items.push_back(item{dst, src, bytes, total});
        total += (std::max<size_t>(bytes + zero_tail, 16) + 255) & ~(size_t)255;
        return items.size() - 1;
    }
    // the allocation; every *dst points into it afterwards
    hipError_t place(void **arena)
    {
        hipError_t e = hipMalloc(arena, std::max<size_t>(total, 256));
        if (e != hipSuccess)
            return e;
        for (item &it : items)
            *it.dst = static_cast<uint8_t *>(*arena) + it.off;
        return hipSuccess;
    }
    // bytes [lo, hi) of the laid-out image into `out` (gaps and tails are zero)
    void fill(uint8_t *out, size_t lo, size_t hi) const
    {
        memset(out, 0, hi - lo);
        size_t i = std::upper_bound(items.begin(), items.end(), lo, [](size_t v, const item &it) { return v < it.off; }) - items.begin();
        i = i ? i - 1 : 0;
        for (; i < items.size() && items[i].off < hi; ++i) {
            const item &it = items[i];
            const size_t a = std::max(lo, it.off), b = std::min(hi, it.off + it.bytes);
            if (a < b && it.src)
                memcpy(out + (a - lo), static_cast<const uint8_t *>(it.src) + (a - it.off), b - a);
        }
    }
    hipError_t stream(spm_ctx *ctx, void *arena, unsigned n_threads) const
    {
        const size_t half = ctx->stage_half;
        std::unique_ptr<thread_team> team;
        if (total > 4 * half && n_threads > 1)
            team.reset(new thread_team(std::min(n_threads, 8u)));
        hipError_t e = hipSuccess;
        unsigned c = 0;
        for (size_t lo = 0; lo < total && e == hipSuccess; lo += half, ++c) {
            const size_t hi = std::min(total, lo + half);
            uint8_t *h = ctx->h_stage + (c & 1) * half;
            if (c >= 2)
                e = hipEventSynchronize(ctx->stage_ev[c & 1]); // the copy that last read this half
            if (e != hipSuccess)
                break;
            if (team)
                team->run((hi - lo + 65535) / 65536, [&](size_t b0, size_t b1, unsigned) {
                    const size_t a = lo + b0 * 65536, b = std::min(hi, lo + b1 * 65536);
                    if (a < b)
                        fill(h + (a - lo), a, b);
                });
            else
                fill(h, lo, hi);
            e = hipMemcpyAsync(static_cast<uint8_t *>(arena) + lo, h, hi - lo, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipEventRecord(ctx->stage_ev[c & 1], ctx->stream);
        }
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream); // the staging halves are free again; the caller's vectors may go
        return e;
    }
};
} // namespace

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
    needle_upload up; // every table of the set: one device allocation, one stream of pinned chunks at the end
    auto upload = [&](auto **dst, const void *src, size_t bytes) -> hipError_t {
        up.add(reinterpret_cast<void **>(dst), src, bytes);
        return hipSuccess;
    };
    double ms_upload = 0;
    long pass_tab_item = -1;
    std::vector<uint32_t> pk, pk_off;
    std::vector<uint8_t> surplus;
    auto t0 = clk::now();

    // ---- match-mask tables: [group][row][word][lane], needles top-aligned (see brute.hpp) ----
    brute_tables bt;
    {
        build_brute_tables(nv, ps->n_groups, NW, sigma <= 5 || sigma == 15, tune.n_threads(), bt);
        ps->build.ms_tables = ms_since(t0);
        SPM_HIP_CHECK(ctx, upload(&ps->d_peq, bt.peq.data(), bt.peq.size() * sizeof(uint32_t)));
        if (!bt.verify.empty()) // the filter engine verifies exact matchers with the Myers recurrence at k = 0
            SPM_HIP_CHECK(ctx, upload(&ps->d_peq_verify, bt.verify.data(), bt.verify.size() * sizeof(uint32_t)));
        if (!bt.bot.empty())
            SPM_HIP_CHECK(ctx, upload(&ps->d_peq_bot, bt.bot.data(), bt.bot.size() * sizeof(uint32_t)));
        if (!bt.hp0.empty())
            SPM_HIP_CHECK(ctx, upload(&ps->d_hp0, bt.hp0.data(), bt.hp0.size() * sizeof(uint32_t)));
        SPM_HIP_CHECK(ctx, upload(&ps->d_m, ps->m.data(), ps->m.size() * sizeof(int32_t)));
        SPM_HIP_CHECK(ctx, upload(&ps->d_k, ps->k.data(), ps->k.size() * sizeof(int32_t)));
    }

    // ---- filter engine tables (verification reads the brute table) ----
    if ((sigma == 4 || sigma == 5 || sigma == 15) && algo != SPM_ALGO_MYERS_PREFIX && n_patterns > 0) {
        const auto ti = clk::now();
        int rc = build_filter_index(nv, tune, *ps);
        if (rc != SPM_OK)
            return rc;
        ps->build.ms_index = ms_since(ti);
        if (!ps->fidx.empty()) {
            for (filter_index &F : ps->fidx) {
                SPM_HIP_CHECK(ctx, upload(&F.d_bitmap, F.h_image.data(), F.h_image.size() * sizeof(uint32_t)));
                SPM_HIP_CHECK(ctx, upload(&F.d_ht, F.h_ht.data(), F.h_ht.size() * sizeof(u32x4)));
                if (!F.h_buckets.empty())
                    SPM_HIP_CHECK(ctx, upload(&F.d_buckets, F.h_buckets.data(), F.h_buckets.size() * sizeof(uint16_t)));
                ps->build.keys += F.n_keys;
                for (uint32_t d = 0; d < 16; ++d)
                    ps->build.anchor_sixteenths += (F.dimer_set >> d) & 1u;
            }
            pass_tab_item = (long)up.add(reinterpret_cast<void **>(&ps->d_pass_tab), nullptr, ps->fidx.size() * sizeof(pass_entry));
            SPM_HIP_CHECK(ctx, upload(&ps->d_entries, ps->h_entries.data(), ps->h_entries.size() * sizeof(u32x4)));
            // (64 zero bytes behind the symbols: resolve_kernel reads whole dwords around a seed)
            up.add(reinterpret_cast<void **>(&ps->d_ranks), ps->ranks.data(), ps->ranks.size(), 64);
            SPM_HIP_CHECK(ctx, upload(&ps->d_offsets, ps->offsets.data(), ps->offsets.size() * sizeof(uint32_t)));
            if (ps->sigma == 4) { // the same symbols 2 bits each, 16 per word, every needle from a word of its own
                pack_needles(nv, pk, pk_off, tune.n_threads());
                SPM_HIP_CHECK(ctx, upload(&ps->d_needle_pk, pk.data(), pk.size() * sizeof(uint32_t)));
                SPM_HIP_CHECK(ctx, upload(&ps->d_pk_offsets, pk_off.data(), pk_off.size() * sizeof(uint32_t)));
            }
            SPM_HIP_CHECK(ctx, upload(&ps->d_seed_q, ps->seed_q.data(), ps->seed_q.size() * sizeof(uint16_t)));
        }
        if (!ps->fidx.empty() && ps->max_k >= kMergeMinK && ps->max_k <= 1000) {
            surplus.assign(ps->m.size(), 1);
            for (uint32_t p = 0; p < ps->n; ++p)
                surplus[p] = (uint8_t)(ps->seed_n[p] - (uint32_t)ps->k[p]);
            SPM_HIP_CHECK(ctx, upload(&ps->d_surplus, surplus.data(), surplus.size()));
        }
    }
    {
        const auto tu = clk::now();
        SPM_HIP_CHECK(ctx, up.place(&ps->d_arena));
        const double ms_place = ms_since(tu);
        std::vector<pass_entry> pt;
        for (filter_index &F : ps->fidx)
            pt.push_back(pass_entry{reinterpret_cast<const uint4 *>(F.d_ht), F.ht_mask, 0});
        if (pass_tab_item >= 0)
            up.items[(size_t)pass_tab_item].src = pt.data();
        SPM_HIP_CHECK(ctx, up.stream(ctx, ps->d_arena, tune.n_threads()));
        for (filter_index &F : ps->fidx) {
            F.h_image = std::vector<uint32_t>();
            F.h_ht = std::vector<u32x4>();
            F.h_buckets = std::vector<uint16_t>();
        }
        ps->h_entries = std::vector<u32x4>();
        ps->build.bytes_device = up.total;
        ms_upload = ms_since(tu);
        if (spm_trace_on())
            fprintf(stderr, "[spm_hip] patterns_create: one allocation of %.2f MiB in %.2f ms, streamed up in %.2f ms\n",
                    up.total / 1048576.0, ms_place, ms_upload - ms_place);
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
    hipFree(p->d_arena); // (every d_* table of the set points into it)
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
void state_to_internal(const spm_patterns *p, const void *state, std::vector<uint32_t> &out)
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

void state_from_internal(const spm_patterns *p, const std::vector<uint32_t> &in, void *state)
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

// Host-only self-check of the seed index (no device, no context): index_build.hpp
extern "C" int spm_hip_host_selftest(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                                     const uint16_t *k, uint32_t sigma, uint64_t *stats)
{
    return spm_hip::host_selftest(algo, ranks_concat, offsets, n_patterns, k, sigma, stats);
}

