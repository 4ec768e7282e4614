// scan_filter.hip -- the seed-filter engine (filter.hpp): buffer sizing and the launches of one scan.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "filter.hpp"

namespace
{
template <int NWN>
void launch_verify_nw(const verify_params &V, dim3 grid, hipStream_t s)
{
    // LDS holds (sigma+1)*NWN words per thread; keep the block within ~128 KiB
    uint32_t threads = 256;
    const size_t per_thread = (size_t)(V.sigma + 1) * NWN * 4 + 2 * 16; // match masks + the hits of one 16-symbol block
    while (threads > 64 && per_thread * threads > 128 * 1024)
        threads >>= 1;
    const size_t lds = per_thread * threads + (size_t)(threads / 64) * kHitStage * sizeof(spm_hit);
    hipFuncSetAttribute((const void *)verify_kernel<NWN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((verify_kernel<NWN>), grid, dim3(threads), lds, s, V);
}

template <int G, int NB>
void launch_verify_wave_g(const scan_tuning &T, verify_params V, const uint32_t *peq_bot, uint32_t max_m, dim3 grid, hipStream_t s)
{
    // One wave per workgroup: a scan leaves a few thousand long bands, i.e. far fewer busy waves than the GPU has SIMDs,
    // and each is a serial chain ~1500 steps long.  With four-wave workgroups filled in order, the dispatcher packed the
    // busy waves four to a SIMD on a third of the CUs and left the rest idle.
    const uint32_t threads = (uint32_t)std::max(64, std::min(256, T.verify_wave_threads));
    grid.x *= 256 / threads;
    const uint32_t n_slots = 2 * V.max_k + 1 + V.max_span;
    // text window of one candidate: cold start |P| + k symbols before the first end position, then the end positions
    V.wave_text = ((max_m + V.max_k + n_slots + 16 + 15) & ~15u) + 16;
    const size_t per_group = ((n_slots * 2 + 15) & ~15u) + V.wave_text;
    size_t lds = (size_t)(threads / 64) * (64 / G) * per_group;
    // (diagnostics: a larger LDS claim per workgroup caps how many of them a CU takes at once)
    lds = std::max<size_t>(lds, (size_t)std::max(0, std::min(160, T.verify_wave_lds_kb)) * 1024);
    hipFuncSetAttribute((const void *)verify_wave_kernel<G, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((verify_wave_kernel<G, NB>), grid, dim3(threads), lds, s, V, peq_bot);
}

// long needles: NB 32-row blocks per lane, G = lanes per band >= blocks of the longest needle / NB.  SPM_HIP_VERIFY_WAVE_NB=2:
// two blocks per lane from 17 blocks on (four |P| = 1024 bands share a wave instead of two).
void launch_verify_wave(const scan_tuning &T, uint32_t n_blocks, const verify_params &V, const uint32_t *peq_bot, uint32_t max_m,
                        dim3 grid, hipStream_t s)
{
    // (measured on C5, 2 798 bands of |P| = 1024: one block per lane 0.300 ms, two 0.358 -- a step is a chain of dependent
    // instructions, its latency and not its issue slots set the pace, and the second block lengthens the chain.)
    const bool two = T.verify_wave_nb >= 2;
    if (n_blocks <= 8)
        launch_verify_wave_g<8, 1>(T, V, peq_bot, max_m, grid, s);
    else if (n_blocks <= 16)
        launch_verify_wave_g<16, 1>(T, V, peq_bot, max_m, grid, s);
    else if (n_blocks <= 32) {
        if (two)
            launch_verify_wave_g<16, 2>(T, V, peq_bot, max_m, grid, s);
        else
            launch_verify_wave_g<32, 1>(T, V, peq_bot, max_m, grid, s);
    } else {
        if (two)
            launch_verify_wave_g<32, 2>(T, V, peq_bot, max_m, grid, s);
        else
            launch_verify_wave_g<64, 1>(T, V, peq_bot, max_m, grid, s);
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

} // namespace

// persistent band table of the context: empty between scans (the verification gives every slot back), so a scan pays
// for the bands it has, not for a memset of the table
int ensure_band_table(spm_ctx *ctx, uint64_t slots)
{
    if (ctx->band_slots >= slots && !ctx->band_dirty)
        return SPM_OK;
    if (ctx->band_slots < slots) {
        const auto t0 = clk::now();
        if (ctx->d_band_tab) {
            SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->d_band_tab);
            ctx->d_band_tab = nullptr;
            ctx->band_slots = 0;
        }
        SPM_HIP_CHECK(ctx, hipMalloc(&ctx->d_band_tab, slots * sizeof(ulonglong2)));
        ctx->band_slots = slots;
        if (spm_trace_on())
            fprintf(stderr, "[spm_hip] band table grows to %.1f MiB: %.2f ms\n", slots * sizeof(ulonglong2) / 1048576.0, ms_since(t0));
    }
    SPM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_band_tab, 0xFF, ctx->band_slots * sizeof(ulonglong2), ctx->stream)); // all free
    SPM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_table_poison, 0, 4, ctx->stream)); // (and known to be)
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
    // real texts are not uniform: room for the seed hits of repeat stretches -- one survivor per 2048 symbols (1 % of a
    // text in repeats needs one per 2800; a scan that outgrows its lists is repeated once with room for what it counted, and
    // the needle set remembers).  The first scan on a context pays for these buffers: sized for one survivor per 512
    // symbols (and as many bands) they were 2 GB of lists and a 4 GB band table to clear for a 16 GiB text -- 1.6 ms in
    // front of a 2.7 ms scan.
    est = std::max<uint64_t>(est, (A.end - A.begin) / (ps->scanned ? 4096 : 2048));
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
    const int cc = A.tune.cand_cap;
    if (cc > 0)
        surv_cap = (uint64_t)cc;
    // (+ the first chunk of every wave of resolve_kernel: n_cu x 8 workgroups of 4 waves)
    // bands: at most one per candidate pair, usually far fewer (uniform text: 1 000 for 20 000 survivors; 1 % repeats: 1.6 M
    // for 6 M; 5 %: as many as survivors) -- a quarter of the survivor slots unless earlier scans needed more
    const uint64_t band_want = std::max<uint64_t>(surv_cap / 4, 2 * ps->band_hint);
    uint64_t band_cap = std::max<uint64_t>(band_want, 4096) * (A.band_scale ? A.band_scale : 1) +
                        (uint64_t)ctx->n_cu * 32 * kChunkMin;
    if (A.tune.band_cap > 0 && !A.band_scale) // (tests force the band-list-full path with it; a repeated attempt sizes itself)
        band_cap = (uint64_t)A.tune.band_cap;
    uint64_t band_slots = 1u << 12;
    while (band_slots < 2 * band_cap)
        band_slots <<= 1;
    // bands: Bw diagonals each.  Sets with surplus seeds: (k+1) x factor, overlapping by k + 1 (wider bands mean fewer
    // occurrences whose seeds straddle two of them at the price of more end positions per verification; 4(k+1) measured
    // best for |P| = 1024, k = 64; the lane-per-band kernel keeps its end-position slots per thread: narrow bands there).
    // Other sets: 64 diagonals, no overlap -- every band with a seed hit is verified.
    uint32_t nwn = std::max(1u, (ps->max_m + 31) / 32);
    const int wave_min = A.tune.verify_wave_min_words; // 0 = never use the wave-per-band kernel
    // (the wave-per-band kernel keeps the match masks of <= 5 symbols in registers: dna15 sets use the lane-per-band one)
    const bool use_wave = ps->d_peq_bot && wave_min > 0 && nwn >= (uint32_t)wave_min && ps->sigma <= 5;
    const bool overlap = ps->d_surplus != nullptr;
    uint32_t Bw;
    if (overlap) {
        const uint32_t bw_factor = use_wave ? (uint32_t)std::max(1, A.tune.band_factor) : 1u;
        Bw = (ps->max_k + 1) * bw_factor;
        if (Bw + ps->max_k > 2047)
            Bw = ps->max_k + 1;
    } else {
        Bw = (uint32_t)std::max(8, std::min(64, A.tune.band)); // (one mask bit per diagonal)
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
    const size_t band_bytes = band_cap * sizeof(band_rec) * 2; // (+ the selected bands of overlapping sets / the heads of runs)
    const size_t ovf_bytes = kOvfCap * 2 * sizeof(uint64_t);
    int rc = ensure_scratch(ctx, surv_bytes + seen_bytes + band_bytes + ovf_bytes);
    if (rc != SPM_OK)
        return rc;
    // Exact sets whose needles are their own single seed (k = 0, no `N`, e.g. Shift-Or / Horspool sets): the whole-seed check
    // of the resolve kernel is the whole comparison, so it reports the hits itself -- no band table, no verification launch.
    // (Not for needles that are repeats: their merged index entries skip the per-offset check.)
    bool exact_hits = ps->max_k == 0 && !overlap && ps->filter_max_range == 0 && ps->d_ranks &&
                      A.tune.seed_check != 0 && A.tune.exact_from_resolve != 0;
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
    // Exact sets whose hits come from the resolve kernel report every occurrence once by construction (one sampled window,
    // one entry): no dedupe set, no 4 MiB memset in front of a 0.2 ms scan -- unless a span gives up (the brute-force
    // re-scan of that span would report its hits a second time): then the scan runs again with the set.
    const bool skip_seen = exact_hits && !A.need_seen && A.tune.exact_skip_dedupe != 0;
    const_cast<scan_args &>(A).seen_skipped = skip_seen;
    const_cast<scan_args &>(A).exact_used = exact_hits;
    if (!skip_seen)
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
    P.dense_debug = (uint32_t)A.tune.dense_debug;
    P.buckets = reinterpret_cast<const uint4 *>(F.d_buckets);
    if (fi > 0) // each pass draws its spans from a fresh head
        SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count + 4, 0, sizeof(unsigned long long), ctx->stream));
    const bool bits = F.hash_variant == 4; // presence bits + L2 buckets as level 1 of a sparse pass: the dense kernel's machinery
    const bool use_packed = !F.dense && !bits && A.text->d_packed && ps->sigma == 4 && F.hash_variant == 2 && F.stride >= 2 &&
                            !(A.opts.flags & SPM_SCAN_IGNORE_PACKED);
    // measured best: 8 waves per CU on the 1-byte text when HBM binds, 16 on the 2-bit shadow and at stride 1 with
    // 16-symbol keys (LDS-bound: C4 25.8 vs 29.1 ms; the other stride-1/2 variants need more than 128 VGPRs)
    // stride 2: two chunks per group (16 windows per lane) need < 128 VGPRs, so 16 waves per CU hide the LDS round trips
    // (C5: 0.53 -> 0.46 ms; four chunks per group hold 167 VGPRs at 8 waves)
    const bool narrow2 = F.stride == 2 && !use_packed && A.tune.s2_u == 2;
    const bool wide_ok = use_packed || narrow2 || (F.stride == 1 && F.key_len >= 16 && ps->sigma == 4 &&
                                                   !A.tune.force_masked);
    const uint32_t threads = (F.dense || bits) ? 1024u : (uint32_t)std::max(
        64, std::min(wide_ok ? 1024 : 512, (A.tune.filter_threads > 0 ? A.tune.filter_threads : (wide_ok ? 1024 : 512))));
    // + the workgroup's span-dequeue slot (4 words) + one survivor-chunk record per wave (+ dense: one queue per wave)
    const size_t lds = (size_t)F.lds_words * 4 + 16 + 16 * kCandRec * 4 + ((F.dense || bits) ? 16 * sizeof(dense_queue) : 0);
    const uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 2048 / threads));
    const uint32_t grid = ctx->n_cu * wg_per_cu;
    const uint64_t n_waves = (uint64_t)grid * (threads / 64);
    const uint64_t n_chunks = (P.hi - (P.lo & ~1023ull) + 1023) / 1024;
    uint64_t span = n_chunks / (n_waves * (uint64_t)std::max(1, (A.tune.spans_per_wave > 0 ? A.tune.spans_per_wave : 32))) + 1;
    // small texts: at least 64 KiB per dequeue as long as every wave still gets ~4 spans (a 1 GiB text ran 14 % faster
    // with 64-chunk spans than with the 24 the rule above gives: fewer dequeue rounds, each a workgroup barrier)
    if (span < 64)
        span = std::max<uint64_t>(span, std::min<uint64_t>(64, n_chunks / (n_waves * 4) + 1));
    span = std::min<uint64_t>(std::max<uint64_t>(span, 8), 4096);
    const int fs = A.tune.span;
    if (fs > 0)
        span = (uint64_t)fs;
    span = (span + 7) & ~7ull; // whole groups of chunks
    P.span_chunks = (uint32_t)span;
    P.span_unit = 1024;
    // candidates a span may produce before it gives up and is re-scanned by the brute-force kernel: one per 4 symbols
    // costs the verification about what the re-scan would
    {
        const int sb = A.tune.span_budget;
        P.span_budget = sb > 0 ? (uint32_t)sb : (uint32_t)std::max<uint64_t>(256, span * 1024 / 4);
    }
    // span dequeue: per wave while the dequeue rate stays far below what one atomic word sustains (~88/us, i.e.
    // spans >= 192 KiB at 7 TB/s), per workgroup otherwise (measured: C3 2.52 vs 2.59 ms, C2 0.88 vs 0.20 ms)
    const int dyn = A.tune.dyn;
    P.dynamic = dyn >= 0 ? (uint32_t)std::min(2, dyn) : (span >= 192 ? 1u : 2u);

    const int U = A.tune.filter_u >= 8 ? 8 : 4;
    const bool NT = A.tune.nt != 0;
    P.hash_variant = F.hash_variant;
    const bool short_keys = F.key_len < 16 || A.tune.force_masked != 0; // (the env: diagnostics)
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
    if (bits) {
#define LAUNCH_BITS(S, KM)                                                                                             \
    do {                                                                                                               \
        hipFuncSetAttribute((const void *)seed_filter_dense_kernel<4, 1, S, KM>,                                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
        hipLaunchKernelGGL((seed_filter_dense_kernel<4, 1, S, KM>), dim3(grid), dim3(threads), lds, ctx->stream, P);   \
    } while (0)
        const bool km = F.key_len < 16;
        if (F.stride == 1) {
            if (km)
                LAUNCH_BITS(1, true);
            else
                LAUNCH_BITS(1, false);
        } else {
            if (km)
                LAUNCH_BITS(2, true);
            else
                LAUNCH_BITS(2, false);
        }
#undef LAUNCH_BITS
    } else if (F.dense) {
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
        uint64_t pspan = n_pchunks / (n_waves * (uint64_t)std::max(1, (A.tune.spans_per_wave > 0 ? A.tune.spans_per_wave : 8))) + 1;
        pspan = std::min<uint64_t>(std::max<uint64_t>(pspan, 4), 4096);
        pspan = (pspan + 3) & ~3ull;
        Q.span_chunks = (uint32_t)pspan;
        Q.span_unit = 4096;
        if (A.tune.span_budget <= 0)
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
            const int au = A.tune.anchor_u;
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
    R.needle_ranks = A.tune.seed_check ? ps->d_ranks : nullptr;
    R.needle_offsets = ps->d_offsets;
    R.seed_q = ps->d_seed_q;
    R.flank_check = (ps->sigma == 4 && !overlap && R.needle_ranks && A.tune.flank_check) ? 1u : 0u;
    R.pieces_check = A.tune.pieces_check ? R.flank_check : 0u;
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
    R.seen = skip_seen ? nullptr : d_seen;
    R.seen_mask = (uint32_t)(seen_slots - 1);
    R.hits = H->d_hits;
    R.hit_counter = H->d_count;
    R.overflow = H->d_count + 2;
    R.hit_cap = H->cap;
    R.table_poison = ctx->d_table_poison;
    R.debug_stage = (uint32_t)A.tune.resolve_debug;
    R.table_mask = (uint32_t)(band_slots - 1);
    R.bands = d_bands;
    R.band_cap = band_cap;
    // grid: what earlier scans of this needle set produced (a full grid of idle workgroups costs ~30 us on a 2.6 ms scan);
    // a scan that produces more simply loops
    const uint64_t surv_expect = ps->cand_hint ? 2 * ps->cand_hint : surv_cap;
    // (5 workgroups per CU are resident at once -- LDS queues, 84 VGPRs --: a larger grid only adds a second, partly filled
    // round.  A lane takes ~4 survivors in turn: measured on C5, whose survivors are few and cheap, 0.137 -> 0.10 ms;
    // c3r 1.33 -> 1.25 ms with the cap alone.)
    const uint64_t rmax = (uint64_t)ctx->n_cu * (uint64_t)std::max(1, A.tune.resolve_wgs_per_cu);
    // (a short survivor list: one survivor per lane, its latency is the kernel's; a long one: four per lane)
    const uint64_t per_wg = (surv_expect + 255) / 256 <= rmax ? 256 : (uint64_t)std::max(256, A.tune.resolve_surv_per_wg);
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
    V.table_mask = (uint32_t)(band_slots - 1);
    V.band_bits = 43 - seg_bits;
    // runs of adjacent bands (filter.hpp, band_runs_kernel): when earlier scans of this set left a long band list -- a
    // repeat-rich text --, sets without surplus seeds, lane-per-band verification
    const bool runs = !overlap && !exact_hits && !use_wave && Bw <= 32 && !A.need_seen && A.tune.verify_runs != 0 &&
                      ps->band_hint >= (uint64_t)std::max(0, A.tune.verify_runs_min_bands);
    if (runs) {
        // (heads report most end positions without asking the dedupe set: if a span gives up, the brute-force re-scan
        // could report them again -- the scan then runs once more without runs, as for exact sets without the set)
        const_cast<scan_args &>(A).seen_skipped = true;
        hipLaunchKernelGGL(band_runs_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, V, d_bands, d_bands + band_cap,
                           H->d_count + 11);
        SPM_HIP_CHECK(ctx, hipGetLastError());
        V.runs = 1;
        V.bands = d_bands + band_cap;
        V.band_counter = 11;
    }
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
        launch_verify_wave(A.tune, nwn, V, ps->d_peq_bot, ps->max_m, dim3(ctx->n_cu * 16), ctx->stream);
    } else {
        if (nwn > 8)
            nwn = ps->NW; // power of two beyond 8 words
        const uint64_t band_expect = ps->band_hint ? 2 * ps->band_hint : band_cap;
        const uint32_t vgrid = (uint32_t)std::min<uint64_t>((uint64_t)ctx->n_cu * 4, std::max<uint64_t>(ctx->n_cu / 2, (band_expect + 255) / 256));
        launch_verify(nwn, V, dim3(vgrid), ctx->stream);
    }
    SPM_HIP_CHECK(ctx, hipGetLastError());
    if (runs) {
        hipLaunchKernelGGL(band_release_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, V, d_bands, 3u);
        SPM_HIP_CHECK(ctx, hipGetLastError());
    }
    SPM_HIP_CHECK(ctx, hipEventRecord(H->ev[3], ctx->stream));
    H->cand_cap = surv_cap;
    H->band_cap = band_cap;
    return SPM_OK;
}


void spm_warm_filter_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)resolve_kernel);
}
