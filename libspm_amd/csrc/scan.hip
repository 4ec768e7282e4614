// scan.hip -- the scan driver behind spm_hip_scan / spm_hip_scan_segments: what seqan_pattern_base::operator() does
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-71) -- choose the engine, run it, handle overflow and fallbacks.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "filter_shared.hpp"

int ensure_scratch(spm_ctx *ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes)
        return SPM_OK;
    const auto t0 = clk::now();
    if (ctx->d_scratch) {
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        SPM_HIP_CHECK(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    // what a scan needs follows what earlier scans counted, and those counts wobble by a few per cent from run to run (slots
    // are drawn in chunks): a quarter of headroom, so that a text with millions of candidates does not pay for a new multi-GB
    // allocation (hundreds of ms) every few scans
    size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&ctx->d_scratch, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        SPM_HIP_CHECK(ctx, hipMalloc(&ctx->d_scratch, want));
    }
    ctx->scratch_bytes = want;
    if (spm_trace_on())
        fprintf(stderr, "[spm_hip] scratch (survivor / band lists, dedupe set) grows to %.1f MiB: %.2f ms\n", want / 1048576.0, ms_since(t0));
    return SPM_OK;
}

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

int scan_impl(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end, const spm_patterns *patterns,
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
    bool counters_clear = false;
    {
        // recycle the buffers of an earlier scan (hipMalloc/hipEventCreate per scan cost ~0.2 ms)
        bool reused = false;
        for (size_t i = 0; i < ctx->pool.size(); ++i)
            if (ctx->pool[i].cap == H->cap) {
                const hits_block b = ctx->pool[i];
                counters_clear = b.zeroed;
                ctx->pool.erase(ctx->pool.begin() + i);
                H->d_hits = b.d_hits;
                H->d_count = b.d_count;
                H->h_c = b.h_c;
                H->ev_done = b.ev_done;
                for (int e = 0; e < 4; ++e)
                    H->ev[e] = b.ev[e];
                reused = true;
                break;
            }
        if (!reused) {
            const auto ta = clk::now();
            SPM_HIP_CHECK(ctx, hipMalloc(&H->d_hits, std::max<uint64_t>(H->cap, 1) * sizeof(spm_hit)));
            if (spm_trace_on())
                fprintf(stderr, "[spm_hip] a new hit buffer (%llu records): %.2f ms\n", (unsigned long long)H->cap, ms_since(ta));
            SPM_HIP_CHECK(ctx, hipMalloc(&H->d_count, 16 * sizeof(unsigned long long)));
            for (int i = 0; i < 4; ++i)
                SPM_HIP_CHECK(ctx, hipEventCreate(&H->ev[i]));
        }
    }
    if (!counters_clear) // (a recycled block was cleared when it went back to the pool, off this scan's critical path)
        SPM_HIP_CHECK(ctx, hipMemsetAsync(H->d_count, 0, 16 * sizeof(unsigned long long), ctx->stream));

    scan_args A{ctx, text, begin, end, opts.left_context ? 0 : begin, patterns, opts, state_in, state_out, H.get()};
    A.tune = scan_tuning::from_env();
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
        for (int outer = 0; outer < 3; ++outer) { // (further rounds: the dedupe set was left out, or too small for the re-scan's hits)
        bool again = false;
        for (int attempt = 0;; ++attempt) {
            rc = run_filter(A);
            if (rc != SPM_OK)
                return rc;
            if ((opts.flags & SPM_SCAN_DEFER) && outer == 0 && attempt == 0 && !stateful && !segmented && !after_launch) {
                // deferred completion: the counters travel to this result's own pinned block; nobody waits for them now
                if (!H->h_c)
                    SPM_HIP_CHECK(ctx, hipHostMalloc(&H->h_c, 16 * sizeof(unsigned long long), hipHostMallocDefault));
                if (!H->ev_done)
                    SPM_HIP_CHECK(ctx, hipEventCreateWithFlags(&H->ev_done, hipEventDisableTiming));
                // (the copy itself is enqueued by whoever asks first: spm_hip_hits_copy_fused_device lets its kernel write the
                // counters to the pinned block -- no launch of its own --, anything else enqueues it in spm_complete_deferred)
                H->pending = true;
                H->c_on_the_way = false;
                H->d_count_cleared = false;
                // Sets that go through the band table: the host cannot know yet whether this scan gave every slot back.
                // It assumes so; if the band list or the table overflowed, the device remembers (resolve_params::
                // table_poison), later scans declare themselves void until the host -- completing this one -- has emptied
                // the table, and are repeated when they are completed in turn.
                ctx->band_dirty = false;
                H->d_text = text;
                H->d_patterns = patterns;
                H->d_begin = begin;
                H->d_end = end;
                H->d_opts = opts;
                *out = H.release();
                return SPM_OK;
            }
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
            const bool more_surv = c[1] > H->cand_cap && H->cand_cap < (1ull << 27) && !A.tune.cand_cap;
            const bool more_bands = c[2] != 0 && c[3] > H->band_cap && H->band_cap < (1ull << 28);
            const bool more_seen = c[2] != 0 && !A.seen_full && !more_bands && c[3] <= H->band_cap;
            if (!more_surv && !more_bands && !more_seen)
                break;
            if (spm_trace_on())
                fprintf(stderr, "[spm_hip] scan attempt %d starts over:%s%s%s (survivor slots drawn %llu of %llu, band slots %llu of %llu, "
                                "overflow flags %llu, spans that gave up %llu)\n",
                        attempt, more_surv ? " survivor list too small" : "", more_bands ? " band list too small" : "",
                        more_seen ? " dedupe set too small" : "", c[1], (unsigned long long)H->cand_cap, c[3],
                        (unsigned long long)H->band_cap, c[2], c[6]);
            if (more_surv)
                A.cand_cap_override = std::min<uint64_t>(1ull << 27, std::max<uint64_t>(c[1] + c[1] / 8 + 4096, 4 * H->cand_cap));
            if (more_bands)
                A.band_scale = std::max<uint64_t>(1, A.band_scale) * std::max<uint64_t>(2, (c[3] + H->band_cap - 1) / H->band_cap + 1);
            if (more_seen || ((more_surv || more_bands) && c[0] > ((uint64_t)A.seen_mask + 1) / 8))
                A.seen_full = true; // (an attempt cut short by its lists that nearly filled the set: the full one will not fit)
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
        } else if (c[6] != 0 && A.seen_skipped) {
            // spans gave up in a scan that ran without the dedupe set: once more, with it
            A.need_seen = true;
            again = true;
            H->stats.main_launches = 0;
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


// A deferred scan's counters have arrived (or are waited for here).  The usual case: nothing overflowed, no span gave up --
// the hit list is final.  Otherwise the scan is repeated the ordinary way (with its retries and fallbacks) and its result
// takes this one's place.
int spm_complete_deferred(spm_hits *h)
{
    if (!h->pending)
        return SPM_OK;
    spm_ctx *ctx = h->ctx;
    h->pending = false;
    if (!h->c_on_the_way) {
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(h->h_c, h->d_count, 13 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipEventRecord(h->ev_done, ctx->stream));
        h->c_on_the_way = true;
    }
    SPM_HIP_CHECK(ctx, hipEventSynchronize(h->ev_done));
    const unsigned long long *c = h->h_c;
    h->stats.n_candidates = c[5];
    h->stats.n_bands = (uint32_t)std::min<unsigned long long>(c[7], 0xFFFFFFFFull);
    const bool clean = c[2] == 0 && c[6] == 0 && c[1] <= h->cand_cap;
    if (c[2] != 0)
        ctx->band_dirty = true; // (a list or the table overflowed -- or the scan found the table poisoned: empty it next)
    if (clean || (c[0] > h->cap && c[2] == 0)) { // (more hits than the buffer takes is the caller's overflow, not a reason to scan again)
        if (clean) {
            const spm_patterns *ps = h->d_patterns;
            ps->cand_hint = std::max<uint64_t>(ps->cand_hint, c[1]);
            ps->hit_hint = std::max<uint64_t>(ps->hit_hint, c[0]);
            ps->band_hint = std::max<uint64_t>(ps->band_hint, c[3]);
            ps->scanned = true;
        }
        h->n = c[0];
        h->counted = true;
        return SPM_OK;
    }
    spm_scan_opts o = h->d_opts;
    o.flags &= ~SPM_SCAN_DEFER;
    spm_hits *again = nullptr;
    const int rc = scan_impl(ctx, h->d_text, h->d_begin, h->d_end, h->d_patterns, &o, nullptr, nullptr, nullptr, 0, &again);
    if (rc != SPM_OK)
        return rc;
    std::swap(h->d_hits, again->d_hits);
    std::swap(h->d_count, again->d_count);
    std::swap(h->cap, again->cap);
    for (int e = 0; e < 6; ++e)
        std::swap(h->ev[e], again->ev[e]);
    std::swap(h->d_aux[0], again->d_aux[0]);
    std::swap(h->d_aux[1], again->d_aux[1]);
    h->n = again->n;
    h->counted = again->counted;
    h->stats = again->stats;
    h->cand_cap = again->cand_cap;
    h->band_cap = again->band_cap;
    h->timed = again->timed;
    h->sorted_host = false;
    h->host.clear();
    again->d_count_cleared = h->d_count_cleared; // (the flags follow the buffers they describe)
    h->d_count_cleared = false;
    spm_hip_hits_destroy(again);
    return SPM_OK;
}
