// libspm/matcher/hip_batch.hpp -- many needles, one scan: the batch front-end over the C ABI.
//
// The reference's usage model is one matcher object per needle and one full pass per matcher
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-52).  On the MI355X the throughput is in scanning
// the haystack ONCE for the whole needle set, so this header adds the batch spelling of the same contract:
//   spm::batch_myers_matcher{needles, k}(haystack, callback)    callback(needle_index, finder)
// with the finder the single-needle matchers hand out (seqan2::endPosition / beginPosition work on it), hits delivered
// needle by needle in ascending position -- i.e. exactly the callbacks the per-needle matchers would have produced,
// grouped by needle.  spm::window_size(batch) is the largest window of the set.
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
template <int algo_v, bool reports_begin_v>
class batch_matcher
{
    std::vector<std::uint32_t> _lengths{};
    std::vector<std::uint16_t> _errors{};
    std::uint32_t _sigma{4};
    hip::patterns_ptr _patterns{};

public:
    batch_matcher() = delete;

    // needles: a range of needle ranges; one max_error_count for all, or one per needle
    template <std::ranges::input_range needles_t>
        requires std::ranges::input_range<std::ranges::range_reference_t<needles_t>>
    explicit batch_matcher(needles_t && needles, std::size_t max_error_count = 0)
    {
        std::vector<std::uint16_t> ks;
        for ([[maybe_unused]] auto && n : needles)
            ks.push_back(static_cast<std::uint16_t>(max_error_count));
        build(needles, ks);
    }
    template <std::ranges::input_range needles_t>
        requires std::ranges::input_range<std::ranges::range_reference_t<needles_t>>
    batch_matcher(needles_t && needles, std::vector<std::uint16_t> max_error_counts)
    {
        build(needles, max_error_counts);
    }

    std::size_t size() const noexcept { return _lengths.size(); }
    bool filterable() const noexcept { return spm_hip_patterns_filterable(_patterns.get()) != 0; }

    template <std::ranges::viewable_range haystack_t, typename callback_t>
    void operator()(haystack_t && haystack, callback_t && callback) noexcept
    {
        if (_lengths.empty())
            return;
        std::vector<std::uint8_t> owned;
        std::uint8_t const * ranks;
        std::size_t n;
        if constexpr (detail::byte_contiguous<haystack_t>) {
            ranks = reinterpret_cast<std::uint8_t const *>(std::ranges::data(haystack));
            n = std::ranges::size(haystack);
        } else {
            owned = detail::to_ranks(haystack);
            ranks = owned.data();
            n = owned.size();
        }
        spm_ctx * ctx = hip::default_context();
        spm_text * t = nullptr;
        if (spm_hip_text_upload(ctx, ranks, n, _sigma, &t) != SPM_OK)
            hip::fatal("spm_hip_text_upload", ctx);
        hip::text_ptr text{t};
        run_on(text.get(), 0, n, callback);
    }

    // the whole needle set against a haystack (or a slice of one) that is resident in HBM: no upload, one scan
    template <typename callback_t>
    void operator()(hip::resident_haystack const & haystack, callback_t && callback) noexcept
    {
        if (_lengths.empty() || haystack.empty())
            return;
        if (haystack.sigma() != _sigma) {
            std::fprintf(stderr, "libspm (MI355X back-end): the resident haystack's alphabet (%u symbols) is not the needles' (%u)\n",
                         haystack.sigma(), _sigma);
            std::abort();
        }
        run_on(haystack.text(), haystack.begin_offset(), haystack.size(), callback);
    }

private:
    template <typename callback_t>
    void run_on(spm_text * text, std::size_t base, std::size_t n, callback_t && callback) noexcept
    {
        spm_ctx * ctx = hip::default_context();
        spm_hit const * rec = nullptr;
        std::uint64_t cnt = 0;
        hip::hits_ptr hits = hip::scan_all_hits(
            ctx, spm_scan_opts{},
            [&](spm_scan_opts const & o, spm_hits ** h) {
                return spm_hip_scan(ctx, text, base, base + n, _patterns.get(), &o, nullptr, nullptr, h);
            },
            rec, cnt, "spm_hip_scan");
        for (std::uint64_t i = 0; i < cnt; ++i) {
            std::size_t const m = _lengths[rec[i].pattern];
            std::size_t const pos = static_cast<std::size_t>(rec[i].pos) - base;
            finder f = reports_begin_v ? finder{pos, pos + m, n, 0} : finder{pos >= m ? pos - m : 0, pos, n, rec[i].score};
            callback(static_cast<std::size_t>(rec[i].pattern), f);
        }
    }

private:
    template <typename needles_t>
    void build(needles_t && needles, std::vector<std::uint16_t> const & ks)
    {
        std::vector<std::uint8_t> cat;
        std::vector<std::uint32_t> offsets{0};
        for (auto && n : needles) {
            using symbol_t = std::ranges::range_value_t<decltype(n)>;
            _sigma = detail::sigma_of<symbol_t>();
            auto r = detail::to_ranks(n);
            cat.insert(cat.end(), r.begin(), r.end());
            offsets.push_back(static_cast<std::uint32_t>(cat.size()));
            _lengths.push_back(static_cast<std::uint32_t>(r.size()));
        }
        _errors = ks;
        _errors.resize(_lengths.size(), 0);
        spm_patterns * p = nullptr;
        std::uint8_t const dummy = 0;
        if (spm_hip_patterns_create(hip::default_context(), algo_v, cat.empty() ? &dummy : cat.data(), offsets.data(),
                                    static_cast<std::uint32_t>(_lengths.size()), _errors.data(), _sigma, &p) != SPM_OK)
            hip::fatal("spm_hip_patterns_create", hip::default_context());
        _patterns = hip::patterns_ptr{p, hip::patterns_deleter{}};
    }

    constexpr friend std::size_t tag_invoke(std::tag_t<window_size>, batch_matcher const & me) noexcept
    {
        std::size_t w = 0;
        for (std::size_t i = 0; i < me._lengths.size(); ++i)
            w = std::max<std::size_t>(w, me._lengths[i] + (reports_begin_v ? 0 : me._errors[i]));
        return w;
    }
};

using batch_myers_matcher = batch_matcher<SPM_ALGO_MYERS, false>;
using batch_shiftor_matcher = batch_matcher<SPM_ALGO_SHIFTOR, true>;
using batch_horspool_matcher = batch_matcher<SPM_ALGO_HORSPOOL, true>;
} // namespace spm
