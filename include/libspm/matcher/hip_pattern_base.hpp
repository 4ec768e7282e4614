// libspm/matcher/hip_pattern_base.hpp -- CRTP driver shared by all matchers.
//
// Mirrors /root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:27-100: operator()(haystack, callback) adapts
// the haystack, runs the search and invokes callback(finder) once per hit in ascending position; empty(); default
// window_size = |needle| (0 when empty, :97-99).  Instead of driving seqan2::find symbol by symbol on the host it
// hands the haystack (1-byte ranks) to libspm_hip.so, which scans it in bulk on the MI355X, and then replays the hits
// in order.  `finder` is what the callback receives; seqan2::beginPosition / endPosition / position / length are
// provided for it so that call sites written against the reference compile unchanged
// (test/api/libspm/matcher/myers_matcher_test.cpp:49-51, horspool_matcher_test.cpp:48-50).
#pragma once

#include <algorithm>
#include <concepts>
#include <cstring>
#include <ranges>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include <libspm/hip/context.hpp>
#include <libspm/seqan/container_adapter.hpp>
#include <libspm/matcher/concept.hpp>
#include <libspm/seqan/alphabet.hpp>

namespace spm
{
// What the per-hit callback receives (stands in for seqan2::Finder<haystack>).
class finder
{
    std::size_t _begin{}, _end{}, _haystack_length{};
    int _score{};

public:
    constexpr finder() = default;
    constexpr finder(std::size_t b, std::size_t e, std::size_t n, int score) noexcept :
        _begin{b}, _end{e}, _haystack_length{n}, _score{score}
    {}
    constexpr std::size_t begin_position() const noexcept { return _begin; }
    constexpr std::size_t end_position() const noexcept { return _end; }
    constexpr std::size_t haystack_length() const noexcept { return _haystack_length; }
    // edit distance of this hit (0 for exact matchers); the reference exposes it only as the pattern state's errors
    constexpr int score() const noexcept { return -_score; }
    constexpr int errors() const noexcept { return _score; }
};

namespace detail
{
    template <typename symbol_t>
    constexpr std::uint32_t sigma_of() noexcept
    {
        if constexpr (requires { std::remove_cvref_t<symbol_t>::alphabet_size; })
            return static_cast<std::uint32_t>(std::remove_cvref_t<symbol_t>::alphabet_size);
        else
            return 255; // plain integral symbols: ranks must be < 255
    }

    template <std::ranges::input_range range_t>
    std::vector<std::uint8_t> to_ranks(range_t && r)
    {
        std::vector<std::uint8_t> v;
        if constexpr (std::ranges::sized_range<range_t>)
            v.reserve(std::ranges::size(r));
        for (auto && s : r)
            v.push_back(rank_byte(s));
        return v;
    }
} // namespace detail

namespace hip
{
// A haystack that stays in HBM between calls.  The reference's call sites hand the matcher a host range every time
// (seqan_pattern_base.hpp:40-47) -- here that is one upload per call; a traverser that scans the same sequence with many
// matchers, or chunk by chunk with a restorable one, uploads it ONCE and passes this object instead:
//     spm::hip::resident_haystack hs{sequence};      matcher(hs, callback);      matcher(hs.slice(b, e), callback);
// Copies and slices share the device buffer; a slice is a haystack of its own (positions count from its first symbol,
// nothing before it is seen -- exactly what passing the sub-range as a host view would do).
class resident_haystack
{
    std::shared_ptr<spm_text> _text{};
    std::size_t _begin{0}, _end{0};
    std::uint32_t _sigma{4};

public:
    resident_haystack() = default;
    template <std::ranges::input_range range_t>
        requires(!std::same_as<std::remove_cvref_t<range_t>, resident_haystack>)
    explicit resident_haystack(range_t && sequence)
    {
        _sigma = detail::sigma_of<std::ranges::range_value_t<range_t>>();
        std::vector<std::uint8_t> const ranks = detail::to_ranks(sequence);
        spm_text * t = nullptr;
        std::uint8_t const dummy = 0;
        if (spm_hip_text_upload(default_context(), ranks.empty() ? &dummy : ranks.data(), ranks.size(), _sigma, &t) != SPM_OK)
            fatal("spm_hip_text_upload", default_context());
        _text = std::shared_ptr<spm_text>(t, text_deleter{});
        _end = ranks.size();
    }
    // ranks that already live in HBM (one byte per symbol, 16-byte aligned; validated once): borrowed, not copied
    static resident_haystack wrap(void const * device_ranks, std::size_t n, std::uint32_t sigma)
    {
        resident_haystack h;
        spm_text * t = nullptr;
        if (spm_hip_text_wrap(default_context(), device_ranks, n, sigma, &t) != SPM_OK)
            fatal("spm_hip_text_wrap", default_context());
        h._text = std::shared_ptr<spm_text>(t, text_deleter{});
        h._end = n;
        h._sigma = sigma;
        return h;
    }
    resident_haystack slice(std::size_t begin, std::size_t end) const noexcept
    {
        resident_haystack h = *this;
        h._begin = std::min(_begin + begin, _end);
        h._end = std::min(_begin + std::max(begin, end), _end);
        return h;
    }
    std::size_t size() const noexcept { return _end - _begin; }
    bool empty() const noexcept { return _end == _begin; }
    std::uint32_t sigma() const noexcept { return _sigma; }
    spm_text * text() const noexcept { return _text.get(); }
    std::size_t begin_offset() const noexcept { return _begin; }
    std::size_t end_offset() const noexcept { return _end; }
};
} // namespace hip

template <typename derived_t>
class hip_pattern_base
{
    friend derived_t;

protected:
    hip_pattern_base() = default;
    std::vector<std::uint8_t> _needle{}; // owned copy of the needle ranks (safe superset of the reference's view)
    std::uint32_t _sigma{4};
    std::uint32_t _errors{0};
    hip::patterns_ptr _patterns{};

    template <std::ranges::input_range needle_t>
    void compile(needle_t && needle, int algo, std::uint32_t errors)
    {
        _needle = detail::to_ranks(needle);
        _sigma = detail::sigma_of<std::ranges::range_value_t<needle_t>>();
        _errors = errors;
        std::uint32_t const offsets[2] = {0, static_cast<std::uint32_t>(_needle.size())};
        std::uint16_t const k = static_cast<std::uint16_t>(std::min<std::uint32_t>(errors, 0xFFFF));
        spm_patterns * p = nullptr;
        std::uint8_t const dummy = 0;
        if (spm_hip_patterns_create(hip::default_context(), algo, _needle.empty() ? &dummy : _needle.data(), offsets,
                                    1, &k, _sigma, &p) != SPM_OK)
            hip::fatal("spm_hip_patterns_create", hip::default_context());
        _patterns = hip::patterns_ptr{p, hip::patterns_deleter{}};
    }

    // the haystack, resident in HBM for the duration of one operator() call
    hip::text_ptr upload(std::uint8_t const * ranks, std::size_t n) const noexcept
    {
        spm_ctx * ctx = hip::default_context();
        spm_text * t = nullptr;
        if (spm_hip_text_upload(ctx, ranks, n, _sigma, &t) != SPM_OK)
            hip::fatal("spm_hip_text_upload", ctx);
        return hip::text_ptr{t};
    }

    // one bulk scan of text[begin, end); the hits (positions relative to text[0]) in callback order.  A hit buffer that
    // proves too small is enlarged and the scan repeated (hip::scan_all_hits): the reference's find loop has no limit
    hip::hits_ptr scan_text(spm_text * text, std::size_t begin, std::size_t end, void const * state_in, void * state_out,
                            spm_hit const *& rec, std::uint64_t & cnt) const noexcept
    {
        spm_ctx * ctx = hip::default_context();
        spm_scan_opts opts{};
        opts.engine = SPM_ENGINE_AUTO;
        return hip::scan_all_hits(
            ctx, opts,
            [&](spm_scan_opts const & o, spm_hits ** h) {
                return spm_hip_scan(ctx, text, begin, end, _patterns.get(), &o, state_in, state_out, h);
            },
            rec, cnt, "spm_hip_scan");
    }

public:
    // Note: non-const like the reference ("seqan use non-const pattern", seqan_pattern_base.hpp:39-41).
    template <std::ranges::viewable_range haystack_t, typename callback_t>
    void operator()(haystack_t && haystack, callback_t && callback) noexcept
    {
        // the adapter lends the view its rank buffer: zero-copy for contiguous 1-byte symbols, staged otherwise
        // (the reference wraps the same view for seqan2::Finder, seqan_pattern_base.hpp:43-47)
        if constexpr (requires { make_seqan_container(std::views::all(std::forward<haystack_t>(haystack))); }) {
            auto const adapted = make_seqan_container(std::views::all(std::forward<haystack_t>(haystack)));
            auto const ranks = adapted.ranks();
            static_cast<derived_t *>(this)->run(ranks.data(), ranks.size(), callback);
        } else {
            std::vector<std::uint8_t> const ranks = detail::to_ranks(haystack);
            static_cast<derived_t *>(this)->run(ranks.data(), ranks.size(), callback);
        }
    }

    // the same call on a haystack (or a slice of one) that is resident in HBM: no upload
    template <typename callback_t>
    void operator()(hip::resident_haystack const & haystack, callback_t && callback) noexcept
    {
        if (haystack.sigma() != _sigma && !_needle.empty()) {
            std::fprintf(stderr, "libspm (MI355X back-end): the resident haystack's alphabet (%u symbols) is not the needle's (%u)\n",
                         haystack.sigma(), _sigma);
            std::abort();
        }
        static_cast<derived_t *>(this)->run_on(haystack.text(), haystack.begin_offset(), haystack.size(), callback);
    }

    bool empty() const noexcept { return _needle.empty(); }

protected:
    template <typename callback_t>
    void run(std::uint8_t const * ranks, std::size_t n, callback_t && callback) noexcept
    {
        hip::text_ptr text = upload(ranks, n);
        static_cast<derived_t *>(this)->run_on(text.get(), 0, n, callback);
    }

    // default run_on(): fresh matcher every call (a fresh seqan2::Finder re-initialises the pattern); the haystack is
    // text[base, base + n)
    template <typename callback_t>
    void run_on(spm_text * text, std::size_t base, std::size_t n, callback_t && callback) noexcept
    {
        if (text == nullptr || n == 0)
            return;
        spm_hit const * rec = nullptr;
        std::uint64_t cnt = 0;
        hip::hits_ptr hits = scan_text(text, base, base + n, nullptr, nullptr, rec, cnt);
        for (std::uint64_t i = 0; i < cnt; ++i)
            callback(make_finder(rec[i], n, base));
    }

    // (hit positions arrive relative to text[0]; the finder speaks in haystack coordinates)
    finder make_finder(spm_hit const & h, std::size_t n, std::size_t base = 0) const noexcept
    {
        std::size_t const m = _needle.size();
        std::size_t const pos = static_cast<std::size_t>(h.pos) - base;
        return derived_t::reports_begin ? finder{pos, pos + m, n, 0} : finder{pos >= m ? pos - m : 0, pos, n, h.score};
    }

private:
    constexpr friend std::size_t tag_invoke(std::tag_t<spm::window_size>, hip_pattern_base const & me) noexcept
    {
        return me._needle.size();
    }
};
} // namespace spm

// The names the reference's call sites use on the finder.
namespace seqan2
{
inline std::size_t beginPosition(spm::finder const & f) noexcept { return f.begin_position(); }
inline std::size_t endPosition(spm::finder const & f) noexcept { return f.end_position(); }
inline std::size_t position(spm::finder const & f) noexcept { return f.begin_position(); }
inline std::size_t length(spm::finder const & f) noexcept { return f.end_position() - f.begin_position(); }
} // namespace seqan2
