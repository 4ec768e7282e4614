// libspm/matcher/pigeonhole_matcher.hpp -- spm::pigeonhole_matcher: multi-needle q-gram SEED filter (seed hits only).
//
// API of /root/reference/libspm/libspm/matcher/pigeonhole_matcher.hpp:129-259: construct from one needle or from a
// range of needles plus an error rate; operator()(haystack, callback) fires once per seed hit in ascending haystack
// position; seqan2::beginPosition(finder) is the hit's begin in the haystack; matcher.position() is the
// (needle index, offset in the needle, seed length) of the current hit (PigeonholeSeedOnlyPosition, :39-54);
// window_size = q (:197-199... the q-gram shape length).
//
// Seed layout ([upstream] SeqAn pigeonhole, seed-only): needle i tolerates e_i = floor(error_rate * |needle_i|) errors
// and would be cut into e_i + 1 seeds; the shared seed length is q = min_i floor(|needle_i| / (e_i + 1)); every needle
// contributes the non-overlapping q-grams at offsets 0, q, 2q, ...  A seed hit is an exact occurrence of such a
// q-gram.  Pinned by the reference only for error_rate = 0 (test/api/libspm/matcher/pigeonhole_matcher_test.cpp:30-33,
// 54-85); other rates follow the rule above and are PARITY UNPINNED.
//
// Back-end: the q-grams of all needles form one exact needle set for libspm_hip.so (seed filter engine when
// q >= 16, one-lane-per-q-gram Shift-Or otherwise).
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace seqan2
{
struct PigeonholeSeedOnlyPosition
{
    std::ptrdiff_t index{};  // needle          (signed, as in the reference: pigeonhole_matcher.hpp:39-42)
    std::ptrdiff_t offset{}; // begin of the seed inside the needle
    std::ptrdiff_t count{};  // seed length
    constexpr bool operator==(PigeonholeSeedOnlyPosition const &) const noexcept = default;

    // "<needle, offset, count>", the format the reference's tests print on a mismatch (pigeonhole_matcher.hpp:48-53)
    template <typename stream_t, typename me_t>
        requires std::same_as<std::remove_cvref_t<me_t>, PigeonholeSeedOnlyPosition>
    friend stream_t & operator<<(stream_t & stream, me_t && me)
    {
        stream << "<" << me.index << ", " << me.offset << ", " << me.count << ">";
        return stream;
    }
};
} // namespace seqan2

namespace spm
{
template <std::ranges::random_access_range needle_t>
class pigeonhole_matcher
{
    struct seed
    {
        std::size_t index, offset;
    };
    std::vector<seed> _seeds{};
    std::size_t _q{0};
    double _error_rate{};
    std::uint32_t _sigma{4};
    hip::patterns_ptr _patterns{};
    seqan2::PigeonholeSeedOnlyPosition _position{};

    void build(std::vector<std::vector<std::uint8_t>> const & needles)
    {
        std::size_t q = 0;
        bool first = true;
        for (auto const & n : needles) {
            if (n.empty())
                continue;
            std::size_t const e = static_cast<std::size_t>(_error_rate * static_cast<double>(n.size()));
            std::size_t const qi = n.size() / (e + 1);
            q = first ? qi : std::min(q, qi);
            first = false;
        }
        _q = q;
        std::vector<std::uint8_t> cat;
        std::vector<std::uint32_t> offsets{0};
        if (q > 0)
            for (std::size_t i = 0; i < needles.size(); ++i)
                for (std::size_t o = 0; o + q <= needles[i].size(); o += q) {
                    cat.insert(cat.end(), needles[i].begin() + o, needles[i].begin() + o + q);
                    offsets.push_back(static_cast<std::uint32_t>(cat.size()));
                    _seeds.push_back({i, o});
                }
        spm_patterns * p = nullptr;
        std::uint8_t const dummy = 0;
        if (spm_hip_patterns_create(hip::default_context(), SPM_ALGO_SHIFTOR, cat.empty() ? &dummy : cat.data(),
                                    offsets.data(), static_cast<std::uint32_t>(_seeds.size()), nullptr, _sigma,
                                    &p) != SPM_OK)
            hip::fatal("spm_hip_patterns_create", hip::default_context());
        _patterns = hip::patterns_ptr{p, hip::patterns_deleter{}};
    }

public:
    pigeonhole_matcher() = delete;

    // one needle
    template <std::ranges::viewable_range _needle_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, pigeonhole_matcher> &&
                 !std::ranges::range<std::ranges::range_value_t<_needle_t>>)
    explicit pigeonhole_matcher(_needle_t && needle, double error_rate = 0.0) : _error_rate{error_rate}
    {
        _sigma = detail::sigma_of<std::ranges::range_value_t<_needle_t>>();
        build({detail::to_ranks(needle)});
    }

    // many needles
    template <std::ranges::viewable_range _multi_needle_t>
        requires(!std::same_as<std::remove_cvref_t<_multi_needle_t>, pigeonhole_matcher> &&
                 std::ranges::range<std::ranges::range_value_t<_multi_needle_t>>)
    explicit pigeonhole_matcher(_multi_needle_t && multi_needle, double error_rate = 0.0) : _error_rate{error_rate}
    {
        using inner_t = std::ranges::range_value_t<_multi_needle_t>;
        _sigma = detail::sigma_of<std::ranges::range_value_t<inner_t>>();
        std::vector<std::vector<std::uint8_t>> needles;
        for (auto && n : multi_needle)
            needles.push_back(detail::to_ranks(n));
        build(needles);
    }

    template <std::ranges::viewable_range haystack_t, typename callback_t>
    void operator()(haystack_t && haystack, callback_t && callback) noexcept
    {
        if (_seeds.empty())
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

    // the same on a haystack (or a slice of one) that is resident in HBM (hip_pattern_base.hpp): no upload
    template <typename callback_t>
    void operator()(hip::resident_haystack const & haystack, callback_t && callback) noexcept
    {
        if (_seeds.empty() || haystack.empty())
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
        // seed hits in ascending haystack position (ties: needle order, then offset)
        std::vector<spm_hit> order(rec, rec + cnt);
        std::stable_sort(order.begin(), order.end(), [](spm_hit const & a, spm_hit const & b) { return a.pos < b.pos; });
        for (spm_hit const & x : order) {
            _position = {static_cast<std::ptrdiff_t>(_seeds[x.pattern].index), static_cast<std::ptrdiff_t>(_seeds[x.pattern].offset),
                         static_cast<std::ptrdiff_t>(_q)};
            std::size_t const pos = static_cast<std::size_t>(x.pos) - base;
            finder f{pos, pos + _q, n, 0};
            callback(f);
        }
    }

public:

    constexpr auto position() const noexcept { return _position; }
    bool empty() const noexcept { return _seeds.empty(); }

private:
    constexpr friend std::size_t tag_invoke(std::tag_t<window_size>, pigeonhole_matcher const & me) noexcept
    {
        return me._q;
    }
};

template <std::ranges::viewable_range needle_t>
    requires(!std::ranges::range<std::ranges::range_value_t<needle_t>>)
pigeonhole_matcher(needle_t &&) -> pigeonhole_matcher<std::views::all_t<needle_t>>;

template <std::ranges::viewable_range needle_t>
    requires(!std::ranges::range<std::ranges::range_value_t<needle_t>>)
pigeonhole_matcher(needle_t &&, double) -> pigeonhole_matcher<std::views::all_t<needle_t>>;

template <std::ranges::viewable_range multi_needle_t>
    requires std::ranges::random_access_range<std::ranges::range_reference_t<multi_needle_t>>
pigeonhole_matcher(multi_needle_t &&)
    -> pigeonhole_matcher<std::views::all_t<std::ranges::range_reference_t<multi_needle_t>>>;

template <std::ranges::viewable_range multi_needle_t>
    requires std::ranges::random_access_range<std::ranges::range_reference_t<multi_needle_t>>
pigeonhole_matcher(multi_needle_t &&, double)
    -> pigeonhole_matcher<std::views::all_t<std::ranges::range_reference_t<multi_needle_t>>>;
} // namespace spm
