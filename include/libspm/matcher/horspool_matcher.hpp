// libspm/matcher/horspool_matcher.hpp -- spm::horspool_matcher, exact search.
// API of /root/reference/libspm/libspm/matcher/horspool_matcher.hpp:20-44 (explicit ctor from a needle range, deleted
// default ctor, CTAD, window_size = |P|).  Horspool's data-dependent skips do not map to a wavefront; the device
// reports the same hit set -- the begin position of every occurrence, overlapping ones included -- through the
// multi-pattern exact engine (SPM_ALGO_HORSPOOL).
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class horspool_matcher : public hip_pattern_base<horspool_matcher<needle_t>>
{
    using base_t = hip_pattern_base<horspool_matcher<needle_t>>;
    friend base_t;
    static constexpr bool reports_begin = true;

public:
    horspool_matcher() = delete;
    template <std::ranges::viewable_range _needle_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, horspool_matcher>)
    explicit horspool_matcher(_needle_t && needle)
    {
        this->compile(needle, SPM_ALGO_HORSPOOL, 0);
    }
};

template <std::ranges::viewable_range needle_t>
horspool_matcher(needle_t &&) -> horspool_matcher<std::views::all_t<needle_t>>;
} // namespace spm
