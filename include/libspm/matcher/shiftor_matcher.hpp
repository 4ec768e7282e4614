// libspm/matcher/shiftor_matcher.hpp -- spm::shiftor_matcher, exact search.
// API of /root/reference/libspm/libspm/matcher/shiftor_matcher.hpp:20-44.
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class shiftor_matcher : public hip_pattern_base<shiftor_matcher<needle_t>>
{
    using base_t = hip_pattern_base<shiftor_matcher<needle_t>>;
    friend base_t;
    static constexpr bool reports_begin = true;

public:
    shiftor_matcher() = delete;
    template <std::ranges::viewable_range _needle_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, shiftor_matcher>)
    explicit shiftor_matcher(_needle_t && needle)
    {
        this->compile(needle, SPM_ALGO_SHIFTOR, 0);
    }
};

template <std::ranges::viewable_range needle_t>
shiftor_matcher(needle_t &&) -> shiftor_matcher<std::views::all_t<needle_t>>;
} // namespace spm
