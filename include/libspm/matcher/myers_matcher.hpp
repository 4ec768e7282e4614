// libspm/matcher/myers_matcher.hpp -- spm::myers_matcher, semi-global approximate search with k errors.
// API of /root/reference/libspm/libspm/matcher/myers_matcher.hpp:19-60: explicit ctor (needle, max_error_count = 0),
// CTAD, window_size = |P| + k (:51-53).  The callback's finder gives seqan2::endPosition (exclusive end) and, as an
// extension, finder.errors().
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class myers_matcher : public hip_pattern_base<myers_matcher<needle_t>>
{
    using base_t = hip_pattern_base<myers_matcher<needle_t>>;
    friend base_t;
    static constexpr bool reports_begin = false;

public:
    myers_matcher() = delete;
    template <std::ranges::viewable_range _needle_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, myers_matcher>)
    explicit myers_matcher(_needle_t && needle, std::size_t max_error_count = 0)
    {
        this->compile(needle, SPM_ALGO_MYERS, static_cast<std::uint32_t>(max_error_count));
    }

private:
    constexpr friend std::size_t tag_invoke(std::tag_t<window_size>, myers_matcher const & me) noexcept
    {
        return spm::window_size(static_cast<base_t const &>(me)) + me._errors;
    }
};

template <std::ranges::viewable_range needle_t>
myers_matcher(needle_t &&) -> myers_matcher<std::views::all_t<needle_t>>;

template <std::ranges::viewable_range needle_t>
myers_matcher(needle_t &&, std::size_t) -> myers_matcher<std::views::all_t<needle_t>>;
} // namespace spm
