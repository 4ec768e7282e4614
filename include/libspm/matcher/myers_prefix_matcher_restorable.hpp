// libspm/matcher/myers_prefix_matcher_restorable.hpp -- spm::restorable_myers_prefix_matcher: the needle against the
// PREFIXES of the haystack (global start), used by a traverser to extend seeds.
// API and bounds of /root/reference/libspm/libspm/matcher/myers_prefix_matcher_restorable.hpp:117-163: at most
// min(|haystack|, |P| + k + 1) symbols are scanned (:54-55), an empty needle finds nothing (:39,:52),
// window_size = |P| + k (:157-159).  The reference has no test for this matcher; parity is pinned on the Sellers DP.
#pragma once

#include <libspm/matcher/restorable_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class restorable_myers_prefix_matcher : public restorable_base<restorable_myers_prefix_matcher<needle_t>>
{
    using base_t = restorable_base<restorable_myers_prefix_matcher<needle_t>>;
    friend base_t;
    friend hip_pattern_base<restorable_myers_prefix_matcher<needle_t>>;
    static constexpr bool reports_begin = false;

    std::size_t bound(std::size_t n) const noexcept
    {
        return std::min<std::size_t>(n, this->_needle.size() + this->_errors + 1);
    }

public:
    using state_type = typename base_t::state_type;

    restorable_myers_prefix_matcher() = delete;
    template <std::ranges::viewable_range _needle_t, std::unsigned_integral error_count_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, restorable_myers_prefix_matcher>)
    explicit restorable_myers_prefix_matcher(_needle_t && needle, error_count_t const error_count)
    {
        this->compile(needle, SPM_ALGO_MYERS_PREFIX, static_cast<std::uint32_t>(error_count));
        this->init_state();
    }

private:
    constexpr friend std::size_t tag_invoke(std::tag_t<window_size>,
                                            restorable_myers_prefix_matcher const & me) noexcept
    {
        return me._needle.empty() ? 0 : me._needle.size() + me._errors;
    }
};

template <std::ranges::viewable_range needle_t, std::unsigned_integral error_count_t>
restorable_myers_prefix_matcher(needle_t &&, error_count_t)
    -> restorable_myers_prefix_matcher<std::views::all_t<needle_t>>;
} // namespace spm
