// libspm/matcher/myers_matcher_restorable.hpp -- spm::restorable_myers_matcher.
// API of /root/reference/libspm/libspm/matcher/myers_matcher_restorable.hpp:110-156: ctor (needle, unsigned error
// count), capture() -> state const &, restore(state), window_size = |P| + k (:150-152).
#pragma once

#include <libspm/matcher/restorable_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class restorable_myers_matcher : public restorable_base<restorable_myers_matcher<needle_t>>
{
    using base_t = restorable_base<restorable_myers_matcher<needle_t>>;
    friend base_t;
    friend hip_pattern_base<restorable_myers_matcher<needle_t>>;
    static constexpr bool reports_begin = false;

public:
    using state_type = typename base_t::state_type;

    restorable_myers_matcher() = delete;
    template <std::ranges::viewable_range _needle_t, std::unsigned_integral error_count_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, restorable_myers_matcher>)
    explicit restorable_myers_matcher(_needle_t && needle, error_count_t const error_count)
    {
        this->compile(needle, SPM_ALGO_MYERS, static_cast<std::uint32_t>(error_count));
        this->init_state();
    }

private:
    constexpr friend std::size_t tag_invoke(std::tag_t<window_size>, restorable_myers_matcher const & me) noexcept
    {
        return me._needle.size() + me._errors;
    }
};

template <std::ranges::viewable_range needle_t, std::unsigned_integral error_count_t>
restorable_myers_matcher(needle_t &&, error_count_t) -> restorable_myers_matcher<std::views::all_t<needle_t>>;
} // namespace spm
