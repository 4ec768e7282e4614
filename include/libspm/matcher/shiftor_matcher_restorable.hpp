// libspm/matcher/shiftor_matcher_restorable.hpp -- spm::restorable_shiftor_matcher: Shift-Or whose register R
// (SeqAn's prefSufMatch) is the captured state.
// API of /root/reference/libspm/libspm/matcher/shiftor_matcher_restorable.hpp:89-129.  A resumed scan continues
// after the previous chunk; an occurrence that straddles a chunk border is reported in the later chunk with a begin
// position relative to that chunk's first symbol (it can be "negative": positions are size_t, so such hits are reported
// through finder.begin_position() wrapped -- callers add their chunk offset, as the reference's chunked test does for
// Myers).  The reference has no test for this matcher; parity is pinned on naive search.
#pragma once

#include <libspm/matcher/restorable_base.hpp>

namespace spm
{
template <std::ranges::random_access_range needle_t>
class restorable_shiftor_matcher : public restorable_base<restorable_shiftor_matcher<needle_t>>
{
    using base_t = restorable_base<restorable_shiftor_matcher<needle_t>>;
    friend base_t;
    friend hip_pattern_base<restorable_shiftor_matcher<needle_t>>;
    static constexpr bool reports_begin = true;

public:
    using state_type = typename base_t::state_type;

    restorable_shiftor_matcher() = delete;
    template <std::ranges::viewable_range _needle_t>
        requires(!std::same_as<std::remove_cvref_t<_needle_t>, restorable_shiftor_matcher>)
    explicit restorable_shiftor_matcher(_needle_t && needle)
    {
        this->compile(needle, SPM_ALGO_SHIFTOR, 0);
        this->init_state();
    }
};

template <std::ranges::viewable_range needle_t>
restorable_shiftor_matcher(needle_t &&) -> restorable_shiftor_matcher<std::views::all_t<needle_t>>;
} // namespace spm
