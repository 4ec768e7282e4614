// libspm/matcher/restorable_base.hpp -- shared part of the restorable matchers: explicit state that survives
// across haystack chunks, capture() / restore().
//
// Reference semantics (/root/reference/libspm/libspm/matcher/myers_matcher_restorable.hpp:35-82,
// shiftor_matcher_restorable.hpp:35-67): the pattern state is initialised ONCE, in the constructor; every
// operator() call continues from the current state; capture() returns it, restore(state) overwrites it; states are
// std::semiregular (matcher/concept.hpp:132-148).  capture() called from inside the per-hit callback returns the
// state AT THAT HIT, because the reference's scan is paused there: here the bulk scan has already finished, so the
// state at the hit is recomputed on the device by scanning the chunk prefix [0, end of hit) from the entry state.
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
// POD-like matcher state: the ABI blob of include/spm_hip.h (one record).  Myers: score, VP words, VN words.
class matcher_state
{
    std::vector<std::uint8_t> _blob{};

public:
    matcher_state() = default;
    explicit matcher_state(std::vector<std::uint8_t> b) noexcept : _blob{std::move(b)} {}
    bool operator==(matcher_state const &) const noexcept = default;
    std::vector<std::uint8_t> & blob() noexcept { return _blob; }
    std::vector<std::uint8_t> const & blob() const noexcept { return _blob; }
    // Myers: edit distance of the last column (the reference's `errors`)
    int errors() const noexcept
    {
        int s = 0;
        if (_blob.size() >= 4)
            std::memcpy(&s, _blob.data(), 4);
        return s;
    }
};

template <typename derived_t>
class restorable_base : public hip_pattern_base<derived_t>
{
    using base_t = hip_pattern_base<derived_t>;
    friend base_t;
    friend derived_t;
    restorable_base() = default;

    matcher_state _state{};
    // replay context, valid only while callbacks of one operator() call run
    mutable matcher_state _state_at_hit{};
    mutable std::uint8_t const * _replay_ranks{nullptr};
    mutable matcher_state const * _entry_state{nullptr};
    mutable std::size_t _replay_end{0};

    void init_state()
    {
        std::vector<std::uint8_t> b(spm_hip_patterns_state_stride(this->_patterns.get()));
        if (spm_hip_patterns_state_init(this->_patterns.get(), b.data()) != SPM_OK)
            hip::fatal("spm_hip_patterns_state_init", hip::default_context());
        _state = matcher_state{std::move(b)};
    }

    template <typename callback_t>
    void run(std::uint8_t const * ranks, std::size_t n, callback_t && callback) noexcept
    {
        n = static_cast<derived_t *>(this)->bound(n);
        if (this->_needle.empty())
            return; // empty needle: nothing to find (myers_prefix_matcher_restorable.hpp:39,52)
        matcher_state const entry = _state;
        matcher_state out{std::vector<std::uint8_t>(entry.blob().size())};
        hip::hits_ptr hits = this->scan(ranks, n, entry.blob().data(), out.blob().data());
        _state = out; // state after the last symbol; what capture() returns once the call has finished
        _replay_ranks = ranks;
        _entry_state = &entry;
        this->replay(hits.get(), n, callback);
        _replay_ranks = nullptr;
        _entry_state = nullptr;
    }

    void on_hit(finder const & f) const noexcept { _replay_end = f.end_position(); }
    std::size_t bound(std::size_t n) const noexcept { return n; }

public:
    using state_type = matcher_state;

    state_type const & capture() const noexcept
    {
        if (_replay_ranks == nullptr)
            return _state;
        // inside a callback: the state right after the hit's last symbol
        matcher_state at{std::vector<std::uint8_t>(_entry_state->blob().size())};
        std::size_t const upto = derived_t::reports_begin ? _replay_end : _replay_end;
        hip::hits_ptr ignored = this->scan(_replay_ranks, upto, _entry_state->blob().data(), at.blob().data());
        _state_at_hit = std::move(at);
        return _state_at_hit;
    }

    void restore(state_type state) noexcept { _state = std::move(state); }
};
} // namespace spm
