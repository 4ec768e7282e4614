// libspm/matcher/restorable_base.hpp -- shared part of the restorable matchers: explicit state that survives
// across haystack chunks, capture() / restore().
//
// Reference semantics (/root/reference/libspm/libspm/matcher/myers_matcher_restorable.hpp:35-82,
// shiftor_matcher_restorable.hpp:35-67): the pattern state is initialised ONCE, in the constructor; every
// operator() call continues from the current state; capture() returns it, restore(state) overwrites it; states are
// std::semiregular (matcher/concept.hpp:132-148).  The reference's scan is PAUSED while a callback runs, so
//   * capture() inside a callback returns the state AT THAT HIT: here the bulk scan has already finished, so that state
//     is computed on the device from the last state captured in this call (or the entry state) over the symbols in
//     between -- the chunk stays resident in HBM, successive captures cost O(chunk) in total, not O(hits x chunk);
//   * restore(state) inside a callback makes the rest of the chunk continue from `state`: the hits still pending from
//     the bulk scan are dropped and the remainder [end of this hit, end of chunk) is scanned again from it.
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
    // replay context, valid only while the callbacks of one operator() call run
    mutable spm_text * _replay_text{nullptr};
    mutable std::size_t _replay_base{0}; // where the haystack of this call begins in _replay_text (a slice of a resident one)
    mutable matcher_state _cap_state{}; // state after text[0, _cap_pos) ...
    mutable std::size_t _cap_pos{0};
    mutable std::size_t _replay_end{0}; // end of the hit whose callback is running
    bool _restore_pending{false};

    void init_state()
    {
        std::vector<std::uint8_t> b(spm_hip_patterns_state_stride(this->_patterns.get()));
        if (spm_hip_patterns_state_init(this->_patterns.get(), b.data()) != SPM_OK)
            hip::fatal("spm_hip_patterns_state_init", hip::default_context());
        _state = matcher_state{std::move(b)};
    }

    // the haystack (a chunk, usually) is text[base, base + n): uploaded by the caller of this function -- hip_pattern_base::
    // run() for a host range, nobody for a spm::hip::resident_haystack, whose chunks are slices of one resident sequence
    template <typename callback_t>
    void run_on(spm_text * text_handle, std::size_t base, std::size_t n, callback_t && callback) noexcept
    {
        n = static_cast<derived_t *>(this)->bound(n);
        if (this->_needle.empty() || text_handle == nullptr)
            return; // empty needle: nothing to find (myers_prefix_matcher_restorable.hpp:39,52)
        struct
        {
            spm_text * t;
            spm_text * get() const noexcept { return t; }
        } const text{text_handle};
        std::size_t from = 0;
        for (;;) {
            matcher_state const entry = _state;
            matcher_state out{std::vector<std::uint8_t>(entry.blob().size())};
            spm_hit const * rec = nullptr;
            std::uint64_t cnt = 0;
            hip::hits_ptr hits =
                this->scan_text(text.get(), base + from, base + n, entry.blob().data(), out.blob().data(), rec, cnt);
            _state = out; // state after the last symbol; what capture() returns once the call has finished
            _replay_text = text.get();
            _replay_base = base;
            _cap_state = entry;
            _cap_pos = from;
            bool restarted = false;
            for (std::uint64_t i = 0; i < cnt && !restarted; ++i) {
                finder const f = this->make_finder(rec[i], n, base);
                _replay_end = f.end_position();
                _restore_pending = false;
                callback(f);
                if (_restore_pending) { // the callback restored a state: the rest of the chunk continues from it
                    from = std::min(_replay_end, n);
                    restarted = true;
                }
            }
            _replay_text = nullptr;
            _restore_pending = false;
            if (!restarted)
                break;
        }
    }

    std::size_t bound(std::size_t n) const noexcept { return n; }

public:
    using state_type = matcher_state;

    state_type const & capture() const noexcept
    {
        // outside a callback, or right after restore() inside one: the live state -- what the reference's capture()
        // returns, its pattern state being the restored one from that point on (myers_matcher_restorable.hpp:55-61)
        if (_replay_text == nullptr || _restore_pending)
            return _state;
        // inside a callback: the state right after the hit's last symbol, continued from the previous capture
        if (_replay_end > _cap_pos) {
            matcher_state at{std::vector<std::uint8_t>(_cap_state.blob().size())};
            spm_hit const * rec = nullptr;
            std::uint64_t cnt = 0;
            hip::hits_ptr ignored =
                this->scan_text(_replay_text, _replay_base + _cap_pos, _replay_base + _replay_end, _cap_state.blob().data(),
                                at.blob().data(), rec, cnt);
            _cap_state = std::move(at);
            _cap_pos = _replay_end;
        }
        return _cap_state;
    }

    void restore(state_type state) noexcept
    {
        _state = std::move(state);
        if (_replay_text != nullptr)
            _restore_pending = true;
    }
};
} // namespace spm
