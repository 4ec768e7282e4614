// libspm MI355X back-end -- view -> rank-buffer adapter.
//
// Stands where the reference has libspm/seqan/container_adapter.hpp:23-100 (spm::seqan_container_adapter,
// make_seqan_container :92-97, seqan_container_t :99-100): there the adapter lends a std::ranges::view the container
// interface SeqAn2's Finder/Pattern expect.  Here the consumer of a haystack or needle is the C ABI
// (spm_hip_text_upload / spm_hip_patterns_create, include/spm_hip.h), which wants one rank byte per symbol, so the
// adapter additionally lends the view a `ranks()` buffer: zero-copy when the view already is contiguous 1-byte
// symbols (std::vector<spm::dna4>, alphabet.hpp), a staging copy otherwise (reversed, transformed, journaled views).
#pragma once

#include <cstddef>
#include <cstdint>
#include <iterator>
#include <optional>
#include <ranges>
#include <span>
#include <type_traits>
#include <utility>
#include <vector>

namespace spm
{
namespace detail
{
    // Contiguous ranges of 1-byte symbols are handed to the device without a host copy.
    template <typename range_t>
    concept byte_contiguous = std::ranges::contiguous_range<range_t> && std::ranges::sized_range<range_t> &&
                              sizeof(std::ranges::range_value_t<range_t>) == 1;

    template <typename symbol_t>
    constexpr std::uint8_t rank_byte(symbol_t const & s) noexcept
    {
        return static_cast<std::uint8_t>(static_cast<unsigned>(s));
    }
} // namespace detail

template <typename range_t>
class seqan_container_adapter
{
    std::optional<range_t> _view{};
    mutable std::vector<std::uint8_t> _staging{}; // filled on the first ranks() of a non-contiguous view
    mutable bool _staged{false};

    range_t & view() noexcept { return *_view; }
    range_t const & view() const noexcept { return *_view; }

public:
    using value_type = std::ranges::range_value_t<range_t>;
    using reference = std::ranges::range_reference_t<range_t>;
    using iterator = std::ranges::iterator_t<range_t>;
    using const_reference = std::ranges::range_reference_t<std::remove_const_t<range_t> const>; // (container_adapter.hpp:32-34)
    using const_iterator = std::ranges::iterator_t<std::remove_const_t<range_t> const>;
    using difference_type = std::ranges::range_difference_t<range_t>;
    using size_type = std::make_unsigned_t<difference_type>;

    seqan_container_adapter() = default;
    explicit seqan_container_adapter(range_t v) noexcept(std::is_nothrow_move_constructible_v<range_t>) :
        _view{std::in_place, std::move(v)}
    {}

    bool has_view() const noexcept { return _view.has_value(); }

    constexpr iterator begin() { return std::ranges::begin(view()); }
    constexpr auto begin() const
        requires std::ranges::range<range_t const>
    {
        return std::ranges::begin(view());
    }
    constexpr auto end() { return std::ranges::end(view()); }
    constexpr auto end() const
        requires std::ranges::range<range_t const>
    {
        return std::ranges::end(view());
    }

    constexpr reference operator[](difference_type i)
        requires std::ranges::random_access_range<range_t>
    {
        return std::ranges::begin(view())[i];
    }
    constexpr decltype(auto) operator[](difference_type i) const
        requires std::ranges::random_access_range<range_t const>
    {
        return std::ranges::begin(view())[i];
    }

    constexpr size_type size() const
    {
        if (!_view)
            return 0;
        if constexpr (std::ranges::sized_range<range_t const>)
            return static_cast<size_type>(std::ranges::size(view()));
        else {
            auto & v = const_cast<range_t &>(view());
            return static_cast<size_type>(std::ranges::distance(std::ranges::begin(v), std::ranges::end(v)));
        }
    }
    constexpr bool empty() const { return size() == 0; }

    // One rank byte per symbol, in the layout spm_hip_text_upload takes.  The span stays valid as long as the adapter
    // (and, for the zero-copy case, the viewed storage) lives.
    std::span<std::uint8_t const> ranks() const
    {
        if (!_view)
            return {};
        if constexpr (detail::byte_contiguous<range_t const>) {
            return {reinterpret_cast<std::uint8_t const *>(std::ranges::data(view())), std::ranges::size(view())};
        } else {
            if (!_staged) {
                auto & v = const_cast<range_t &>(view());
                _staging.clear();
                if constexpr (std::ranges::sized_range<range_t>)
                    _staging.reserve(std::ranges::size(v));
                for (auto && s : v)
                    _staging.push_back(detail::rank_byte(s));
                _staged = true;
            }
            return {_staging.data(), _staging.size()};
        }
    }

    static constexpr bool zero_copy = detail::byte_contiguous<range_t const>;
};

// Same entry point and constraints as the reference (container_adapter.hpp:92-97).
template <std::ranges::view range_t>
    requires(std::ranges::common_range<range_t> && std::ranges::random_access_range<range_t>)
constexpr auto make_seqan_container(range_t v) noexcept(std::is_nothrow_move_constructible_v<range_t>)
{
    return seqan_container_adapter<range_t>{std::move(v)};
}

template <std::ranges::view range_t>
using seqan_container_t = decltype(make_seqan_container(std::declval<range_t>()));

template <typename range_t>
inline void assign(seqan_container_adapter<range_t> & target, seqan_container_adapter<range_t> const & source)
{
    target = source;
}
} // namespace spm

// SeqAn2-style free functions over the adapter (the reference specialises the seqan2 metafunctions,
// container_adapter.hpp:111-209; only the accessors have a meaning without SeqAn).
namespace seqan2
{
template <typename range_t>
constexpr auto length(spm::seqan_container_adapter<range_t> const & c)
{
    return c.size();
}
template <typename range_t>
constexpr bool empty(spm::seqan_container_adapter<range_t> const & c)
{
    return c.empty();
}
template <typename range_t>
constexpr auto begin(spm::seqan_container_adapter<range_t> & c)
{
    return c.begin();
}
template <typename range_t>
constexpr auto end(spm::seqan_container_adapter<range_t> & c)
{
    return c.end();
}
} // namespace seqan2
