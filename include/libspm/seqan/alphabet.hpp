// libspm/seqan/alphabet.hpp -- spm::dna4 / dna5 / dna15 symbols, one byte each, value = rank.
// Stands in for /root/reference/libspm/libspm/seqan/alphabet.hpp:34-151 (seqan2::alphabet_adaptor over the seqan3
// alphabets) without SeqAn: same size (1 byte, :37), same conversions (explicit from char / rank :41-52, implicit to
// any integral = rank :68-72, to char :74-77), ordering by rank (:62-65), alphabet sizes 4 / 5 / 15 (:100-105) and the
// ""_dna4 / ""_dna5 literals (:122-150).  Rank orders are seqan3's: dna4 ACGT, dna5 ACGNT, dna15 ABCDGHKMNRSTVWY.
#pragma once

#include <compare>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <string_view>
#include <vector>

namespace spm
{
namespace detail
{
    template <std::size_t N>
    struct alphabet_table
    {
        char chars[N + 1];
        std::uint8_t unknown; // rank that characters outside the alphabet map to
    };
    inline constexpr alphabet_table<4> dna4_table{"ACGT", 0};
    inline constexpr alphabet_table<5> dna5_table{"ACGNT", 3};
    inline constexpr alphabet_table<15> dna15_table{"ABCDGHKMNRSTVWY", 8};

    constexpr char upper(char c) noexcept { return (c >= 'a' && c <= 'z') ? char(c - 'a' + 'A') : c; }
} // namespace detail

template <std::size_t sigma, auto const & table>
struct nucleotide
{
    std::uint8_t _rank{};

    static constexpr std::size_t alphabet_size = sigma;

    constexpr nucleotide() = default;
    constexpr explicit nucleotide(char c) noexcept : _rank{char_to_rank(c)} {}
    template <std::integral rank_t>
        requires(!std::same_as<rank_t, char>)
    constexpr explicit nucleotide(rank_t r) noexcept : _rank{static_cast<std::uint8_t>(r)} {}

    static constexpr std::uint8_t char_to_rank(char c) noexcept
    {
        c = detail::upper(c);
        if (c == 'U')
            c = 'T';
        for (std::size_t r = 0; r < sigma; ++r)
            if (table.chars[r] == c)
                return static_cast<std::uint8_t>(r);
        if constexpr (sigma == 4) { // seqan3::dna4 folds IUPAC codes onto a member base
            switch (c) {
            case 'B': case 'S': case 'Y': case 'M': case 'H': case 'V': return 1;
            case 'K': return 2;
            default: return 0;
            }
        }
        return table.unknown;
    }

    constexpr std::uint8_t to_rank() const noexcept { return _rank; }
    constexpr char to_char() const noexcept { return table.chars[_rank]; }
    constexpr nucleotide & assign_rank(std::uint8_t r) noexcept { _rank = r; return *this; }
    constexpr nucleotide & assign_char(char c) noexcept { _rank = char_to_rank(c); return *this; }

    constexpr bool operator==(nucleotide const &) const noexcept = default;
    constexpr std::strong_ordering operator<=>(nucleotide const & o) const noexcept { return _rank <=> o._rank; }

    template <std::integral int_t>
        requires(!std::same_as<int_t, char>)
    constexpr operator int_t() const noexcept { return static_cast<int_t>(_rank); }
    constexpr operator char() const noexcept { return to_char(); }
};

// Serialisation as the rank, the way the reference gives its symbols to cereal (seqan/alphabet.hpp:85-98,328-341: minimal
// save / load functions found by ADL).  cereal is not a dependency of this header: the two names below are the ones
// CEREAL_SAVE_MINIMAL_FUNCTION_NAME / CEREAL_LOAD_MINIMAL_FUNCTION_NAME expand to by default, so an archive that is present
// picks them up; any type can stand in for `archive_t`.
template <typename archive_t, std::size_t sigma, auto const & table>
constexpr std::uint8_t save_minimal(archive_t const &, nucleotide<sigma, table> const & symbol) noexcept
{
    return symbol.to_rank();
}
template <typename archive_t, std::size_t sigma, auto const & table>
constexpr void load_minimal(archive_t const &, nucleotide<sigma, table> & symbol, std::uint8_t const & rank) noexcept
{
    symbol.assign_rank(rank);
}

using dna4 = nucleotide<4, detail::dna4_table>;
using dna5 = nucleotide<5, detail::dna5_table>;
using dna15 = nucleotide<15, detail::dna15_table>;
static_assert(sizeof(dna4) == 1 && sizeof(dna5) == 1 && sizeof(dna15) == 1);

template <typename symbol_t>
inline constexpr std::size_t alphabet_size_v = std::remove_cvref_t<symbol_t>::alphabet_size;

inline namespace literals
{
    inline std::vector<dna4> operator""_dna4(char const * s, std::size_t n)
    {
        std::vector<dna4> v;
        v.reserve(n < 16 ? 16 : n);
        for (std::size_t i = 0; i < n; ++i)
            v.emplace_back(s[i]);
        return v;
    }
    inline std::vector<dna5> operator""_dna5(char const * s, std::size_t n)
    {
        std::vector<dna5> v;
        v.reserve(n < 16 ? 16 : n);
        for (std::size_t i = 0; i < n; ++i)
            v.emplace_back(s[i]);
        return v;
    }
    inline std::vector<dna15> operator""_dna15(char const * s, std::size_t n)
    {
        std::vector<dna15> v;
        v.reserve(n);
        for (std::size_t i = 0; i < n; ++i)
            v.emplace_back(s[i]);
        return v;
    }
} // namespace literals
} // namespace spm

// The two SeqAn2 metafunctions the reference specialises for its symbols (seqan/alphabet.hpp:100-112), under the names its
// call sites use: number of symbols and bits per symbol (dna4: 4 / 2, dna5: 5 / 3, dna15: 15 / 4).
namespace seqan2
{
template <typename symbol_t>
struct ValueSize;
template <typename symbol_t>
struct BitsPerValue;

template <std::size_t sigma, auto const & table>
struct ValueSize<spm::nucleotide<sigma, table>>
{
    using Type = std::size_t;
    static constexpr Type VALUE = sigma;
};
template <std::size_t sigma, auto const & table>
struct BitsPerValue<spm::nucleotide<sigma, table>>
{
    using Type = std::size_t;
    static constexpr Type VALUE = sigma <= 2 ? 1 : sigma <= 4 ? 2 : sigma <= 8 ? 3 : sigma <= 16 ? 4 : 8;
};
} // namespace seqan2
