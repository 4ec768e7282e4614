// libspm/jst/journaled_sequence.hpp -- Journal / JournalEntry / JournaledSequence.
//
// The reference ships only a design for these (specs/journaled_sequence_class_diagram.drawio: JournaledSequence
// :7-109, Journal :115-238, JournalEntry :241-298, invariants note :444); there is no code to mirror.  This is a
// from-scratch implementation of that design, reduced to what a haplotype needs:
//   * a Journal is a sorted vector of entries {begin_position, segment}; segments are NON-OWNING views into a source
//     sequence (the reference, or the storage of an alternative allele);
//   * invariants: the first entry begins at 0, adjacent entries are contiguous
//     (e1.end_position() == e2.begin_position()), a sentinel last entry makes size() == back().begin_position();
//   * record_sequence_edit(i, j, s) replaces positions [i, j) by segment s (insertion: i == j, deletion: s empty);
//   * JournaledSequence wraps a Journal and offers a pure random-access sequence interface.
#pragma once

#include <algorithm>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <span>
#include <vector>

namespace spm
{
template <typename value_t>
class journal_entry
{
public:
    using sequence_type = std::span<value_t const>;
    using size_type = std::size_t;

private:
    size_type _position{};
    sequence_type _segment{};

public:
    journal_entry() = default;
    journal_entry(size_type p, sequence_type s) noexcept : _position{p}, _segment{s} {}
    size_type begin_position() const noexcept { return _position; }
    size_type end_position() const noexcept { return _position + _segment.size(); }
    sequence_type segment() const noexcept { return _segment; }
    void shift(std::ptrdiff_t d) noexcept { _position = static_cast<size_type>(static_cast<std::ptrdiff_t>(_position) + d); }
    friend bool position_is_covered_by(journal_entry const & e, size_type p) noexcept
    {
        return e.begin_position() <= p && p < e.end_position();
    }
    // split at offset `o` inside the segment -> (left, right)
    friend std::pair<journal_entry, journal_entry> split_at(journal_entry const & e, size_type o) noexcept
    {
        return {journal_entry{e._position, e._segment.first(o)}, journal_entry{e._position + o, e._segment.subspan(o)}};
    }
};

template <typename value_t>
class journal
{
public:
    using entry_type = journal_entry<value_t>;
    using sequence_type = typename entry_type::sequence_type;
    using size_type = std::size_t;
    using const_iterator = typename std::vector<entry_type>::const_iterator;

private:
    std::vector<entry_type> _entries{entry_type{0, {}}}; // sentinel only: the empty sequence

public:
    journal() = default;
    explicit journal(sequence_type source) { reset(source); }

    void reset(sequence_type source)
    {
        _entries.clear();
        if (!source.empty())
            _entries.emplace_back(0, source);
        _entries.emplace_back(source.size(), sequence_type{});
    }

    size_type size() const noexcept { return _entries.back().begin_position(); }
    const_iterator begin() const noexcept { return _entries.begin(); }
    const_iterator end() const noexcept { return _entries.end() - 1; } // the sentinel is not an element
    std::size_t entry_count() const noexcept { return _entries.size() - 1; }

    // first entry whose end_position() > p  (the entry covering p, for p < size())
    const_iterator upper_bound(size_type p) const noexcept
    {
        return std::upper_bound(_entries.begin(), _entries.end() - 1, p,
                                [](size_type v, entry_type const & e) { return v < e.end_position(); });
    }
    // first entry whose begin_position() >= p
    const_iterator lower_bound(size_type p) const noexcept
    {
        return std::lower_bound(_entries.begin(), _entries.end() - 1, p,
                                [](entry_type const & e, size_type v) { return e.begin_position() < v; });
    }
    const_iterator find(size_type p) const noexcept { return upper_bound(p); }

    // Replace the positions [i, j) by `s`.  Returns the position just behind the recorded segment.
    size_type record_sequence_edit(size_type i, size_type j, sequence_type s)
    {
        assert(i <= j && j <= size());
        std::vector<entry_type> out;
        out.reserve(_entries.size() + 2);
        std::ptrdiff_t const delta = static_cast<std::ptrdiff_t>(s.size()) - static_cast<std::ptrdiff_t>(j - i);
        bool placed = false;
        auto place = [&] {
            if (!placed) {
                if (!s.empty())
                    out.emplace_back(i, s);
                placed = true;
            }
        };
        for (std::size_t n = 0; n + 1 < _entries.size(); ++n) {
            entry_type const & e = _entries[n];
            if (e.end_position() <= i) { // entirely left of the edit
                out.push_back(e);
                continue;
            }
            if (e.begin_position() >= j) { // entirely right
                place();
                entry_type r = e;
                r.shift(delta);
                out.push_back(r);
                continue;
            }
            // overlaps [i, j) (or, for an insertion, contains i strictly inside)
            if (e.begin_position() < i)
                out.push_back(split_at(e, i - e.begin_position()).first);
            place();
            if (e.end_position() > j) {
                entry_type r = split_at(e, j - e.begin_position()).second;
                r.shift(delta);
                out.push_back(r);
            }
        }
        place();
        out.emplace_back(static_cast<size_type>(static_cast<std::ptrdiff_t>(size()) + delta), sequence_type{});
        _entries.swap(out);
        assert(check_journal_invariants());
        return i + s.size();
    }

    bool check_journal_invariants() const noexcept
    {
        if (_entries.empty() || _entries.front().begin_position() != 0)
            return false;
        for (std::size_t n = 0; n + 1 < _entries.size(); ++n)
            if (_entries[n].end_position() != _entries[n + 1].begin_position() || _entries[n].segment().empty())
                return false;
        return _entries.back().segment().empty();
    }
};

template <typename value_t>
class journaled_sequence
{
public:
    using journal_type = journal<value_t>;
    using segment_type = typename journal_type::sequence_type;
    using value_type = value_t;
    using size_type = std::size_t;
    using difference_type = std::ptrdiff_t;

    class const_iterator
    {
        journaled_sequence const * _host{};
        size_type _pos{};

    public:
        using iterator_category = std::random_access_iterator_tag;
        using value_type = value_t;
        using difference_type = std::ptrdiff_t;
        using pointer = value_t const *;
        using reference = value_t const &;
        const_iterator() = default;
        const_iterator(journaled_sequence const * h, size_type p) noexcept : _host{h}, _pos{p} {}
        reference operator*() const { return (*_host)[_pos]; }
        reference operator[](difference_type d) const { return (*_host)[_pos + d]; }
        const_iterator & operator++() noexcept { ++_pos; return *this; }
        const_iterator operator++(int) noexcept { auto t = *this; ++_pos; return t; }
        const_iterator & operator--() noexcept { --_pos; return *this; }
        const_iterator operator--(int) noexcept { auto t = *this; --_pos; return t; }
        const_iterator & operator+=(difference_type d) noexcept { _pos += d; return *this; }
        const_iterator & operator-=(difference_type d) noexcept { _pos -= d; return *this; }
        friend const_iterator operator+(const_iterator i, difference_type d) noexcept { return i += d; }
        friend const_iterator operator+(difference_type d, const_iterator i) noexcept { return i += d; }
        friend const_iterator operator-(const_iterator i, difference_type d) noexcept { return i -= d; }
        friend difference_type operator-(const_iterator const & a, const_iterator const & b) noexcept
        {
            return static_cast<difference_type>(a._pos) - static_cast<difference_type>(b._pos);
        }
        friend bool operator==(const_iterator const & a, const_iterator const & b) noexcept { return a._pos == b._pos; }
        friend auto operator<=>(const_iterator const & a, const_iterator const & b) noexcept { return a._pos <=> b._pos; }
        size_type position() const noexcept { return _pos; }
    };
    using iterator = const_iterator;

private:
    journal_type _journal{};

public:
    journaled_sequence() = default;
    explicit journaled_sequence(segment_type source) : _journal{source} {}

    const_iterator begin() const noexcept { return {this, 0}; }
    const_iterator end() const noexcept { return {this, size()}; }
    value_t const & operator[](size_type p) const
    {
        auto it = _journal.find(p);
        return it->segment()[p - it->begin_position()];
    }
    void clear() { _journal = journal_type{}; }
    const_iterator insert(const_iterator i, segment_type s)
    {
        _journal.record_sequence_edit(i.position(), i.position(), s);
        return {this, i.position()};
    }
    const_iterator erase(const_iterator i, const_iterator j)
    {
        _journal.record_sequence_edit(i.position(), j.position(), segment_type{});
        return {this, i.position()};
    }
    const_iterator erase(const_iterator i) { return erase(i, i + 1); }
    const_iterator replace(const_iterator i, const_iterator j, segment_type s)
    {
        _journal.record_sequence_edit(i.position(), j.position(), s);
        return {this, i.position()};
    }
    size_type size() const noexcept { return _journal.size(); }
    bool empty() const noexcept { return size() == 0; }
    journal_type const & get_journal() const noexcept { return _journal; }

    // contiguous copy (what gets uploaded to the GPU); segment-wise, no per-element lookup
    std::vector<value_t> materialize() const
    {
        std::vector<value_t> out;
        out.reserve(size());
        for (auto const & e : _journal)
            out.insert(out.end(), e.segment().begin(), e.segment().end());
        return out;
    }
};
} // namespace spm
