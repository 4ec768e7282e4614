// reference_cases.cpp -- the reference's matcher test CASES (not its files) against the MI355X back-end, through the
// C++ mirror of the spm:: API.  Sources of the cases:
//   /root/reference/test/api/libspm/matcher/horspool_matcher_test.cpp:28-52
//   /root/reference/test/api/libspm/matcher/shiftor_matcher_test.cpp:28-52
//   /root/reference/test/api/libspm/matcher/myers_matcher_test.cpp:29-53
//   /root/reference/test/api/libspm/matcher/myers_matcher_restorable_test.cpp:29-74
//   /root/reference/test/api/libspm/matcher/pigeonhole_matcher_test.cpp:20-85
//   /root/reference/test/api/libspm/seqan/alphabet_test.cpp:21-47 (rank/char contract; seqan3 concepts are not available)
// plus cases for what the reference leaves untested (restorable Shift-Or, prefix matcher, dna5, capture inside a
// callback, copies).  No gtest in this image: a 20-line EXPECT harness, exit code = number of failures.
#include <algorithm>
#include <cstdio>
#include <sstream>
#include <utility>
#include <vector>

#include <libspm/matcher/concept.hpp>
#include <libspm/matcher/hip_batch.hpp>
#include <libspm/matcher/horspool_matcher.hpp>
#include <libspm/matcher/myers_matcher.hpp>
#include <libspm/matcher/myers_matcher_restorable.hpp>
#include <libspm/matcher/myers_prefix_matcher_restorable.hpp>
#include <libspm/matcher/pigeonhole_matcher.hpp>
#include <libspm/matcher/shiftor_matcher.hpp>
#include <libspm/matcher/shiftor_matcher_restorable.hpp>
#include <libspm/seqan/alphabet.hpp>
#include <libspm/seqan/container_adapter.hpp>

using spm::operator""_dna4;
using spm::operator""_dna5;

static int failures = 0;
static int checks = 0;
#define EXPECT_TRUE(cond)                                                                                              \
    do {                                                                                                               \
        ++checks;                                                                                                      \
        if (!(cond)) {                                                                                                 \
            ++failures;                                                                                                \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                              \
        }                                                                                                              \
    } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))

using sequence_t = std::vector<spm::dna4>;
                            //0         1         2         3         4
                            //012345678901234567890123456789012345678901234
static sequence_t const haystack = "ACGTGACTAGCACGTGACTAGCACGTGACTAGCACGTGACTAGC"_dna4;
static sequence_t const needle = "GCACG"_dna4;

static void horspool_cases()
{
    std::vector<std::size_t> const expected{9, 20, 31};
    auto matcher = spm::horspool_matcher{needle};
    EXPECT_TRUE(spm::window_matcher<decltype(matcher)>);
    EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle));
    std::vector<std::size_t> actual{};
    matcher(haystack, [&](auto const & finder) { actual.push_back(seqan2::beginPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(actual, expected));
}

static void shiftor_cases()
{
    std::vector<std::size_t> const expected{9, 20, 31};
    auto matcher = spm::shiftor_matcher{needle};
    EXPECT_TRUE(spm::window_matcher<decltype(matcher)>);
    EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle));
    std::vector<std::size_t> actual{};
    matcher(haystack, [&](auto const & finder) { actual.push_back(seqan2::beginPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(actual, expected));
    // the matcher is std::copyable and a copy is independent
    auto copy = matcher;
    actual.clear();
    copy(haystack, [&](auto const & finder) { actual.push_back(seqan2::beginPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(actual, expected));
}

static void myers_cases()
{
    std::size_t const errors = 1;
    std::vector<std::size_t> const expected{13, 14, 15, 24, 25, 26, 35, 36, 37};
    std::vector<int> const expected_errors{1, 0, 1, 1, 0, 1, 1, 0, 1};
    auto matcher = spm::myers_matcher{needle, errors};
    EXPECT_TRUE(spm::window_matcher<decltype(matcher)>);
    EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle) + errors);
    std::vector<std::size_t> actual{};
    std::vector<int> actual_errors{};
    matcher(haystack, [&](auto const & finder) {
        actual.push_back(seqan2::endPosition(finder));
        actual_errors.push_back(finder.errors());
    });
    EXPECT_TRUE(std::ranges::equal(actual, expected));
    EXPECT_TRUE(std::ranges::equal(actual_errors, expected_errors));
    // default max_error_count = 0 -> exact occurrences only, reported by their exclusive end
    auto exact = spm::myers_matcher{needle};
    actual.clear();
    exact(haystack, [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(actual, std::vector<std::size_t>{14, 25, 36}));
    EXPECT_EQ(spm::window_size(exact), std::ranges::size(needle));
}

static void restorable_myers_cases()
{
    unsigned const errors = 1;
    std::vector<std::size_t> const expected{13, 14, 15, 24, 25, 26, 35, 36, 37};
    auto get_matcher = [&] { return spm::restorable_myers_matcher{needle, errors}; };
    using matcher_t = decltype(get_matcher());
    EXPECT_TRUE(spm::window_matcher<matcher_t>);
    EXPECT_TRUE(spm::restorable_matcher<matcher_t>);
    {
        auto matcher = get_matcher();
        EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle) + errors);
        std::vector<std::size_t> actual{};
        matcher(haystack, [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder)); });
        EXPECT_TRUE(std::ranges::equal(actual, expected));
    }
    { // dna4_pattern_captured: chunks of 13, restore(state) before and capture() after each chunk
        std::size_t const chunk_size{13};
        std::ptrdiff_t const chunk_count = (haystack.size() + chunk_size - 1) / chunk_size;
        auto matcher = get_matcher();
        auto state = matcher.capture();
        std::vector<std::size_t> actual{};
        for (std::ptrdiff_t chunk_idx = 0; chunk_idx < chunk_count; ++chunk_idx) {
            std::ptrdiff_t const offset = chunk_idx * chunk_size;
            sequence_t chunk{haystack.begin() + offset,
                             haystack.begin() + std::min<std::ptrdiff_t>(offset + chunk_size, haystack.size())};
            matcher.restore(state);
            matcher(chunk, [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder) + offset); });
            state = matcher.capture();
        }
        EXPECT_TRUE(std::ranges::equal(actual, expected));
    }
    { // capture() inside the callback is the state AT that hit: resuming from it on the rest of the haystack
      // yields exactly the remaining hits
        auto matcher = get_matcher();
        spm::matcher_state_t<matcher_t> at_fourth{};
        std::size_t fourth_end = 0;
        int seen = 0;
        matcher(haystack, [&](auto const & finder) {
            if (++seen == 4) {
                at_fourth = spm::capture(matcher);
                fourth_end = seqan2::endPosition(finder);
            }
        });
        EXPECT_EQ(fourth_end, std::size_t{24});
        EXPECT_EQ(at_fourth.errors(), 1);
        auto resumed = get_matcher();
        spm::restore(resumed, at_fourth);
        sequence_t rest{haystack.begin() + fourth_end, haystack.end()};
        std::vector<std::size_t> actual{};
        resumed(rest, [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder) + fourth_end); });
        EXPECT_TRUE(std::ranges::equal(actual, std::vector<std::size_t>{25, 26, 35, 36, 37}));
    }
}

static void restorable_shiftor_cases()
{
    std::vector<std::size_t> const expected{9, 20, 31};
    auto matcher = spm::restorable_shiftor_matcher{needle};
    using matcher_t = decltype(matcher);
    EXPECT_TRUE(spm::window_matcher<matcher_t>);
    EXPECT_TRUE(spm::restorable_matcher<matcher_t>);
    EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle));
    for (std::size_t chunk_size : {44u, 13u, 11u, 5u, 3u}) {
        auto m = spm::restorable_shiftor_matcher{needle};
        auto state = m.capture();
        std::vector<std::size_t> actual{};
        for (std::size_t offset = 0; offset < haystack.size(); offset += chunk_size) {
            sequence_t chunk{haystack.begin() + offset,
                             haystack.begin() + std::min(offset + chunk_size, haystack.size())};
            m.restore(state);
            m(chunk, [&](auto const & finder) { actual.push_back(seqan2::beginPosition(finder) + offset); });
            state = m.capture();
        }
        EXPECT_TRUE(std::ranges::equal(actual, expected));
    }
}

static void prefix_cases()
{
    // needle vs. prefixes of the haystack: D[0][j] = j.  "GCACG" against "GCCGTT..." : prefix "GCCG" is 1 edit away.
    sequence_t const hay = "GCCGTTTTTTTT"_dna4;
    unsigned const errors = 1;
    auto matcher = spm::restorable_myers_prefix_matcher{needle, errors};
    EXPECT_TRUE(spm::window_matcher<decltype(matcher)>);
    EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle) + errors);
    std::vector<std::size_t> ends{};
    std::vector<int> errs{};
    matcher(hay, [&](auto const & finder) {
        ends.push_back(seqan2::endPosition(finder));
        errs.push_back(finder.errors());
    });
    // Sellers, global start: prefixes of length 4 (GCCG: 1 edit... delete A) and 5 (GCCGT: 2) -> only end 4 has <= 1
    EXPECT_TRUE(std::ranges::equal(ends, std::vector<std::size_t>{4}));
    EXPECT_TRUE(std::ranges::equal(errs, std::vector<int>{1}));
    sequence_t const empty{};
    auto none = spm::restorable_myers_prefix_matcher{empty, errors};
    EXPECT_EQ(spm::window_size(none), std::size_t{0});
    int calls = 0;
    none(hay, [&](auto const &) { ++calls; });
    EXPECT_EQ(calls, 0);
}

static void pigeonhole_cases()
{
    { // the position prints as the reference's does
        std::ostringstream os;
        os << seqan2::PigeonholeSeedOnlyPosition{1, 5, 5};
        EXPECT_TRUE(os.str() == "<1, 5, 5>");
    }

    using needle_position_t = seqan2::PigeonholeSeedOnlyPosition;
    sequence_t const needle2 = "TGACTAGCAC"_dna4;
    std::vector<sequence_t> const multi_needle{needle, needle2};
    double const errors = 0.0;
    std::vector<std::size_t> const expected_positions{9, 20, 31};
    std::vector<std::size_t> const expected_multi_positions{3, 8, 9, 14, 19, 20, 25, 30, 31, 36};
    std::vector<needle_position_t> const expected_needle_positions{{1, 0, 5}, {1, 5, 5}, {0, 0, 5}, {1, 0, 5},
                                                                   {1, 5, 5}, {0, 0, 5}, {1, 0, 5}, {1, 5, 5},
                                                                   {0, 0, 5}, {1, 0, 5}};
    {
        auto matcher = spm::pigeonhole_matcher{needle, errors};
        EXPECT_TRUE(spm::window_matcher<decltype(matcher)>);
        EXPECT_EQ(spm::window_size(matcher), std::ranges::size(needle));
        std::vector<std::size_t> actual{};
        matcher(haystack, [&](auto const & finder) { actual.push_back(seqan2::beginPosition(finder)); });
        EXPECT_TRUE(std::ranges::equal(actual, expected_positions));
    }
    {
        auto matcher = spm::pigeonhole_matcher{multi_needle, errors};
        std::vector<std::size_t> actual{};
        std::vector<needle_position_t> actual_needle_positions{};
        matcher(haystack, [&](auto const & finder) {
            actual.push_back(seqan2::beginPosition(finder));
            actual_needle_positions.push_back(matcher.position());
        });
        EXPECT_TRUE(std::ranges::equal(actual, expected_multi_positions));
        EXPECT_TRUE(actual_needle_positions == expected_needle_positions);
    }
    { // seeds long enough for the device seed filter: 2 needles of 40, error rate 0.05 -> e = 2, q = 13 ... use 0.0 -> q = 40
        std::vector<spm::dna4> hay;
        for (int i = 0; i < 4000; ++i)
            hay.emplace_back(static_cast<std::uint8_t>((i * 7 + (i >> 3) + (i >> 7)) & 3));
        sequence_t n1(hay.begin() + 1000, hay.begin() + 1040), n2(hay.begin() + 2500, hay.begin() + 2540);
        std::vector<sequence_t> const two{n1, n2};
        auto matcher = spm::pigeonhole_matcher{two, 0.0};
        EXPECT_EQ(spm::window_size(matcher), std::size_t{40});
        std::vector<std::size_t> begins{};
        matcher(hay, [&](auto const & finder) { begins.push_back(seqan2::beginPosition(finder)); });
        EXPECT_TRUE(std::ranges::find(begins, std::size_t{1000}) != begins.end());
        EXPECT_TRUE(std::ranges::find(begins, std::size_t{2500}) != begins.end());
        EXPECT_TRUE(std::ranges::is_sorted(begins));
    }
}

static void batch_cases()
{
    // the batch front-end delivers exactly the callbacks of the per-needle matchers, grouped by needle
    sequence_t const needle2 = "TGACTAGCAC"_dna4;
    std::vector<sequence_t> const needles{needle, needle2, "ACGT"_dna4};
    auto batch = spm::batch_myers_matcher{needles, 1};
    EXPECT_TRUE(spm::window_matcher<decltype(batch)>);
    EXPECT_EQ(spm::window_size(batch), std::size_t{11});
    std::vector<std::vector<std::size_t>> got(3), want(3);
    std::vector<std::vector<int>> got_e(3), want_e(3);
    batch(haystack, [&](std::size_t i, auto const & finder) {
        got[i].push_back(seqan2::endPosition(finder));
        got_e[i].push_back(finder.errors());
    });
    for (std::size_t i = 0; i < 3; ++i) {
        auto single = spm::myers_matcher{needles[i], 1};
        single(haystack, [&](auto const & finder) {
            want[i].push_back(seqan2::endPosition(finder));
            want_e[i].push_back(finder.errors());
        });
        EXPECT_TRUE(got[i] == want[i] && got_e[i] == want_e[i] && !want[i].empty());
    }
    EXPECT_TRUE(std::ranges::equal(got[0], std::vector<std::size_t>{13, 14, 15, 24, 25, 26, 35, 36, 37}));
    auto exact = spm::batch_shiftor_matcher{needles};
    std::vector<std::vector<std::size_t>> begins(3);
    exact(haystack, [&](std::size_t i, auto const & finder) { begins[i].push_back(seqan2::beginPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(begins[0], std::vector<std::size_t>{9, 20, 31}));
    EXPECT_TRUE(std::ranges::equal(begins[1], std::vector<std::size_t>{3, 14, 25}));
    EXPECT_TRUE(std::ranges::equal(begins[2], std::vector<std::size_t>{0, 11, 22, 33}));
}

static void alphabet_cases()
{
    { // serialisation as the rank (the minimal save / load pair cereal looks up by ADL: alphabet.hpp:85-98 in the reference)
        struct archive {} const ar{};
        spm::dna5 const n{'N'};
        EXPECT_EQ(save_minimal(ar, n), std::uint8_t{3});
        spm::dna5 back{};
        load_minimal(ar, back, save_minimal(ar, n));
        EXPECT_TRUE(back == n && back.to_char() == 'N');
        spm::dna15 y{};
        load_minimal(ar, y, save_minimal(ar, spm::dna15{'Y'}));
        EXPECT_EQ(y.to_char(), 'Y');
        static_assert(save_minimal(0, spm::dna4{'G'}) == 2);
    }
    // the trait spellings of tag_invoke (std/tag_invoke.hpp:79-102 in the reference)
    static_assert(std::is_tag_invocable_v<std::tag_t<spm::window_size>, spm::myers_matcher<sequence_t> const &>);
    static_assert(!std::is_tag_invocable<std::tag_t<spm::window_size>, int>::value);
    static_assert(std::same_as<std::tag_invoke_result<std::tag_t<spm::window_size>, spm::myers_matcher<sequence_t> const &>::type, std::size_t>);
    // seqan2::ValueSize / BitsPerValue of the symbols (alphabet.hpp:100-112 in the reference)
    static_assert(seqan2::ValueSize<spm::dna4>::VALUE == 4 && seqan2::BitsPerValue<spm::dna4>::VALUE == 2);
    static_assert(seqan2::ValueSize<spm::dna5>::VALUE == 5 && seqan2::BitsPerValue<spm::dna5>::VALUE == 3);
    static_assert(seqan2::ValueSize<spm::dna15>::VALUE == 15 && seqan2::BitsPerValue<spm::dna15>::VALUE == 4);

    static_assert(sizeof(spm::dna4) == 1 && sizeof(spm::dna5) == 1 && sizeof(spm::dna15) == 1);
    static_assert(std::semiregular<spm::dna4> && std::totally_ordered<spm::dna4>);
    EXPECT_EQ(spm::alphabet_size_v<spm::dna4>, std::size_t{4});
    EXPECT_EQ(spm::alphabet_size_v<spm::dna5>, std::size_t{5});
    EXPECT_EQ(spm::alphabet_size_v<spm::dna15>, std::size_t{15});
    auto s = "ACGTacgu"_dna4;
    std::vector<int> ranks(s.begin(), s.end());
    EXPECT_TRUE(std::ranges::equal(ranks, std::vector<int>{0, 1, 2, 3, 0, 1, 2, 3}));
    EXPECT_EQ(static_cast<char>(spm::dna4{'G'}), 'G');
    EXPECT_EQ(static_cast<int>(spm::dna5{'N'}), 3);
    EXPECT_EQ(static_cast<int>(spm::dna5{'T'}), 4);
    EXPECT_EQ(static_cast<int>(spm::dna5{'X'}), 3);
    EXPECT_EQ(static_cast<char>(spm::dna15{std::uint8_t{14}}), 'Y');
    EXPECT_TRUE(spm::dna4{'A'} < spm::dna4{'C'});
    // dna5 haystack with N (brute engine: sigma = 5)
    auto hay5 = "ACGNTACGNTAACGT"_dna5;
    auto ndl5 = "ACGNT"_dna5;
    auto m5 = spm::myers_matcher{ndl5, 0};
    std::vector<std::size_t> ends{};
    m5(hay5, [&](auto const & finder) { ends.push_back(seqan2::endPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(ends, std::vector<std::size_t>{5, 10}));
}

// The view adapter of container_adapter.hpp:23-100: container interface over a view + the rank buffer the C ABI takes.
static void container_adapter_cases()
{
    auto whole = spm::make_seqan_container(std::views::all(haystack));
    static_assert(decltype(whole)::zero_copy);
    static_assert(std::same_as<decltype(whole), spm::seqan_container_t<std::views::all_t<sequence_t const &>>>);
    static_assert(std::copyable<decltype(whole)> && std::default_initializable<decltype(whole)>);
    EXPECT_EQ(whole.size(), haystack.size());
    EXPECT_EQ(seqan2::length(whole), haystack.size());
    EXPECT_TRUE(!whole.empty() && !seqan2::empty(whole));
    EXPECT_TRUE(whole[9] == spm::dna4{'G'} && *whole.begin() == spm::dna4{'A'});
    EXPECT_TRUE(std::ranges::equal(whole, haystack));
    EXPECT_TRUE(static_cast<void const *>(whole.ranks().data()) == static_cast<void const *>(haystack.data())); // no copy
    EXPECT_EQ(whole.ranks().size(), haystack.size());

    decltype(whole) blank{};
    EXPECT_TRUE(blank.empty() && blank.ranks().empty() && !blank.has_view());
    spm::assign(blank, whole);
    EXPECT_TRUE(blank.has_view() && blank.size() == haystack.size());

    // a non-contiguous view is staged into rank bytes
    auto reversed = spm::make_seqan_container(haystack | std::views::reverse);
    static_assert(!decltype(reversed)::zero_copy);
    EXPECT_EQ(reversed.size(), haystack.size());
    EXPECT_TRUE(reversed[0] == haystack.back());
    auto rr = reversed.ranks();
    bool same = rr.size() == haystack.size();
    for (std::size_t i = 0; same && i < rr.size(); ++i)
        same = rr[i] == static_cast<std::uint8_t>(static_cast<unsigned>(haystack[haystack.size() - 1 - i]));
    EXPECT_TRUE(same);

    // matchers take such views directly: reversed haystack, reversed needle -> mirrored begin positions
    auto rneedle = needle | std::views::reverse;
    auto matcher = spm::shiftor_matcher{rneedle};
    std::vector<std::size_t> begins{};
    matcher(haystack | std::views::reverse, [&](auto const & finder) { begins.push_back(seqan2::beginPosition(finder)); });
    std::vector<std::size_t> expected{};
    for (std::size_t b : {31, 20, 9})
        expected.push_back(haystack.size() - (b + needle.size()));
    EXPECT_TRUE(std::ranges::equal(begins, expected));

    // a transformed view (complement) over a subrange
    auto complement = [](spm::dna4 c) { return spm::dna4{static_cast<std::uint8_t>(3 - static_cast<unsigned>(c))}; };
    auto comp_hay = haystack | std::views::drop(4) | std::views::transform(complement);
    auto comp_needle = needle | std::views::transform(complement);
    auto cm = spm::myers_matcher{comp_needle, 0};
    std::vector<std::size_t> ends{};
    cm(comp_hay, [&](auto const & finder) { ends.push_back(seqan2::endPosition(finder)); });
    EXPECT_TRUE(std::ranges::equal(ends, std::vector<std::size_t>{10, 21, 32}));
}

// What the paused scan of the reference implies for calls made INSIDE the per-hit callback, and the absence of a hit limit
// in its find loop (seqan_pattern_base.hpp:49-51).
static void paused_scan_cases()
{
    unsigned const errors = 1;
    auto get_matcher = [&] { return spm::restorable_myers_matcher{needle, errors}; };
    { // capture() at EVERY hit: each state, resumed on the rest of the haystack, yields exactly the remaining hits
        auto matcher = get_matcher();
        using state_t = spm::matcher_state_t<decltype(matcher)>;
        std::vector<std::pair<std::size_t, state_t>> at;
        std::vector<std::size_t> all;
        matcher(haystack, [&](auto const & finder) {
            all.push_back(seqan2::endPosition(finder));
            at.emplace_back(seqan2::endPosition(finder), spm::capture(matcher));
        });
        EXPECT_EQ(all.size(), std::size_t{9});
        for (std::size_t i = 0; i < at.size(); ++i) {
            auto resumed = get_matcher();
            spm::restore(resumed, at[i].second);
            sequence_t rest{haystack.begin() + at[i].first, haystack.end()};
            std::vector<std::size_t> got{};
            resumed(rest, [&](auto const & finder) { got.push_back(seqan2::endPosition(finder) + at[i].first); });
            EXPECT_TRUE(std::ranges::equal(got, std::vector<std::size_t>(all.begin() + i + 1, all.end())));
        }
        // the state after the call is the state after the last symbol, whatever was captured on the way
        auto plain = get_matcher();
        plain(haystack, [](auto const &) {});
        EXPECT_TRUE(spm::capture(matcher) == spm::capture(plain));
    }
    { // restore() inside a callback: the rest of the chunk continues from the restored state.  Restoring the
      // constructor-time state at the 2nd hit (end 14 = the exact occurrence) makes the matcher forget the symbols read
      // so far: the hit at 15 (one deletion, needs the occurrence's prefix) disappears, the later occurrences stay
        auto matcher = get_matcher();
        auto const fresh = spm::capture(matcher);
        std::vector<std::size_t> got{};
        int seen = 0;
        matcher(haystack, [&](auto const & finder) {
            got.push_back(seqan2::endPosition(finder));
            if (++seen == 2)
                spm::restore(matcher, fresh);
        });
        EXPECT_TRUE(std::ranges::equal(got, std::vector<std::size_t>{13, 14, 24, 25, 26, 35, 36, 37}));
        // ... which is what two separate scans of the two parts give
        auto a = get_matcher(), b = get_matcher();
        std::vector<std::size_t> two{};
        sequence_t const part1{haystack.begin(), haystack.begin() + 14}, part2{haystack.begin() + 14, haystack.end()};
        a(part1, [&](auto const & f) { two.push_back(seqan2::endPosition(f)); });
        b(part2, [&](auto const & f) { two.push_back(seqan2::endPosition(f) + 14); });
        EXPECT_TRUE(std::ranges::equal(got, two));
    }
    { // restore() then capture() inside ONE callback: capture returns the live pattern state, i.e. the restored one
      // (myers_matcher_restorable.hpp:55-61), not the state at the hit
        auto matcher = get_matcher();
        auto const fresh = spm::capture(matcher);
        int seen = 0;
        bool same_as_restored = false, differs_before = false;
        matcher(haystack, [&](auto const &) {
            if (++seen == 2) {
                differs_before = !(spm::capture(matcher) == fresh); // (the state at the 2nd hit has read 14 symbols)
                spm::restore(matcher, fresh);
                same_as_restored = spm::capture(matcher) == fresh;
            }
        });
        EXPECT_TRUE(differs_before);
        EXPECT_TRUE(same_as_restored);
    }
    { // more hits than the default device hit buffer (2^20): |P| = 5, k = 1 on 3 Mbases of ACGT..: no limit, no abort
        std::size_t const n = 3u << 20;
        sequence_t big(n);
        for (std::size_t i = 0; i < n; ++i)
            big[i] = spm::dna4{static_cast<std::uint8_t>(i & 3)};
        sequence_t const short_needle = "ACGTA"_dna4;
        auto matcher = spm::myers_matcher{short_needle, 1};
        std::size_t hits = 0, last = 0;
        bool ascending = true;
        matcher(big, [&](auto const & finder) {
            ascending = ascending && seqan2::endPosition(finder) >= last;
            last = seqan2::endPosition(finder);
            ++hits;
        });
        EXPECT_TRUE(hits > (1u << 21)); // every position from the 4th on ends an occurrence with <= 1 error
        EXPECT_TRUE(ascending);
        EXPECT_EQ(last, n);
        // the pigeonhole matcher: every window of 8 symbols with the right phase is a seed hit
        std::vector<sequence_t> seeds{"ACGTACGT"_dna4};
        auto pm = spm::pigeonhole_matcher{seeds, 0.0};
        std::size_t seed_hits = 0;
        pm(big, [&](auto const &) { ++seed_hits; });
        EXPECT_EQ(seed_hits, (n - 8) / 4 + 1);
    }
}

static void resident_haystack_cases()
{
    // the reference's cases again with the haystack uploaded ONCE (spm::hip::resident_haystack) and handed to every
    // matcher -- whole, and as slices for the chunk-of-13 capture/restore walk (myers_matcher_restorable_test.cpp:55-74)
    spm::hip::resident_haystack const hs{haystack};
    EXPECT_EQ(hs.size(), haystack.size());
    {
        std::vector<std::size_t> a, b, c;
        spm::horspool_matcher{needle}(hs, [&](auto const & f) { a.push_back(seqan2::beginPosition(f)); });
        spm::shiftor_matcher{needle}(hs, [&](auto const & f) { b.push_back(seqan2::beginPosition(f)); });
        spm::myers_matcher{needle, 1}(hs, [&](auto const & f) { c.push_back(seqan2::endPosition(f)); });
        EXPECT_TRUE(std::ranges::equal(a, std::vector<std::size_t>{9, 20, 31}));
        EXPECT_TRUE(std::ranges::equal(b, std::vector<std::size_t>{9, 20, 31}));
        EXPECT_TRUE(std::ranges::equal(c, std::vector<std::size_t>{13, 14, 15, 24, 25, 26, 35, 36, 37}));
    }
    { // a slice is a haystack of its own: positions from its first symbol, nothing before it is seen
        std::vector<std::size_t> got, want;
        spm::myers_matcher{needle, 1}(hs.slice(11, 40), [&](auto const & f) { got.push_back(seqan2::endPosition(f)); });
        sequence_t const sub{haystack.begin() + 11, haystack.begin() + 40};
        spm::myers_matcher{needle, 1}(sub, [&](auto const & f) { want.push_back(seqan2::endPosition(f)); });
        EXPECT_TRUE(got == want && !want.empty());
        std::vector<std::size_t> none;
        spm::shiftor_matcher{needle}(hs.slice(10, 13), [&](auto const & f) { none.push_back(seqan2::beginPosition(f)); });
        EXPECT_TRUE(none.empty());
        EXPECT_EQ(hs.slice(40, 100).size(), std::size_t{4});
        EXPECT_TRUE(hs.slice(7, 3).empty());
    }
    { // chunks of 13 as slices, restore(state) before and capture() after each
        std::size_t const chunk_size{13};
        auto matcher = spm::restorable_myers_matcher{needle, 1u};
        auto state = matcher.capture();
        std::vector<std::size_t> actual{};
        for (std::size_t offset = 0; offset < haystack.size(); offset += chunk_size) {
            matcher.restore(state);
            matcher(hs.slice(offset, offset + chunk_size),
                    [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder) + offset); });
            state = matcher.capture();
        }
        EXPECT_TRUE(std::ranges::equal(actual, std::vector<std::size_t>{13, 14, 15, 24, 25, 26, 35, 36, 37}));
    }
    { // capture() inside a callback on a slice, resumed on the rest of the resident sequence
        auto matcher = spm::restorable_myers_matcher{needle, 1u};
        spm::matcher_state_t<decltype(matcher)> at_second{};
        std::size_t second_end = 0;
        int seen = 0;
        matcher(hs.slice(5, 44), [&](auto const & finder) {
            if (++seen == 2) {
                at_second = spm::capture(matcher);
                second_end = seqan2::endPosition(finder) + 5;
            }
        });
        EXPECT_EQ(second_end, std::size_t{14});
        auto resumed = spm::restorable_myers_matcher{needle, 1u};
        spm::restore(resumed, at_second);
        std::vector<std::size_t> actual{};
        resumed(hs.slice(second_end, 44), [&](auto const & finder) { actual.push_back(seqan2::endPosition(finder) + second_end); });
        EXPECT_TRUE(std::ranges::equal(actual, std::vector<std::size_t>{15, 24, 25, 26, 35, 36, 37}));
    }
    { // pigeonhole seed hits (pigeonhole_matcher_test.cpp:54-85) on the resident haystack and on a slice of it
        sequence_t const needle2 = "TGACTAGCAC"_dna4;
        std::vector<sequence_t> const needles{needle, needle2};
        auto matcher = spm::pigeonhole_matcher{needles};
        std::vector<std::size_t> got, want, got_s, want_s;
        matcher(hs, [&](auto const & f) { got.push_back(seqan2::beginPosition(f)); });
        matcher(haystack, [&](auto const & f) { want.push_back(seqan2::beginPosition(f)); });
        EXPECT_TRUE(got == want && std::ranges::equal(want, std::vector<std::size_t>{3, 8, 9, 14, 19, 20, 25, 30, 31, 36}));
        matcher(hs.slice(2, 33), [&](auto const & f) { got_s.push_back(seqan2::beginPosition(f)); });
        sequence_t const sub{haystack.begin() + 2, haystack.begin() + 33};
        matcher(sub, [&](auto const & f) { want_s.push_back(seqan2::beginPosition(f)); });
        EXPECT_TRUE(got_s == want_s && !want_s.empty());
    }
    { // the batch front-end on the resident haystack == on the host range
        sequence_t const needle2 = "TGACTAGCAC"_dna4;
        std::vector<sequence_t> const needles{needle, needle2, "ACGT"_dna4};
        auto batch = spm::batch_myers_matcher{needles, 1};
        std::vector<std::pair<std::size_t, std::size_t>> got, want, got_s, want_s;
        batch(hs, [&](std::size_t i, auto const & f) { got.emplace_back(i, seqan2::endPosition(f)); });
        batch(haystack, [&](std::size_t i, auto const & f) { want.emplace_back(i, seqan2::endPosition(f)); });
        EXPECT_TRUE(got == want && want.size() > 9);
        batch(hs.slice(3, 30), [&](std::size_t i, auto const & f) { got_s.emplace_back(i, seqan2::endPosition(f)); });
        sequence_t const sub{haystack.begin() + 3, haystack.begin() + 30};
        batch(sub, [&](std::size_t i, auto const & f) { want_s.emplace_back(i, seqan2::endPosition(f)); });
        EXPECT_TRUE(got_s == want_s && !want_s.empty());
    }
}

int main()
{
    horspool_cases();
    shiftor_cases();
    myers_cases();
    restorable_myers_cases();
    restorable_shiftor_cases();
    prefix_cases();
    pigeonhole_cases();
    batch_cases();
    alphabet_cases();
    container_adapter_cases();
    paused_scan_cases();
    resident_haystack_cases();
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures;
}
