// tables_build.hpp -- host side of the bit-vector engines: the per-needle match-mask tables (what the reference's matcher
// constructors build, /root/reference/libspm/libspm/matcher/myers_matcher.hpp:40-43, shiftor_matcher.hpp:38-40).  PURE HOST
// C++17, threaded over lane groups.  Layout (brute.hpp): [group][row = symbol, sigma = "invalid"][word][lane], NW 32-bit words
// per needle.
#pragma once

#include <vector>

#include "index_build.hpp"

namespace spm_hip
{

struct brute_tables
{
    std::vector<uint32_t> peq;    // the engine's own table: Myers match masks (bit set = match) or Shift-Or masks (bit clear = match), top-aligned
    std::vector<uint32_t> verify; // exact matchers: Myers-style match masks, top-aligned (the filter engine verifies with Myers at k = 0)
    std::vector<uint32_t> bot;    // Myers: match masks, bottom-aligned (cut-off kernel, wave-per-band verification)
    std::vector<uint32_t> hp0;    // prefix matcher: [group][word][lane], the carry-in bit at the needle's first row
};

// bits [lo, hi) of the NW-word column of one lane: words lie 64 apart
inline void column_fill(uint32_t *col, uint32_t lo, uint32_t hi, bool set)
{
    for (uint32_t w = lo / 32; w * 32 < hi; ++w) {
        const uint32_t b0 = std::max(lo, w * 32) - w * 32, b1 = std::min(hi, w * 32 + 32) - w * 32;
        const uint32_t mask = (b1 >= 32 ? 0xFFFFFFFFu : ((1u << b1) - 1)) & ~((1u << b0) - 1);
        if (set)
            col[(size_t)w * 64] |= mask;
        else
            col[(size_t)w * 64] &= ~mask;
    }
}

inline void build_brute_tables(const needle_view &nv, uint32_t n_groups, uint32_t NW, bool want_verify, unsigned n_threads,
                               brute_tables &T)
{
    const uint32_t rows = nv.sigma + 1;
    const bool myers = nv.is_myers();
    const size_t words = (size_t)n_groups * rows * NW * 64;
    T.peq.assign(words, myers ? 0u : 0xFFFFFFFFu);
    if (!myers && want_verify)
        T.verify.assign(words, 0);
    if (nv.algo == SPM_ALGO_MYERS)
        T.bot.assign(words, 0);
    if (nv.algo == SPM_ALGO_MYERS_PREFIX)
        T.hp0.assign((size_t)n_groups * NW * 64, 0);
    parallel_slices(n_groups, n_threads, [&](size_t g0, size_t g1, unsigned) {
        for (size_t g = g0; g < g1; ++g)
            for (uint32_t l = 0; l < 64; ++l) {
                const size_t p = g * 64 + l;
                const uint32_t m = p < nv.n ? (uint32_t)nv.m[p] : 0;
                if (m == 0)
                    continue; // (Shift-Or: all ones = never matches; Myers: all zero)
                const uint32_t off = NW * 32 - m;
                const uint8_t *pat = nv.ranks + nv.offsets[p];
                auto col = [&](std::vector<uint32_t> &tab, uint32_t row) { return tab.data() + ((g * rows + row) * NW) * 64 + l; };
                // rows above the needle (bits < off) are wildcards: they match every symbol, also the invalid one
                for (uint32_t row = 0; row < rows; ++row) {
                    column_fill(col(T.peq, row), 0, off, myers);
                    if (!T.verify.empty())
                        column_fill(col(T.verify, row), 0, off, true);
                }
                for (uint32_t j = 0; j < m; ++j) {
                    const uint8_t c = pat[j];
                    if (c >= nv.sigma)
                        continue;
                    const uint32_t b = off + j;
                    if (myers)
                        col(T.peq, c)[(size_t)(b / 32) * 64] |= 1u << (b % 32);
                    else
                        col(T.peq, c)[(size_t)(b / 32) * 64] &= ~(1u << (b % 32));
                    if (!T.verify.empty())
                        col(T.verify, c)[(size_t)(b / 32) * 64] |= 1u << (b % 32);
                    if (!T.bot.empty())
                        col(T.bot, c)[(size_t)(j / 32) * 64] |= 1u << (j % 32);
                }
                if (!T.hp0.empty())
                    T.hp0[(g * NW + off / 32) * 64 + l] = 1u << (off % 32);
            }
    });
}

// dna4 needles 2 bits per symbol, 16 per word, every needle from a word of its own (piece count of the resolve kernel)
inline void pack_needles(const needle_view &nv, std::vector<uint32_t> &pk, std::vector<uint32_t> &pk_off, unsigned n_threads = 1)
{
    pk_off.assign(nv.n, 0);
    size_t total = 0;
    for (uint32_t p = 0; p < nv.n; ++p) {
        pk_off[p] = (uint32_t)total;
        total += ((uint32_t)nv.m[p] + 15) / 16;
    }
    pk.assign(total + 1, 0);
    // (every needle owns its words: slices of needles are independent; 15 M symbols one at a time were 10 ms on one thread)
    parallel_slices(nv.n, n_threads, [&](size_t b, size_t e, unsigned) {
        for (size_t p = b; p < e; ++p) {
            const uint8_t *nd = nv.ranks + nv.offsets[p];
            const uint32_t m = (uint32_t)nv.m[p];
            uint32_t *out = pk.data() + pk_off[p];
            for (uint32_t y0 = 0; y0 < m; y0 += 16) {
                uint32_t w = 0;
                const uint32_t lim = m - y0 < 16 ? m - y0 : 16;
                for (uint32_t y = 0; y < lim; ++y)
                    w |= (uint32_t)(nd[y0 + y] & 3) << (2 * y);
                out[y0 / 16] = w;
            }
        }
    });
}

} // namespace spm_hip
