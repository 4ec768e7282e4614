// index_host.cpp -- the host-only part of libspm_hip.so as a library of its own, built with plain g++ (no HIP): the seed
// index build and its self-check.  `make -C libspm_amd/csrc asan` compiles it under -fsanitize=address,undefined for the
// CPU test leg (tests/test_host_sanitizers.py); the product compiles the same header into libspm_hip.so.
#include "comm_protocol.hpp"
#include "index_build.hpp"
#include "tables_build.hpp"

extern "C" int spm_hip_host_selftest(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                                     const uint16_t *k, uint32_t sigma, uint64_t *stats)
{
    return spm_hip::host_selftest(algo, ranks_concat, offsets, n_patterns, k, sigma, stats);
}

extern "C" int spm_hip_gatherv_plan(const uint64_t *counts, uint32_t world, uint32_t record_bytes, uint64_t *offsets)
{
    return spm_hip::gatherv_plan(counts, world, record_bytes, offsets);
}

extern "C" int spm_hip_comm_selftest(int world, int root, int scenario, int victim, uint32_t record_bytes, uint64_t seed,
                                     int *detail)
{
    return spm_hip::comm_selftest(world, root, scenario, victim, record_bytes, seed, detail);
}

// (this library only) the match-mask tables of tables_build.hpp checked against their definition, bit by bit: row c of
// needle p has bit (NW * 32 - m + j) set iff pattern[j] == c (Myers; cleared for Shift-Or), every bit above the needle is a
// wildcard, the bottom-aligned copy has bit j.  Returns the number of wrong words.
extern "C" uint64_t spm_host_tables_check(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                                          uint32_t sigma, uint32_t n_threads)
{
    using namespace spm_hip;
    std::vector<int32_t> m(n_patterns), k(n_patterns, 0);
    uint32_t max_m = 0;
    for (uint32_t p = 0; p < n_patterns; ++p) {
        m[p] = (int32_t)(offsets[p + 1] - offsets[p]);
        max_m = std::max(max_m, (uint32_t)m[p]);
    }
    needle_view nv;
    nv.algo = algo;
    nv.n = n_patterns;
    nv.sigma = sigma;
    nv.ranks = ranks_concat;
    nv.offsets = offsets;
    nv.m = m.data();
    nv.k = k.data();
    const uint32_t NW = next_pow2(std::max(1u, (max_m + 31) / 32)), n_groups = std::max(1u, (n_patterns + 63) / 64), rows = sigma + 1;
    brute_tables T;
    build_brute_tables(nv, n_groups, NW, true, n_threads, T);
    std::vector<uint32_t> pk, pk_off;
    if (sigma == 4)
        pack_needles(nv, pk, pk_off);
    uint64_t wrong = 0;
    const bool myers = nv.is_myers();
    for (uint32_t p = 0; p < n_groups * 64; ++p) {
        const uint32_t g = p / 64, l = p % 64, mm = p < n_patterns ? (uint32_t)m[p] : 0, off = NW * 32 - mm;
        for (uint32_t row = 0; row < rows; ++row)
            for (uint32_t w = 0; w < NW; ++w) {
                uint32_t want = 0, want_bot = 0;
                for (uint32_t b = 0; b < 32 && mm; ++b) {
                    const uint32_t bit = w * 32 + b;
                    if (bit < off || (row < sigma && ranks_concat[offsets[p] + bit - off] == row))
                        want |= 1u << b;
                    if (bit < mm && row < sigma && ranks_concat[offsets[p] + bit] == row)
                        want_bot |= 1u << b;
                }
                const size_t at = (((size_t)g * rows + row) * NW + w) * 64 + l;
                const uint32_t got = myers ? T.peq[at] : ~T.peq[at];
                wrong += (mm ? got != want : (myers ? T.peq[at] != 0 : T.peq[at] != 0xFFFFFFFFu)) ? 1 : 0;
                if (!T.verify.empty())
                    wrong += T.verify[at] != want ? 1 : 0;
                if (!T.bot.empty())
                    wrong += T.bot[at] != want_bot ? 1 : 0;
            }
        if (sigma == 4 && p < n_patterns)
            for (uint32_t j = 0; j < mm; ++j)
                wrong += ((pk[pk_off[p] + j / 16] >> (2 * (j % 16))) & 3u) != (ranks_concat[offsets[p] + j] & 3u) ? 1 : 0;
    }
    return wrong;
}
