// synth.hpp -- deterministic synthetic inputs of the benchmark configs.
//
// Text:    base(i) = (mix64(seed + (i >> 5)) >> (2*(i & 31))) & 3      -- uniform dna4 ranks
// Needles: copy L bases from a pseudo-random text offset, plant e = p mod (kmax+1) edits, re-trim to L.
// The text is generated directly in HBM (16 GiB never crosses PCIe); needles are generated on the host from the
// same counter-based function.  oracle/spm_oracle.c carries an independent restatement that tests cross-check.
#pragma once

#include "common.hpp"
#include "hd.hpp" // mix64

namespace spm_hip
{

__host__ __device__ inline uint8_t synth_base(uint64_t seed, uint64_t i)
{
    return (uint8_t)((mix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3);
}

// ---- repeat-rich text (bench workload c3r) -----------------------------------------------------------------------
// Uniform text with a stated fraction `ppm` (parts per million) of its bases inside low-complexity stretches, the kind
// of sequence real genomes have and a q-gram filter meets as bursts of seed hits.  The text is cut into 1024-base
// blocks; block b holds ONE stretch with probability ppm * 1024 / 136 / 10^6 (136 = mean stretch length), fully inside
// the block:  length 16..256 (multiples of 16), and
//   kind 0  tandem repeat: unit of 1..6 bases repeated, 1 base in 64 replaced by another one (impure microsatellite)
//   kind 1  low complexity: one dominant base with probability 7/8, a uniform base otherwise
// Everything is a pure function of (seed, ppm, position), so any slice can be regenerated on host and device alike
// (oracle/spm_oracle.c carries an independent restatement).
struct repeat_stretch
{
    bool present;
    uint32_t off, len; // inside the block
    uint64_t r;        // the block's hash (kind, unit length, ...)
};

__host__ __device__ inline repeat_stretch repeat_block(uint64_t seed, uint32_t ppm, uint64_t block)
{
    repeat_stretch s;
    const uint64_t r = mix64((seed ^ 0x7E9EA7ull) + block * 0x9E3779B97F4A7C15ull);
    // P(present) = ppm * 1024 / 136 / 1e6, compared on 32 bits
    const uint64_t thr = ((uint64_t)ppm * 1024ull * 4294967296ull) / (136ull * 1000000ull);
    s.present = (r & 0xFFFFFFFFull) < thr;
    s.len = 16u * (1u + (uint32_t)((r >> 32) & 15));
    s.off = (uint32_t)((r >> 36) & 0xFFFF) % (1024u - s.len + 1u);
    s.r = r;
    return s;
}

__host__ __device__ inline uint8_t repeat_stretch_base(const repeat_stretch &s, uint32_t j)
{
    const uint64_t r2 = mix64(s.r);
    const uint32_t x = (uint32_t)(mix64(r2 + 1 + (j >> 3)) >> (8 * (j & 7))) & 0xFF; // one random byte per base
    if (((s.r >> 52) & 1) == 0) {
        const uint32_t u = 1u + (uint32_t)((s.r >> 53) % 6);
        uint32_t sym = (uint32_t)(r2 >> (2 * (j % u))) & 3u;
        if ((x & 63u) == 0)
            sym = (sym + 1u + (x >> 6) % 3u) & 3u;
        return (uint8_t)sym;
    }
    const uint32_t dom = (uint32_t)(s.r >> 53) & 3u;
    return (uint8_t)((x & 7u) != 0 ? dom : ((x >> 3) & 3u));
}

__host__ __device__ inline uint8_t repeat_base(uint64_t seed, uint32_t ppm, uint64_t i)
{
    const repeat_stretch s = repeat_block(seed, ppm, i >> 10);
    const uint32_t in = (uint32_t)(i & 1023);
    if (s.present && in >= s.off && in < s.off + s.len)
        return repeat_stretch_base(s, in - s.off);
    return synth_base(seed, i);
}

inline uint64_t pat_rnd(uint64_t seed_pat, uint32_t p, uint32_t t)
{
    return mix64(seed_pat + ((uint64_t)p << 16) + t);
}

// Needle p: L bases copied from text offset o, with e planted edits (substitute / delete / insert), then padded
// from the following text so that the needle has exactly L bases.  Returns o.
template <typename BaseFn>
inline uint64_t synth_pattern_at(BaseFn base, uint64_t o, uint64_t seed_pat, uint32_t p, uint32_t L, uint32_t kmax,
                                 uint8_t *out)
{
    const uint32_t e = p % (kmax + 1);
    const uint32_t ne = e < 64 ? e : 64;
    uint32_t epos[64], etype[64], ebase[64];
    for (uint32_t i = 0; i < ne; ++i) {
        uint32_t pos = (uint32_t)(pat_rnd(seed_pat, p, 1 + 2 * i) % L);
        bool clash = true;
        while (clash) {
            clash = false;
            for (uint32_t j = 0; j < i; ++j)
                if (epos[j] == pos)
                    clash = true;
            if (clash)
                pos = (pos + 1) % L;
        }
        const uint64_t r = pat_rnd(seed_pat, p, 2 + 2 * i);
        epos[i] = pos;
        etype[i] = (uint32_t)(r % 3);
        ebase[i] = (uint32_t)((r >> 8) & 3);
    }
    uint32_t produced = 0;
    for (uint64_t x = 0; produced < L; ++x) {
        const uint8_t b = base(o + x);
        int hit = -1;
        if (x < L)
            for (uint32_t i = 0; i < ne; ++i)
                if (epos[i] == x)
                    hit = (int)i;
        if (hit < 0) {
            out[produced++] = b;
        } else if (etype[hit] == 0) {
            out[produced++] = (uint8_t)((b + 1 + ebase[hit] % 3) & 3);
        } else if (etype[hit] == 1) {
            // deleted
        } else {
            out[produced++] = (uint8_t)ebase[hit];
            if (produced < L)
                out[produced++] = b;
        }
    }
    return o;
}

inline uint64_t synth_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                              uint32_t kmax, uint8_t *out)
{
    const uint64_t o = pat_rnd(seed_pat, p, 0) % (n_total - 2ull * L);
    return synth_pattern_at([&](uint64_t i) { return synth_base(seed_text, i); }, o, seed_pat, p, L, kmax, out);
}

// Needles of the repeat-rich workload: as above over repeat_base -- cut at uniformly random positions, so they cross a
// stretch at the natural rate (about (136 + L) / 1024 x the block probability: 1.7 % for L = 100 at ppm = 10 000) -- and
// on top of that every needle with p % every == every - 1 is cut ACROSS a stretch on purpose (the first block at or
// after a pseudo-random one that holds a stretch), overlapping it by 1 .. min(L, len) bases.
inline uint64_t synth_repeat_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                                     uint32_t kmax, uint32_t ppm, uint32_t every, uint8_t *out)
{
    uint64_t o = pat_rnd(seed_pat, p, 0) % (n_total - 2ull * L);
    if (every > 0 && p % every == every - 1 && ppm > 0) {
        const uint64_t n_blocks = n_total >> 10;
        uint64_t b = n_blocks ? pat_rnd(seed_pat, p, 200) % n_blocks : 0;
        for (uint32_t tries = 0; tries < 65536 && n_blocks; ++tries, b = (b + 1) % n_blocks) {
            const repeat_stretch s = repeat_block(seed_text, ppm, b);
            if (!s.present)
                continue;
            const uint64_t start = (b << 10) + s.off;
            const uint64_t shift = pat_rnd(seed_pat, p, 201) % (s.len + L - 1); // needle begin = start - (L-1) + shift
            if (start + shift + 1 >= L && start + shift + 1 - L < n_total - 2ull * L) {
                o = start + shift + 1 - L;
                break;
            }
        }
    }
    return synth_pattern_at([&](uint64_t i) { return repeat_base(seed_text, ppm, i); }, o, seed_pat, p, L, kmax, out);
}

} // namespace spm_hip
