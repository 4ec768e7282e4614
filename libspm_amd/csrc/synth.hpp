// synth.hpp -- deterministic synthetic inputs of the benchmark configs.
//
// Text:    base(i) = (mix64(seed + (i >> 5)) >> (2*(i & 31))) & 3      -- uniform dna4 ranks
// Needles: copy L bases from a pseudo-random text offset, plant e = p mod (kmax+1) edits, re-trim to L.
// The text is generated directly in HBM (16 GiB never crosses PCIe); needles are generated on the host from the
// same counter-based function.  oracle/spm_oracle.c carries an independent restatement that tests cross-check.
#pragma once

#include "common.hpp"

namespace spm_hip
{

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint8_t synth_base(uint64_t seed, uint64_t i)
{
    return (uint8_t)((mix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3);
}

// One thread writes 32 bases (one mix64 word) as two 16-byte stores; consecutive lanes write consecutive
// 32-byte runs, so a wave stores 2 KiB contiguously.
__global__ __launch_bounds__(256) void synth_text_kernel(uint8_t *__restrict__ out, uint64_t seed,
                                                         uint64_t global_begin, uint64_t n)
{
    // word index w covers global bases [32w, 32w+32); global_begin is a multiple of 32 (host guarantees)
    const uint64_t n_words = (n + 31) / 32;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint64_t r = mix64(seed + (global_begin >> 5) + w);
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t b = (uint32_t)(r >> (8 * q)) & 0xFF; // 4 bases, 2 bits each
            v[q] = (b & 3) | ((b >> 2) & 3) << 8 | ((b >> 4) & 3) << 16 | ((b >> 6) & 3) << 24;
        }
        const uint64_t o = w * 32;
        if (o + 32 <= n) {
            uint4 *dst = reinterpret_cast<uint4 *>(out + o);
            dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
            dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
        } else {
            for (uint64_t i = o; i < n; ++i)
                out[i] = (uint8_t)((v[(i - o) >> 2] >> (8 * ((i - o) & 3))) & 0xFF);
        }
    }
}

inline uint64_t pat_rnd(uint64_t seed_pat, uint32_t p, uint32_t t)
{
    return mix64(seed_pat + ((uint64_t)p << 16) + t);
}

// Needle p: L bases copied from text offset o, with e planted edits (substitute / delete / insert), then padded
// from the following text so that the needle has exactly L bases.  Returns o.
inline uint64_t synth_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                              uint32_t kmax, uint8_t *out)
{
    const uint64_t o = pat_rnd(seed_pat, p, 0) % (n_total - 2ull * L);
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
        const uint8_t b = synth_base(seed_text, o + x);
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

} // namespace spm_hip
