// pack.hpp -- device helpers every kernel that reads the 1-byte text shares: 16 text bytes -> one 2-bit-packed word (dna4,
// dna5, dna15), guarded 16-byte loads, streaming (nontemporal) loads.
#pragma once

#include "common.hpp"

namespace spm_hip
{

// text bytes 16*lane .. 16*lane+15 of a chunk -> 32-bit word, base i at bits 2i..2i+1
__device__ __forceinline__ uint32_t pack16(const uint4 v)
{
    const uint32_t W = 0x40100401u; // byte weights 1,4,16,64
    const uint32_t p0 = __builtin_amdgcn_udot4(v.x, W, 0u, false);
    const uint32_t p1 = __builtin_amdgcn_udot4(v.y, W, 0u, false);
    const uint32_t p2 = __builtin_amdgcn_udot4(v.z, W, 0u, false);
    const uint32_t p3 = __builtin_amdgcn_udot4(v.w, W, 0u, false);
    return p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}

// dna5 haystacks (seqan3 ranks A0 C1 G2 N3 T4): same 2-bit word with T folded onto 3, plus a 16-bit mask of the N
// positions; a window that contains an N cannot equal any (N-free) key and is dropped.  ~7 VALU per dword.
__device__ __forceinline__ uint32_t pack16_dna5(const uint4 v, uint32_t &nmask)
{
    const uint32_t W = 0x40100401u, WN = 0x08040201u;
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t code = 0;
    nmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t isT = (x[i] >> 2) & 0x01010101u;            // byte == 4
        const uint32_t isN = x[i] & (x[i] >> 1) & 0x01010101u;     // byte == 3
        code |= __builtin_amdgcn_udot4(x[i] - isT, W, 0u, false) << (8 * i);
        nmask |= __builtin_amdgcn_udot4(isN, WN, 0u, false) << (4 * i);
    }
    return code;
}

// dna15 haystacks (seqan3 ranks A0 B1 C2 D3 G4 H5 K6 M7 N8 R9 S10 T11 V12 W13 Y14): A, C, G, T -> 0..3, every
// ambiguity code marked like dna5's N (a window that holds one equals no key).  SWAR on the four bytes of a dword:
// a key symbol is T (11) or an even rank <= 4; its code is rank >> 1 (T: 3).  ~14 VALU per dword.
__device__ __forceinline__ uint32_t pack16_dna15(const uint4 v, uint32_t &nmask)
{
    const uint32_t W = 0x40100401u, WN = 0x08040201u;
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t code = 0;
    nmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = x[i] ^ 0x0B0B0B0Bu;                                          // byte == 11 <=> zero byte
        const uint32_t isT = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu) >> 7; // exact per-byte zero test
        const uint32_t odd = x[i] & 0x01010101u;
        const uint32_t big = ((x[i] + 0x7B7B7B7Bu) & 0x80808080u) >> 7;                   // byte >= 5
        const uint32_t amb = (odd | big) & ~isT;
        const uint32_t c2 = (((x[i] >> 1) & 0x03030303u) & ~(isT * 3u)) | (isT * 3u);
        code |= __builtin_amdgcn_udot4(c2 & ~(amb * 3u), W, 0u, false) << (8 * i);
        nmask |= __builtin_amdgcn_udot4(amb, WN, 0u, false) << (4 * i);
    }
    return code;
}

template <int SIG>
__device__ __forceinline__ uint32_t pack16_sig(const uint4 v, uint32_t &nmask)
{
    if constexpr (SIG == 5)
        return pack16_dna5(v, nmask);
    else if constexpr (SIG == 15)
        return pack16_dna15(v, nmask);
    else {
        nmask = 0;
        return pack16(v);
    }
}

__device__ __forceinline__ uint4 load_text16(const uint8_t *text, uint64_t idx, uint64_t limit)
{
    // idx % 16 == 0.  Bytes at or beyond `limit` read as 0.
    if (idx + 16 <= limit)
        return *reinterpret_cast<const uint4 *>(text + idx);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; ++b)
        if (idx + b < limit)
            w[b >> 2] |= (uint32_t)text[idx + b] << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool NT>
__device__ __forceinline__ uint4 load16_stream(const uint8_t *p)
{
    if (NT) {
        // streamed once: nontemporal keeps it from displacing the key table in L2 and measures +13 % on a pure
        // 16 GiB read (tools/hbm_read_probe: 7.0 vs 6.2 TB/s)
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                          __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
    }
    return *reinterpret_cast<const uint4 *>(p);
}

} // namespace spm_hip
