// synth_kernels.hpp -- the synthetic texts generated directly in HBM (text.hip; the functions they evaluate: synth.hpp)
#pragma once

#include "synth.hpp"

namespace spm_hip
{

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

// one thread per 16 bases (one 16-byte store); a stretch touches few of them, the rest is the uniform generator
__global__ __launch_bounds__(256) void synth_repeat_text_kernel(uint8_t *__restrict__ out, uint64_t seed, uint32_t ppm,
                                                                uint64_t global_begin, uint64_t n)
{
    const uint64_t n_q = (n + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_q; q += stride) {
        const uint64_t g = global_begin + q * 16; // global_begin % 32 == 0: the 16 bases share one mix64 word and one block
        const uint64_t word = mix64(seed + (g >> 5)) >> (2 * (g & 31));
        const repeat_stretch s = repeat_block(seed, ppm, g >> 10);
        const uint32_t in0 = (uint32_t)(g & 1023);
        uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t b = (uint32_t)(word >> (2 * i)) & 3u;
            const uint32_t in = in0 + i;
            if (s.present && in >= s.off && in < s.off + s.len)
                b = repeat_stretch_base(s, in - s.off);
            v[i >> 2] |= b << (8 * (i & 3));
        }
        const uint64_t o = q * 16;
        if (o + 16 <= n) {
            *reinterpret_cast<uint4 *>(out + o) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            for (uint64_t i = o; i < n; ++i)
                out[i] = (uint8_t)((v[(i - o) >> 2] >> (8 * ((i - o) & 3))) & 0xFF);
        }
    }
}

} // namespace spm_hip
