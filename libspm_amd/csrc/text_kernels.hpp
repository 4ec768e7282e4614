// text_kernels.hpp -- kernels of the haystack entry points (text.hip): the 2-bit shadow and the validation of borrowed buffers.
#pragma once

#include "pack.hpp"

namespace spm_hip
{

// ---------------------------------------------------------------------------------------------------
// Optional 2-bit shadow of a dna4 haystack (spm_hip_text_pack): 16 symbols per uint32, same bit order as pack16.
// A text that is scanned many times (one reference, many needle batches) is then streamed at a quarter of the
// HBM traffic; hits are identical.  text_pack_kernel builds it in one pass and flags symbols outside {0..3}.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_pack_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                        uint32_t *__restrict__ packed, uint64_t n_words,
                                                        unsigned int *bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned int any_bad = 0;
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += stride) {
        const uint4 v = load_text16(text, wi * 16, n);
        any_bad |= (v.x | v.y | v.z | v.w) & 0xFCFCFCFCu;
        packed[wi] = pack16(v);
    }
    if (any_bad)
        atomicOr(bad, 1u);
}

// A borrowed haystack (spm_hip_text_wrap) is read once to make sure every symbol is a rank < sigma: the filter's 2-bit
// packing would fold a stray byte into its neighbours' codes and could then miss an occurrence the brute-force engine
// reports.
__global__ __launch_bounds__(256) void text_validate_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t sigma,
                                                            unsigned int *bad)
{
    const uint64_t n_q = (n + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned int any_bad = 0;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_q; q += stride) {
        const uint4 v = load_text16(text, q * 16, n); // bytes past n read as 0
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                any_bad |= ((w[i] >> (8 * b)) & 0xFFu) >= sigma ? 1u : 0u;
    }
    if (any_bad)
        atomicOr(bad, 1u);
}

} // namespace spm_hip
