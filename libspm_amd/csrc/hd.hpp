// hd.hpp -- what host-only translation units (g++: the index build, its sanitizer build) and device code share.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define SPM_HD __host__ __device__
#else
#define SPM_HD
#endif

namespace spm_hip
{

// 16 bytes, the layout of HIP's uint4: what the host builds and the kernels read as uint4
struct u32x4
{
    uint32_t x, y, z, w;
};

SPM_HD inline uint64_t mix64(uint64_t z) // splitmix64 finaliser
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace spm_hip
