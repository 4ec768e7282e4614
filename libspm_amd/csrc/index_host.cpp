// index_host.cpp -- the host-only part of libspm_hip.so as a library of its own, built with plain g++ (no HIP): the seed
// index build and its self-check.  `make -C libspm_amd/csrc asan` compiles it under -fsanitize=address,undefined for the
// CPU test leg (tests/test_host_sanitizers.py); the product compiles the same header into libspm_hip.so.
#include "comm_protocol.hpp"
#include "index_build.hpp"

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
