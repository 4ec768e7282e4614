// l2_gather_probe.hip -- what a streaming kernel may spend on element-granular table lookups that miss LDS:
//   (1) random gathers of 4 / 16 bytes from a table of T bytes (L2 / Infinity Cache / HBM resident), NIF loads in flight per
//       lane, a fraction of the lanes active -- lane-gathers per second;
//   (2) the same gathers issued from inside a 16-byte-per-lane nontemporal stream over a large buffer (the seed filter's
//       access shape): G gathers per lane and 1 KiB chunk with lane probability p, consumed one group later.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/l2_gather_probe tools/l2_gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}

template <typename E>
__device__ __forceinline__ uint32_t fold(const E &e);
template <>
__device__ __forceinline__ uint32_t fold<uint32_t>(const uint32_t &e) { return e; }
template <>
__device__ __forceinline__ uint32_t fold<uint4>(const uint4 &e) { return e.x ^ e.y ^ e.z ^ e.w; }

// (1) pure gather
template <typename E, int NIF>
__global__ __launch_bounds__(1024) void gather_kernel(const E *__restrict__ tab, uint32_t mask, uint32_t iters,
                                                      uint32_t active_mod, uint32_t *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = mix32(tid * 2654435761u + 12345u);
    uint32_t acc = 0;
    const bool act = (s % 16u) < active_mod; // a fixed subset of the lanes takes part
    if (act) {
        for (uint32_t it = 0; it < iters; it += NIF) {
            E v[NIF];
#pragma unroll
            for (int j = 0; j < NIF; ++j) {
                s = s * 1664525u + 1013904223u;
                v[j] = tab[mix32(s) & mask];
            }
#pragma unroll
            for (int j = 0; j < NIF; ++j)
                acc += fold<E>(v[j]);
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

__device__ __forceinline__ uint4 ldnt(const uint4 *p)
{
    const uint32_t *q = (const uint32_t *)p;
    return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
                      __builtin_nontemporal_load(q + 3));
}

// (2) stream + gather: U chunks per group; per chunk and lane G gather slots, each taken with probability p16/16
template <typename E, int U, int G>
__global__ __launch_bounds__(1024) void stream_gather_kernel(const uint4 *__restrict__ src, uint64_t n_chunks,
                                                             uint64_t span_chunks, const E *__restrict__ tab, uint32_t mask,
                                                             uint32_t p256, unsigned long long *head, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const uint64_t n_spans = (n_chunks + span_chunks - 1) / span_chunks;
    E pend[U * G];
    bool pend_on[U * G];
#pragma unroll
    for (int i = 0; i < U * G; ++i)
        pend_on[i] = false;
    for (;;) {
        unsigned long long t = 0;
        if (lane == 0)
            t = atomicAdd(head, 1ull);
        const uint64_t sp = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)t);
        if (sp >= n_spans)
            break;
        const uint64_t c0 = sp * span_chunks, c1 = c0 + span_chunks < n_chunks ? c0 + span_chunks : n_chunks;
        uint4 nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            nxt[u] = ldnt(src + (c0 + u) * 64 + lane);
        for (uint64_t c = c0; c + U <= c1; c += U) {
            uint4 cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                cur[u] = nxt[u];
            const uint64_t pf = c + 2 * U <= c1 ? c + U : c;
#pragma unroll
            for (int u = 0; u < U; ++u)
                nxt[u] = ldnt(src + (pf + u) * 64 + lane);
            __builtin_amdgcn_sched_barrier(0);
            // consume the gathers of the previous group, issue this group's
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (pend_on[u * G + g])
                        acc += fold<E>(pend[u * G + g]);
                    const uint32_t w = g == 0 ? cur[u].x : (g == 1 ? cur[u].y : (g == 2 ? cur[u].z : cur[u].w));
                    const uint32_t h = mix32(w + (uint32_t)c * 7919u + lane);
                    pend_on[u * G + g] = (h >> 24) < p256;
                    if (pend_on[u * G + g])
                        pend[u * G + g] = tab[(h * 0x9E3779B1u >> 3) & mask];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc ^= cur[u].x + cur[u].y + cur[u].z + cur[u].w;
        }
    }
#pragma unroll
    for (int i = 0; i < U * G; ++i)
        if (pend_on[i])
            acc += fold<E>(pend[i]);
    if (acc == 0x12345678u)
        sink[0] = acc;
}

template <typename E, int NIF>
static void run_gather(const void *tab, size_t tab_bytes, uint32_t active16, int n_cu, uint32_t *sink)
{
    const uint32_t n = (uint32_t)(tab_bytes / sizeof(E));
    const uint32_t iters = 2048;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_kernel<E, NIF>), dim3(n_cu), dim3(1024), 0, 0, (const E *)tab, n - 1, iters, active16, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double lanes = (double)n_cu * 1024 * active16 / 16.0;
    const double rate = lanes * iters / (best * 1e-3);
    printf("gather  table %7.2f MB  elem %2zu B  nif %d  active %2u/16  %8.3f ms  %.3e lane-gathers/s  %.1f GB/s useful\n",
           tab_bytes / 1048576.0, sizeof(E), NIF, active16, best, rate, rate * sizeof(E) / 1e9);
}

template <typename E, int U, int G>
static void run_stream(const uint4 *src, uint64_t bytes, const void *tab, size_t tab_bytes, uint32_t p256, int n_cu,
                       unsigned long long *head, uint32_t *sink)
{
    const uint32_t n = (uint32_t)(tab_bytes / sizeof(E));
    const uint64_t n_chunks = bytes / 1024;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(head, 0, 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_gather_kernel<E, U, G>), dim3(n_cu), dim3(1024), 0, 0, src, n_chunks, (uint64_t)256,
                           (const E *)tab, n - 1, p256, head, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double gathers = (double)bytes / 16.0 * G * p256 / 256.0;
    printf("stream  %5.1f GiB  U %d  G %d  p %3u/256  table %6.2f MB elem %2zu B  %7.3f ms  %.2f TB/s stream  %.3e gathers (%.3e /s)\n",
           bytes / 1073741824.0, U, G, p256, tab_bytes / 1048576.0, sizeof(E), best, bytes / (best * 1e-3) / 1e12, gathers,
           gathers / (best * 1e-3));
}

int main(int argc, char **argv)
{
    const uint64_t stream_bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096ull) << 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, n_cu);
    const size_t max_tab = 64u << 20;
    void *tab;
    CK(hipMalloc(&tab, max_tab));
    CK(hipMemset(tab, 0x5A, max_tab));
    uint32_t *sink;
    CK(hipMalloc(&sink, 64));
    unsigned long long *head;
    CK(hipMalloc(&head, 64));
    for (size_t mb : {1, 2, 4, 16, 64}) {
        run_gather<uint32_t, 1>(tab, mb << 20, 16, n_cu, sink);
        run_gather<uint32_t, 4>(tab, mb << 20, 16, n_cu, sink);
        run_gather<uint4, 1>(tab, mb << 20, 16, n_cu, sink);
        run_gather<uint4, 4>(tab, mb << 20, 16, n_cu, sink);
        run_gather<uint4, 4>(tab, mb << 20, 4, n_cu, sink);
    }
    uint4 *src;
    CK(hipMalloc(&src, stream_bytes));
    CK(hipMemset(src, 0x33, stream_bytes));
    for (size_t mb : {1, 2, 4, 8}) {
        for (uint32_t p : {0u, 16u, 64u, 128u, 256u}) {
            run_stream<uint4, 4, 1>(src, stream_bytes, tab, mb << 20, p, n_cu, head, sink);
            if (mb != 2 && mb != 4)
                break;
        }
    }
    for (uint32_t p : {64u, 256u})
        run_stream<uint32_t, 4, 1>(src, stream_bytes, tab, 2u << 20, p, n_cu, head, sink);
    for (uint32_t p : {128u, 256u})
        run_stream<uint4, 4, 2>(src, stream_bytes, tab, 2u << 20, p, n_cu, head, sink);
    return 0;
}
