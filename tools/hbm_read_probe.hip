// hbm_read_probe.hip -- attainable streaming-READ bandwidth of one MI355X for the access shapes the seed filter
// can use (16 B per lane, U loads in flight per lane, span vs interleaved mapping, nontemporal or not).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_read_probe tools/hbm_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__device__ __forceinline__ uint4 ld(const uint4 *p)
{
    if (NT) {
        const uint32_t *q = (const uint32_t *)p;
        return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                          __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
    }
    return *p;
}

// MODE 0: spans (each wave streams its own contiguous span, U KiB groups)   MODE 1: interleaved (wave w takes
// group w, w + n_waves, ...: the whole grid sweeps the buffer front to back)
template <int U, bool NT, int MODE>
__global__ __launch_bounds__(1024) void read_kernel(const uint4 *__restrict__ src, uint64_t n_chunks /*1 KiB*/,
                                                    uint64_t span_chunks, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wpw = blockDim.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * wpw + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * wpw;
    uint32_t acc = 0;
    if (MODE == 0) {
        const uint64_t n_spans = (n_chunks + span_chunks - 1) / span_chunks;
        for (uint64_t sp = wave; sp < n_spans; sp += n_waves) {
            const uint64_t c0 = sp * span_chunks, c1 = c0 + span_chunks < n_chunks ? c0 + span_chunks : n_chunks;
            uint4 nxt[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                nxt[u] = c0 + u < c1 ? ld<NT>(src + (c0 + u) * 64 + lane) : make_uint4(0, 0, 0, 0);
            for (uint64_t c = c0; c < c1; c += U) {
                uint4 cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    cur[u] = nxt[u];
                if (c + U < c1) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        nxt[u] = c + U + u < c1 ? ld<NT>(src + (c + U + u) * 64 + lane) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
            }
        }
    } else {
        const uint64_t n_groups = (n_chunks + U - 1) / U;
        uint4 nxt[U];
        uint64_t g = wave;
        if (g < n_groups) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                nxt[u] = g * U + u < n_chunks ? ld<NT>(src + (g * U + u) * 64 + lane) : make_uint4(0, 0, 0, 0);
        }
        for (; g < n_groups; g += n_waves) {
            uint4 cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                cur[u] = nxt[u];
            const uint64_t g2 = g + n_waves;
            if (g2 < n_groups) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    nxt[u] = g2 * U + u < n_chunks ? ld<NT>(src + (g2 * U + u) * 64 + lane) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

template <int U, bool NT, int MODE>
float run(const uint4 *d, uint64_t bytes, int threads, int blocks, uint64_t span, uint32_t *sink, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((read_kernel<U, NT, MODE>), dim3(blocks), dim3(threads), 0, 0, d, bytes / 1024, span, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((read_kernel<U, NT, MODE>), dim3(blocks), dim3(threads), 0, 0, d, bytes / 1024, span, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const uint64_t gib = argc > 1 ? atoll(argv[1]) : 16;
    const uint64_t bytes = gib << 30;
    uint4 *d;
    uint32_t *sink;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 1, bytes));
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, buffer %llu GiB\n", prop.name, cu, (unsigned long long)gib);
    printf("%-6s %-3s %-3s %-8s %-7s %-8s %10s %10s\n", "mode", "U", "NT", "threads", "blk/CU", "span", "ms", "GB/s");
#define RUN(U, NT, MODE, threads, bpc, span)                                                                           \
    do {                                                                                                               \
        float ms = run<U, NT, MODE>(d, bytes, threads, cu * bpc, span, sink, 5);                                       \
        printf("%-6s %-3d %-3d %-8d %-7d %-8d %10.3f %10.1f\n", MODE ? "inter" : "span", U, (int)NT, threads, bpc,     \
               (int)span, ms, bytes / (ms * 1e-3) / 1e9);                                                              \
        fflush(stdout);                                                                                                \
    } while (0)
    for (int bpc : {1, 2}) {
        RUN(4, false, 0, 1024, bpc, 512);
        RUN(4, true, 0, 1024, bpc, 512);
        RUN(8, false, 0, 1024, bpc, 512);
        RUN(8, true, 0, 1024, bpc, 512);
        RUN(4, false, 1, 1024, bpc, 0);
        RUN(4, true, 1, 1024, bpc, 0);
        RUN(8, false, 1, 1024, bpc, 0);
        RUN(8, true, 1, 1024, bpc, 0);
        RUN(2, true, 1, 1024, bpc, 0);
        RUN(1, true, 1, 1024, bpc, 0);
    }
    for (int bpc : {4, 8}) {
        RUN(4, true, 1, 256, bpc, 0);
        RUN(2, true, 1, 256, bpc, 0);
        RUN(4, false, 1, 256, bpc, 0);
        RUN(4, true, 0, 256, bpc, 64);
        RUN(1, true, 1, 256, bpc, 0);
    }
    RUN(4, true, 0, 1024, 1, 16);
    RUN(4, true, 0, 1024, 1, 64);
    RUN(4, true, 0, 1024, 1, 4096);
    return 0;
}
