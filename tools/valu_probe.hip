// valu_probe.hip -- measured integer-VALU issue peak of one MI355X (the brute-force engines are
// bounded by VALU issue, not by HBM).  Every lane runs chains of independent 32-bit VALU instructions; the figure
// reported is lane-ops/s = lanes x instructions / time.
//   hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

// KIND 0: v_add_u32, 1: v_bitop3_b32 (the 3-input logic op the Myers step is built from), 2: v_alignbit_b32
template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t *out, uint32_t iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i)
        a[i] = seed * (threadIdx.x + 1) + i;
    const uint32_t b = seed ^ blockIdx.x, c = seed + 77;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0)
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 1)
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
                else
                    asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i)
        s ^= a[i];
    if (s == 0x12345678u)
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static double run(const char *name, uint32_t *out, int waves_per_simd)
{
    const int grid = 256 * waves_per_simd; // 256 CUs x 4 SIMDs: one 256-thread workgroup = 4 waves = one per SIMD
    const uint32_t iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, 100u, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 3u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
            best = ms;
    }
    const double ops = (double)grid * 256 * (double)iters * 64.0;
    const double rate = ops / (best * 1e-3);
    std::printf("{\"probe\": \"valu\", \"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"lane_ops_per_s\": %.4e, "
                "\"lane_ops_per_cycle_per_cu_at_2400MHz\": %.1f}\n",
                name, waves_per_simd, best, rate, rate / 256.0 / 2.4e9);
    return rate;
}

int main()
{
    uint32_t *out = nullptr;
    CHECK(hipMalloc(&out, 256u * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", out, w);
        run<1>("v_bitop3_b32", out, w);
        run<2>("v_alignbit_b32", out, w);
    }
    CHECK(hipFree(out));
    return 0;
}
