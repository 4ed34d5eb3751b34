// Do MFMA (32x32x16 f16) and VALU work overlap on one SIMD of gfx950?  Per loop iteration: 7 MFMAs (two accumulators) and
// NV VALU instructions (plain v_add_f32 or transcendental v_exp_f32), either in the SAME wave (interleaved 1 MFMA : NV/7 VALU)
// or split over waves (even waves MFMA only, odd waves VALU only).  Time per iteration per SIMD = slowest wave.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/overlap_bench tools/ubench/overlap_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__
DEV f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// KIND 0: v_add_f32, 1: v_exp_f32.  NVG = VALU instructions per MFMA gap (7 gaps).  SPLIT: 0 same wave, 1 role per wave parity
template <int KIND, int NVG, int SPLIT>
__global__ __launch_bounds__(1024) void bench(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 o0, o1;
    float a[16];
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; a[e] = -0.01f * (e + lane) - 0.3f; }
    u32x4 x = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, y = {0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u};
    const bool do_m = SPLIT == 0 || (wave & 1) == 0, do_v = SPLIT == 0 || (wave & 1) == 1;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            if (do_m) {
                if (g & 1) o1 = mfma_h(x, y, o1); else o0 = mfma_h(x, y, o0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) {
#pragma unroll
                for (int v = 0; v < NVG; ++v) {
                    const int i = (g * NVG + v) & 15;
                    if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 5) & 15]));
                    else asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += a[e] + o0[e] + o1[e];
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x * 16 + wave] = c1 - c0;
}

template <int KIND, int NVG, int SPLIT>
static void run() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 8192);
    const int iters = 2000;
    printf("%s x %2d per gap (%3d per 7 MFMAs), %s:", KIND ? "v_exp_f32" : "v_add_f32", NVG, 7 * NVG, SPLIT ? "MFMA waves | VALU waves" : "same wave              ");
    for (int threads : {256, 512, 1024}) {
        if (SPLIT && threads == 256) { printf("        -"); continue; }
        hipLaunchKernelGGL((bench<KIND, NVG, SPLIT>), dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipLaunchKernelGGL((bench<KIND, NVG, SPLIT>), dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 256; ++b) { double mx = 0; for (int w = 0; w < threads / 64; ++w) { double v = (double)h[b * 16 + w] / iters; if (v > mx) mx = v; } s += mx; }
        // work per SIMD per iteration: (threads/256) waves, of which all (same wave) or half (split) do each kind
        printf("  %7.1f", s / 256);
    }
    printf("   cycles per iteration per SIMD at 1 / 2 / 4 waves per SIMD\n");
    hipFree(d); hipFree(sink);
}

int main() {
    run<0, 0, 0>();
    run<0, 4, 0>(); run<0, 8, 0>(); run<0, 16, 0>();
    run<1, 2, 0>(); run<1, 4, 0>();
    run<0, 8, 1>(); run<0, 16, 1>(); run<1, 4, 1>();
    return 0;
}
