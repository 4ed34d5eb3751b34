// Micro-benchmark: does one wave overlap its own VALU work with its MFMAs (matrix pipe busy 32 cycles per
// v_mfma_f32_16x16x4_f32)?  NV independent v_fma_f32 / v_exp_f32 are placed behind every MFMA of a 2-chain stream.
// build: hipcc --offload-arch=gfx950 -O3 coissue_bench.hip -o coissue_bench
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>          // KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_pk_fma_f32
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int reps) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    float v[12];
    float2 w[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { v[i] = 0.5f + i; w[i] = make_float2(0.1f * i, 0.2f); }
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(y));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % 12]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = a[0] + b[0];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i] + w[i].x;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int KIND>
void run(int waves) {
    const int grid = 256, reps = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, grid * 1024 * sizeof(float));
    hipMalloc(&cyc, grid * 16 * sizeof(long long));
    hipMemset(cyc, 0, grid * 16 * sizeof(long long));
    hipLaunchKernelGGL((k<NV, KIND>), dim3(grid), dim3(64 * waves), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 16);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; int n = 0;
    for (auto c : h) if (c) { mean += c; ++n; if (c > mx) mx = c; }
    printf("kind %d  %2d VALU per MFMA, %2d wave(s)/CU: mean %6.1f max %6.1f cycles per MFMA per wave -> pipe use (by max) %5.1f%%\n", KIND, NV, waves,
           mean / n / (reps * 16.0), mx / (reps * 16.0), 100.0 * 32.0 * (waves / 4.0) / (mx / (reps * 16.0)));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>(4); run<0, 0>(8);
    run<4, 0>(4); run<4, 0>(8); run<4, 0>(12); run<4, 0>(16);
    run<6, 0>(4); run<6, 0>(8); run<6, 0>(12); run<6, 0>(16);
    run<8, 0>(8); run<8, 0>(12); run<8, 0>(16);
    run<2, 1>(4); run<2, 1>(8); run<2, 1>(12); run<2, 1>(16);
    run<4, 2>(8); run<4, 2>(12);
    return 0;
}
