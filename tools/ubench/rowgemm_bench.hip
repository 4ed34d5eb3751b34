// Micro-benchmark: sustained MFMA rate of the rowgemm building block (weights in LDS, row data in registers).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../protein_redesign_amd/csrc rowgemm_bench.hip -o rowgemm_bench
#include <cstdio>
#include <vector>
#include "prd_common.h"

template <int NB, int NW>
__global__ __launch_bounds__(NW * 64) void bench_kernel(float* out, const float* w, int reps) {
    constexpr int K = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stage_weight_cll<K>(smem, w, NB * 32, K, threadIdx.x, NW * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    float x[K / 2];
#pragma unroll
    for (int s = 0; s < K / 2; ++s) x[s] = 0.001f * (lane + s);
    f32x16 acc[NB];
    zero_acc(acc);
    for (int it = 0; it < reps; ++it) {
        rowgemm<K, NB>(smem, x, acc, r, hi);
        x[it & 31] += acc[0][0] * 1e-30f;      // keep a dependence so nothing is hoisted
    }
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[nb][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NB, int NW>
void run(int reps) {
    float *out, *w;
    const int grid = 256;
    hipMalloc(&out, grid * NW * 64 * sizeof(float));
    hipMalloc(&w, NB * 32 * 64 * sizeof(float));
    hipMemset(w, 0, NB * 32 * 64 * sizeof(float));
    const size_t lds = NB * 32 * 68 * sizeof(float);
    hipFuncSetAttribute((const void*)bench_kernel<NB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench_kernel<NB, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, w, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench_kernel<NB, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, w, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * reps * NB * 32.0 * (2.0 * 32 * 32 * 2);
    printf("NB=%d waves/CU=%2d  %8.3f ms  %7.1f TF/s\n", NB, NW, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(w);
}

int main() {
    run<1, 4>(2000); run<1, 8>(2000); run<1, 12>(2000); run<1, 16>(2000);
    run<2, 4>(1000); run<2, 8>(1000); run<2, 12>(1000);
    run<4, 4>(500); run<4, 8>(500); run<4, 12>(500);
    run<8, 4>(250); run<8, 8>(250);
    return 0;
}
