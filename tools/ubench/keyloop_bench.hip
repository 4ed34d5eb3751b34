// Micro-benchmark: the triangle-attention key loop (ta_keyloop) alone -- K / V^T resident in LDS, NTQ query tiles
// per wave, NW waves per CU -- to separate per-wave latency from matrix-pipe throughput.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../protein_redesign_amd/csrc -DTRI_SRC='"../../protein_redesign_amd/csrc/prd_tri.hip"' keyloop_bench.hip -o keyloop_bench
#include <cstdio>
#include <vector>
#include TRI_SRC

template <int NTQ, bool MASKED, int NW>
__global__ __launch_bounds__(NW * 64) void kl_kernel(float* out, long long* cyc, int npad, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Kl = smem;
    float* Vt = Kl + npad * KP;
    float* kadd = Vt + 16 * (npad + 4);
    for (int i = threadIdx.x; i < npad * KP + 16 * (npad + 4); i += NW * 64) smem[i] = 0.01f * ((i * 37) % 101 - 50);
    for (int i = threadIdx.x; i < npad; i += NW * 64) kadd[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ql = lane & 15, g4 = lane >> 4;
    float4 qf[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) qf[t] = make_float4(0.01f * ql, 0.02f * g4, 0.03f * t, 0.01f * wave);
    float acc = 0.f;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f32x4 o[NTQ];
        float l[NTQ];
        ta_keyloop<NTQ, MASKED>(Kl, Vt, kadd, qf, npad, ql, g4, o, l);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) { acc += o[t][0] / l[t]; qf[t].x += acc * 1e-30f; }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * NW * 64 + threadIdx.x] = acc;
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[gridDim.x * NW] = r1 - r0;
}

template <int NTQ, bool MASKED, int NW>
void run(int npad, int reps) {
    const int grid = 256;
    float* out;
    long long* cyc;
    hipMalloc(&out, grid * NW * 64 * sizeof(float));
    hipMalloc(&cyc, (grid * NW + 1) * sizeof(long long));
    const size_t lds = 100 * 1024;               // one workgroup per CU
    hipFuncSetAttribute((const void*)kl_kernel<NTQ, MASKED, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kl_kernel<NTQ, MASKED, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, cyc, npad, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kl_kernel<NTQ, MASKED, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, cyc, npad, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * NW + 1);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid * NW; ++i) mean += (double)h[i];
    mean /= grid * NW;
    const double ghz = (double)h[0] / ((double)h[grid * NW] * 10.0);     // s_memrealtime ticks at 100 MHz
    printf("   per-wave Mcycles:");
    for (int w = 0; w < NW; ++w) { double m = 0; for (int g = 0; g < grid; ++g) m += (double)h[g * NW + w]; printf(" %.2f", m / grid / 1e6); }
    printf("\n");
    const double blocks = (double)reps * (npad / 32);
    const double per_block = mean / blocks;
    const double mfma_cycles = NTQ * 16 * 32.0 * (NW / 4.0);           // matrix-pipe cycles per SIMD per block round
    const double flops = (double)grid * NW * blocks * NTQ * 16 * 2048.0;
    printf("NTQ=%d masked=%d waves/CU=%2d  %7.3f ms  %6.1f TF/s  cycles/block/wave %7.1f  pipe use %5.1f%%  counter %.2f GHz\n", NTQ, (int)MASKED, NW, ms,
           flops / (ms * 1e-3) / 1e12, per_block, 100.0 * mfma_cycles / per_block, ghz);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int npad = 320, reps = 200;
    run<1, false, 12>(npad, reps); run<1, false, 16>(npad, reps);
    run<2, false, 12>(npad, reps); run<2, false, 16>(npad, reps);
    run<1, true, 16>(npad, reps);
    return 0;
}
