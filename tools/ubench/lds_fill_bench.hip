// How fast can a workgroup bring a weight image (L2-resident, the same for every workgroup) into its LDS?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ubench/lds_fill_bench.hip -o tools/ubench/lds_fill_bench
// (a) register staging: global_load_dwordx4 -> ds_write_b128 (what the row kernels' prologues do, minus the fp16 split);
// (b) LDS-DMA: global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPRs, no ds_write);
// for 16 .. 151 KB per workgroup, 256 workgroups of 256 / 512 / 768 threads.  Prints cycles (s_memtime) per workgroup, GB/s per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void fill_regs(const float4* __restrict__ src, int n16, unsigned long long* out, float* sink) {
    extern __shared__ float4 sm[];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = threadIdx.x; i < n16; i += blockDim.x) sm[i] = src[i];
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (sm[(threadIdx.x * 7) % n16].x == 123.456f) sink[0] = 1.f;
}

__global__ void fill_dma(const float4* __restrict__ src, int n16, unsigned long long* out, float* sink) {
    extern __shared__ float4 sm[];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wave; c * 64 < n16; c += nw) {          // one 1 KiB piece per wave instruction: LDS dest = wave-uniform base + lane * 16
        const int i = c * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (i < n16 ? i : 0)),
                                         (__attribute__((address_space(3))) void*)(sm + c * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (sm[(threadIdx.x * 7) % n16].x == 123.456f) sink[0] = 1.f;
}

int main() {
    const int maxb = 152 * 1024;
    float4* src; unsigned long long* out; float* sink;
    hipMalloc(&src, maxb); hipMemset(src, 0, maxb);
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)fill_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)fill_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(256);
    for (int threads : {256, 512, 768}) {
        for (int kb : {16, 64, 151}) {
            const int n16 = kb * 1024 / 16;
            for (int which = 0; which < 2; ++which) {
                double best = 1e30, mean = 0;
                for (int rep = 0; rep < 5; ++rep) {
                    if (which == 0) hipLaunchKernelGGL(fill_regs, dim3(256), dim3(threads), kb * 1024, 0, src, n16, out, sink);
                    else hipLaunchKernelGGL(fill_dma, dim3(256), dim3(threads), kb * 1024, 0, src, n16, out, sink);
                    hipDeviceSynchronize();
                    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
                    double m = 0;
                    for (auto v : h) m += (double)v;
                    m /= 256;
                    if (m < best) best = m;
                    mean = m;
                }
                printf("%-26s %3d KB, %3d threads/WG, 256 WGs: %8.0f cycles per workgroup (best of 5 means) = %6.1f B/cycle/CU\n",
                       which ? "global_load_lds_dwordx4" : "global_load -> ds_write", kb, threads, best, kb * 1024.0 / best);
            }
        }
    }
    return 0;
}
