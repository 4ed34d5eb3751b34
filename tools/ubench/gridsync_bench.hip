// Micro-benchmark: cost of a software grid barrier (agent-scope atomics + fences) across 256 co-resident workgroups.
// build: hipcc --offload-arch=gfx950 -O3 gridsync_bench.hip -o gridsync_bench
#include <cstdio>
#include <hip/hip_runtime.h>

template <int MODE>
__device__ __forceinline__ void grid_barrier(int* ctr, int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) __threadfence();
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (MODE == 0) __threadfence();
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void k(int* ctr, float* data, int nbar) {
    float v = threadIdx.x;
    for (int i = 0; i < nbar; ++i) {
        data[blockIdx.x * 512 + threadIdx.x] = v;                           // some global traffic between barriers
        grid_barrier<MODE>(ctr, (i + 1) * gridDim.x);
        v += data[((blockIdx.x + 1) % gridDim.x) * 512 + threadIdx.x];      // read a neighbour's data (must be visible)
    }
    data[blockIdx.x * 512 + threadIdx.x] = v;
}

int main() {
    int* ctr; float* data;
    hipMalloc(&ctr, 4); hipMalloc(&data, 256 * 512 * 4);
    for (int mode = 0; mode < 3; ++mode)
    for (int nbar : {0, 10, 100}) {
        hipMemset(ctr, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, ctr, data, nbar); else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, ctr, data, nbar); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, ctr, data, nbar);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (0 threadfence x2, 1 no fence, 2 release/acquire agent): %3d barriers: %.1f us total -> %.2f us per barrier\n", mode, nbar, ms * 1e3, nbar ? ms * 1e3 / nbar : 0.0);
    }
    return 0;
}
