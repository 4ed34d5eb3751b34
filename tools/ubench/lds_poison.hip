// Debug aid: fill the whole LDS of every CU (and, optionally, a block of device memory) with a NaN pattern, so that a kernel that
// reads LDS it never wrote shows up deterministically (LDS keeps what the previous kernel left; on a freshly booted box that is
// whatever ran before).  extern "C" int prd_dbg_poison_lds(unsigned pattern, void* stream)
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/ubench/liblds_poison.so tools/ubench/lds_poison.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void poison_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 97) % (160 * 1024 / 4)] == 12345u) sink[0] = 1;      // keep the stores
}
extern "C" int prd_dbg_poison_lds(unsigned pattern, void* stream) {
    static unsigned* sink = nullptr;
    if (!sink) { if (hipMalloc(&sink, 64) != hipSuccess) return -1; }
    (void)hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // 160 KB per workgroup: one workgroup per CU at a time; 4 rounds over the 256 CUs so that every CU is hit whatever the placement
    hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, sink);
    return (int)hipGetLastError();
}
