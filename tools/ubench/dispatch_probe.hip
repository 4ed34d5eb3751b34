// Probe: where and when do the workgroups of a 256-WG persistent launch start?  (one WG per CU expected)
// build: hipcc --offload-arch=gfx950 -O3 dispatch_probe.hip -o dispatch_probe
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>
#include <hip/hip_runtime.h>

__global__ void probe(long long* rec, int spin, float* out) {
    extern __shared__ float smem[];
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    smem[threadIdx.x] = a;
    __syncthreads();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4 + 0] = r0;
        rec[blockIdx.x * 4 + 1] = r1;
        rec[blockIdx.x * 4 + 2] = hwid;
        rec[blockIdx.x * 4 + 3] = xcc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = smem[(threadIdx.x + 1) % blockDim.x];
}

void run(int grid, int threads, size_t lds, int spin) {
    long long* rec; float* out;
    hipMalloc(&rec, grid * 4 * sizeof(long long));
    hipMalloc(&out, grid * threads * sizeof(float));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, rec, spin, out);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 4);
    hipMemcpy(h.data(), rec, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = h[0], t1 = 0;
    for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 1]); }
    std::map<long long, int> per_cu;
    int late = 0;
    double dur = 0;
    for (int i = 0; i < grid; ++i) {
        const unsigned hw = (unsigned)h[4 * i + 2];
        const long long cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[4 * i + 3] & 0xf;
        per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
        dur += (h[4 * i + 1] - h[4 * i]) / 100.0;
        if (h[4 * i] - t0 > (h[1] - h[0]) / 2) ++late;
    }
    std::vector<long long> starts, ends;
    for (int i = 0; i < grid; ++i) { starts.push_back(h[4 * i] - t0); ends.push_back(h[4 * i + 1] - t0); }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    printf("   start skew us: p50 %.2f p90 %.2f max %.2f | end: min %.2f p50 %.2f max %.2f\n", starts[grid / 2] / 100.0, starts[grid * 9 / 10] / 100.0,
           starts[grid - 1] / 100.0, ends[0] / 100.0, ends[grid / 2] / 100.0, ends[grid - 1] / 100.0);
    int mx = 0;
    for (auto& kv : per_cu) mx = std::max(mx, kv.second);
    printf("grid %d x %d thr, LDS %zu KB: distinct CUs %zu, max WGs on one CU %d, WGs starting late %d, mean WG %.1f us, span %.1f us\n",
           grid, threads, lds / 1024, per_cu.size(), mx, late, dur / grid, (t1 - t0) / 100.0);
    if (grid <= 256 && lds > 80 * 1024) {
        std::map<int, int> per_xcc;
        for (int i = 0; i < grid; ++i) per_xcc[(int)(h[4 * i + 3] & 0xf)]++;
        printf("   WGs per XCC:");
        for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
        printf("   first 16 blockIdx -> xcc:");
        for (int i = 0; i < 16; ++i) printf(" %lld", h[4 * i + 3] & 0xf);
        printf("\n");
    }
    hipFree(rec); hipFree(out);
}

int main() {
    // which workgroups share a CU?  1024 WGs x 256 threads, 36 KB LDS (4 per CU)
    const int grid = 1024, threads = 256; const size_t lds = 36 * 1024;
    long long* rec; float* out;
    hipMalloc(&rec, grid * 4 * sizeof(long long));
    hipMalloc(&out, grid * threads * sizeof(float));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, rec, 20000, out);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 4);
    hipMemcpy(h.data(), rec, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::map<long long, std::vector<int>> per_cu;
    for (int i = 0; i < grid; ++i) {
        const unsigned hw = (unsigned)h[4 * i + 2];
        const long long cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[4 * i + 3] & 0xf;
        per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu].push_back(i);
    }
    int shown = 0;
    for (auto& kv : per_cu) { if (shown++ >= 12) break; printf("cu key %lld:", kv.first); for (int w : kv.second) printf(" %d", w); printf("\n"); }
    // how many of the WGs < 576 per CU
    int mx = 0, mn = 99; for (auto& kv : per_cu) { int c = 0; for (int w : kv.second) c += w < 576; mx = std::max(mx, c); mn = std::min(mn, c); }
    printf("WGs with id < 576 per CU: min %d max %d (CUs %zu)\n", mn, mx, per_cu.size());
    return 0;
}
