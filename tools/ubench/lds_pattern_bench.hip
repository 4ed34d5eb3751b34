// LDS access patterns of the attention cores' V layout, one instruction kind at a time: cycles per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ubench/lds_pattern_bench.hip -o tools/ubench/lds_pattern_bench
// One workgroup of 1 / 4 / 12 waves per CU (256 workgroups), every wave issues the same instruction REPS times back to back with
// per-lane addresses of the pattern under test; s_memtime around the loop, the slowest wave of workgroup 0 is printed.
//   w_b128      the round-3 V store: lane (r, hi) -> 16 bytes at hi * 512 + r * 16                    (ds_write_b128)
//   w_t_plain   transposed 4-byte store, slot = channel                                              (ds_write_b32)
//   w_t_xor     ... slot = channel ^ (2 a + khalf)
//   w_t_gv      ... slot = gv_slot(channel) ^ (2 a + khalf)   (bits 2, 3 of the channel swapped: the shipped form)
//   r_plain     the P V operand read: lane (r, hi) -> 16 bytes at hi * 512 + r * 16                   (ds_read_b128)
//   r_gv        ... at hi * 512 + (gv_slot(r) ^ hi) * 16
//   r_w_full    weight operand read of a full row group: row 32 + r, slot (2 s + hi) ^ ((row >> 1) & 7) (h2_slot), 128-byte rows
//   r_w_dup     ... rows 32 + (r & 15) (the round-3 G / V groups: lanes r and r + 16 read the same address)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

__device__ __forceinline__ int gv_slot(int ch) { return (ch & ~12) | ((ch & 4) << 1) | ((ch & 8) >> 1); }

template <int KIND>
__global__ void pat(unsigned long long* out, int reps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hi = lane >> 5;
    unsigned addr = 0;
    {   // transposed store coordinates (see csrc/prd_tri2.hip, tri_attn_core_v3_kernel, GV)
        const bool odd = r & 1;
        const int kp0 = r & 30, a_ = kp0 >> 4, kk = kp0 & 15, kh_ = (kk >> 2) & 1, w_ = (kk >> 3) * 2 + ((kk & 3) >> 1);
        const int ch = 4 * hi + (odd ? 8 : 0);          // + j (j = 0 here)
        const unsigned vo = a_ * 1024u + kh_ * 512u + w_ * 4u;
        const int sx = 2 * a_ + kh_;
        if (KIND == 0) addr = hi * 512u + r * 16u;
        if (KIND == 1) addr = vo + ch * 16u;
        if (KIND == 2) addr = vo + (ch ^ sx) * 16u;
        if (KIND == 3) addr = vo + (gv_slot(ch) ^ sx) * 16u;
        if (KIND == 4) addr = hi * 512u + r * 16u;
        if (KIND == 5) addr = hi * 512u + (gv_slot(r) ^ hi) * 16u;
        if (KIND == 6) { const int row = 32 + r; addr = 4096u + row * 128u + (((2 * 1 + hi) ^ ((row >> 1) & 7)) << 4); }
        if (KIND == 7) { const int row = 32 + (r & 15); addr = 4096u + row * 128u + (((2 * 1 + hi) ^ ((row >> 1) & 7)) << 4); }
    }
    addr += wave * 13312u;
    unsigned d0 = lane, d1 = lane + 1, d2 = lane + 2, d3 = lane + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, uint4{d0, d1, d2, d3})) : "memory");
            else if (KIND <= 3) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(d0) : "memory");
            else {
                __attribute__((ext_vector_type(4))) unsigned v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("" ::"v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (d0 == 0xffffffffu) sink[0] = d1;
}

int main() {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 64);
    const char* names[8] = {"w_b128", "w_t_plain", "w_t_xor", "w_t_gv", "r_plain", "r_gv", "r_w_full", "r_w_dup"};
    void (*k[8])(unsigned long long*, int, unsigned*) = {pat<0>, pat<1>, pat<2>, pat<3>, pat<4>, pat<5>, pat<6>, pat<7>};
    const int reps = 2000;
    for (int kind = 0; kind < 8; ++kind) {
        std::string line;
        for (int nw : {1, 4, 12}) {
            hipFuncSetAttribute((const void*)k[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k[kind], dim3(256), dim3(64 * nw), 160 * 1024, 0, out, reps, sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(16);
            hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int w = 0; w < nw; ++w) mx = h[w] > mx ? h[w] : mx;
            char buf[96];
            snprintf(buf, sizeof buf, "  %2d waves: %7.2f cycles / instr / wave-slot", nw, (double)mx / (reps * 16.0));
            line += buf;
        }
        printf("%-10s %s\n", names[kind], line.c_str());
    }
    return 0;
}
