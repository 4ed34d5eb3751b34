// Issue cost of single VALU instruction types on gfx950 (cycles per wave64 instruction), 16 INDEPENDENT instructions per loop
// iteration, for 1, 2 and 4 waves per SIMD.  Behind the choice of the fp16 hi|lo split and softmax arithmetic of csrc/prd_tri2.hip.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/valu_rate_bench tools/ubench/valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(1024) void bench(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    float a[16], b[16];
    unsigned u[16];
    for (int e = 0; e < 16; ++e) { a[e] = -0.01f * (e + lane) - 0.3f; b[e] = 1.0f + 0.001f * e; u[e] = 0x3c003800u + e + lane; }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define OP_ADD(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define OP_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(u[i]) : "v"(u[i]), "v"(a[i]));
#define OP_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(a[i]));
#define OP_MIX32(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(b[i]) : "v"(u[i]), "v"(a[i]));
#define OP_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define OP_CVTRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define OP_CVT32(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(b[i]) : "v"(u[i]));
#define OP_AND(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double*>(&a[(i) & 14])) : "v"(*reinterpret_cast<double*>(&a[(i) & 14])), "v"(*reinterpret_cast<double*>(&b[(i) & 14])));
#define OP_PKADDH(i) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define OP_PKFMAH(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %1" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define OP_MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 15]));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define OP_DOT2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define OP_LDEXP(i) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(u[i]));
#define OP_EXP16(i) asm volatile("v_exp_f16 %0, %1" : "=v"(u[i]) : "v"(u[i]));
#define OP_SUBSDWA(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
        if (MODE == 0) { REP16(OP_ADD) }
        else if (MODE == 1) { REP16(OP_FMA) }
        else if (MODE == 2) { REP16(OP_EXP) }
        else if (MODE == 3) { REP16(OP_RCP) }
        else if (MODE == 4) { REP16(OP_MIXLO) }
        else if (MODE == 5) { REP16(OP_MIXHI) }
        else if (MODE == 6) { REP16(OP_MIX32) }
        else if (MODE == 7) { REP16(OP_CVTPK) }
        else if (MODE == 8) { REP16(OP_CVTRTZ) }
        else if (MODE == 9) { REP16(OP_CVT32) }
        else if (MODE == 10) { REP16(OP_AND) }
        else if (MODE == 11) { REP16(OP_PERM) }
        else if (MODE == 12) { REP16(OP_PKADD) }
        else if (MODE == 13) { REP16(OP_PKADDH) }
        else if (MODE == 14) { REP16(OP_PKFMAH) }
        else if (MODE == 15) { REP16(OP_MAX3) }
        else if (MODE == 16) { REP16(OP_MOV) }
        else if (MODE == 17) { REP16(OP_DOT2) }
        else if (MODE == 18) { REP16(OP_LDEXP) }
        else if (MODE == 19) { REP16(OP_EXP16) }
        else if (MODE == 20) { REP16(OP_EXP) REP16(OP_ADD) }       // 16 exp + 16 add interleaved by type
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += a[e] + b[e] + __uint_as_float(u[e]);
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

static const char* NAMES[] = {"v_add_f32", "v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_fma_mix_f32",
                              "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16", "v_and_b32", "v_perm_b32", "v_pk_add_f32",
                              "v_pk_add_f16", "v_pk_fma_f16", "v_max3_f32", "v_mov_b32", "v_dot2_f32_f16", "v_ldexp_f32", "v_exp_f16",
                              "16 v_exp_f32 + 16 v_add_f32"};

template <int MODE>
static void run() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 8192);
    const int iters = 2000;
    double res[3];
    int k = 0;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
        // SIMD-level cost: the waves of a SIMD are served oldest first, so the LAST wave to finish marks the time the SIMD
        // needed for the instructions of all its waves (the mean over waves under-counts)
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) { double mx = 0; for (int w = 0; w < threads / 64; ++w) { double v = (double)h[b * 16 + w] / iters; if (v > mx) mx = v; } s += mx; ++n; }
        res[k++] = s / n / (MODE == 20 ? 32 : 16);
    }
    printf("%-30s SIMD cycles per wave64 instruction: %6.2f (1 wave/SIMD) %6.2f (2) %6.2f (4)\n", NAMES[MODE], res[0], res[1] / 2, res[2] / 4);
    hipFree(d); hipFree(sink);
}

int main() {
    run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<9>(); run<10>(); run<11>(); run<12>();
    run<13>(); run<14>(); run<15>(); run<16>(); run<17>(); run<18>(); run<19>(); run<20>();
    return 0;
}
