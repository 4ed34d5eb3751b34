// Micro-benchmark behind the key-loop schedule of csrc/prd_tri2.hip: cycles per iteration of
//   0: 7 x v_mfma_f32_32x32x16_f16 (chains of the step: 2 + 2 on two accumulators, 3 on a third)
//   1: 16 x v_exp_f32        2: the fp16 hi|lo split of 16 values (8 v_cvt_pk_f16_f32 + 16 v_fma_mix)
//   3: 16 v_add_f32          4: 4 x ds_read_b128
//   5: the whole step of pipe_step (MFMAs + softmax arithmetic + LDS reads) in issue order
// for 1 and 2 waves per SIMD (256- and 512-thread workgroups, one per CU).
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/ubench/tile_step_bench tools/ubench/tile_step_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__
#define FENCE() __builtin_amdgcn_sched_barrier(0)
DEV f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
DEV void split2h_rn(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
DEV void split8_rn(const f32x16& v, int base, u32x4& h, u32x4& l) {
#pragma unroll
    for (int w = 0; w < 4; ++w) { unsigned a, b; split2h_rn(v[base + 2 * w], v[base + 2 * w + 1], a, b); h[w] = a; l[w] = b; }
}

template <int MODE, int OFFS = 0>
__global__ __launch_bounds__(1024) void bench(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 sc, sn, o0, o1, negm;
    for (int e = 0; e < 16; ++e) { sc[e] = -0.01f * (e + lane); sn[e] = 0.f; o0[e] = 0.f; o1[e] = 0.f; negm[e] = -1.0f; }
    u32x4 qh = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, ql = {0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u};
    u32x4 kh = qh, kl = ql, va0 = qh, va1 = ql, ph0 = qh, pl0 = ql, ph1 = qh, pl1 = ql;
    float t0 = 0.f, t1 = 0.f, lsum = 0.f;
    const unsigned laddr = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 512u;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            o0 = mfma_h(va0, ph0, o0); o1 = mfma_h(va1, ph1, o1); o0 = mfma_h(va0, pl0, o0); o1 = mfma_h(va1, pl1, o1);
            sn = mfma_h(kh, qh, negm); sn = mfma_h(kh, ql, sn); sn = mfma_h(kl, qh, sn);
            sc[0] += sn[0] * 1e-30f;
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]) - 1.5f;
        } else if (MODE == 2) {
            split8_rn(sc, 0, ph0, pl0);
            split8_rn(sc, 8, ph1, pl1);
            sc[0] += __uint_as_float(ph0[0] ^ pl0[1] ^ ph1[2] ^ pl1[3]) * 1e-30f;
#pragma unroll
            for (int j = 1; j < 16; ++j) sc[j] += 1e-3f;
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) { t0 += sc[j]; t1 += sc[j + 1]; }
        } else if (MODE == 4) {
            const unsigned a = laddr + ((unsigned)it & 3u) * 4096u;
            va0 = *reinterpret_cast<const u32x4*>(lds + a);
            va1 = *reinterpret_cast<const u32x4*>(lds + a + 1024);
            kh = *reinterpret_cast<const u32x4*>(lds + a + 2048);
            kl = *reinterpret_cast<const u32x4*>(lds + a + 3072);
            t0 += __uint_as_float(va0[0] ^ va1[1] ^ kh[2] ^ kl[3]);
        } else if (MODE == 6) {
            // the plain order of one tile: Q K^T, softmax arithmetic, P V (no software pipelining; other waves fill the gaps)
            const unsigned a = laddr + ((unsigned)it & 3u) * 4096u;
            kh = *reinterpret_cast<const u32x4*>(lds + a + 2048);
            kl = *reinterpret_cast<const u32x4*>(lds + a + 3072);
            va0 = *reinterpret_cast<const u32x4*>(lds + a);
            va1 = *reinterpret_cast<const u32x4*>(lds + a + 1024);
            sc = mfma_h(kh, qh, negm); sc = mfma_h(kh, ql, sc); sc = mfma_h(kl, qh, sc);
            t0 = 0.f; t1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                sc[j] = __builtin_amdgcn_exp2f(sc[j] * 1e-3f); sc[j + 1] = __builtin_amdgcn_exp2f(sc[j + 1] * 1e-3f);
                t0 += sc[j]; t1 += sc[j + 1];
            }
            lsum += t0 + t1;
            split8_rn(sc, 0, ph0, pl0);
            split8_rn(sc, 8, ph1, pl1);
            o0 = mfma_h(va0, ph0, o0); o0 = mfma_h(va1, ph1, o0); o0 = mfma_h(va0, pl0, o0); o0 = mfma_h(va1, pl1, o0);
        } else {
            const unsigned a = laddr + ((unsigned)it & 3u) * 4096u;
            o0 = mfma_h(va0, ph0, o0);
            FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
            t0 = sc[0] + sc[2]; t1 = sc[1] + sc[3];
            FENCE();
            o1 = mfma_h(va1, ph1, o1);
            FENCE();
#pragma unroll
            for (int j = 4; j < 8; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
            t0 += sc[4]; t1 += sc[5]; t0 += sc[6]; t1 += sc[7];
            FENCE();
            o0 = mfma_h(va0, pl0, o0);
            FENCE();
#pragma unroll
            for (int j = 8; j < 12; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
            t0 += sc[8]; t1 += sc[9]; t0 += sc[10]; t1 += sc[11];
            FENCE();
            o1 = mfma_h(va1, pl1, o1);
            FENCE();
            va0 = *reinterpret_cast<const u32x4*>(lds + a);
            va1 = *reinterpret_cast<const u32x4*>(lds + a + 1024);
#pragma unroll
            for (int j = 12; j < 16; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
            t0 += sc[12]; t1 += sc[13]; t0 += sc[14]; t1 += sc[15];
            FENCE();
            sn = mfma_h(kh, qh, negm);
            FENCE();
            split8_rn(sc, 0, ph0, pl0);
            FENCE();
            sn = mfma_h(kh, ql, sn);
            FENCE();
            split8_rn(sc, 8, ph1, pl1);
            FENCE();
            sn = mfma_h(kl, qh, sn);
            FENCE();
            kh = *reinterpret_cast<const u32x4*>(lds + a + 2048);
            kl = *reinterpret_cast<const u32x4*>(lds + a + 3072);
            lsum += t0 + t1;
            FENCE();
            // rotate: the logits of the next tile become the current ones (register renaming is free in the real kernel)
#pragma unroll
            for (int j = 0; j < 16; ++j) sc[j] = sn[j] * 1e-3f - 0.5f - (float)OFFS;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = t0 + t1 + lsum;
    for (int e = 0; e < 16; ++e) acc += sc[e] + sn[e] + o0[e] + o1[e];
    acc += __uint_as_float(ph0[0] ^ pl0[0] ^ ph1[0] ^ pl1[0] ^ va0[0] ^ va1[0] ^ kh[0] ^ kl[0]);
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int MODE, int OFFS = 0>
static void run(const char* name, int threads) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 4096);
    const int iters = 2000;
    hipLaunchKernelGGL((bench<MODE, OFFS>), dim3(256), dim3(threads), 32768, 0, d, sink, iters);
    hipLaunchKernelGGL((bench<MODE, OFFS>), dim3(256), dim3(threads), 32768, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { double v = (double)h[b * 16 + w] / iters; s += v; if (v > mx) mx = v; ++n; }
    printf("%-28s %4d threads/WG: %8.1f cycles per iteration per wave (max %.1f) -> %7.1f per iteration per SIMD\n", name, threads, s / n, mx, s / n / (threads / 256));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("7 MFMA 32x32x16 f16", threads);
        run<1>("16 v_exp_f32 (+16 sub)", threads);
        run<2>("split 16 values (+16 add)", threads);
        run<3>("16 v_add_f32", threads);
        run<4>("4 ds_read_b128", threads);
        run<5>("whole step (pipelined)", threads);
        run<6>("whole step (plain order)", threads);
        run<5, 10>("pipelined, p ~ 2^-10", threads);
        run<5, 18>("pipelined, p ~ 2^-18", threads);
        run<5, 30>("pipelined, p ~ 2^-30", threads);
    }
    return 0;
}
