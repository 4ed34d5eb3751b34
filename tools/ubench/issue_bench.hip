// Round-5 micro-benchmarks behind the key loop of csrc/prd_tri2.hip (gfx950).  Three questions:
//   A  data dependence of the conversion-class instructions: v_cvt_pk_f16_f32 whose results are normal / denormal / zero fp16
//      numbers, v_fma_mix_f32 reading a denormal fp16 source (the lo part of a small probability is a denormal fp16 number)
//   B  do a matrix wave and a vector wave ON THE SAME SIMD overlap?  (waves w and w + 4 of a workgroup share a SIMD: roles by
//      (wave >> 2), so every SIMD carries one wave of each role -- tools/ubench/overlap_bench.hip split the roles by wave parity,
//      which puts two waves of the SAME role on a SIMD)
//   C  one key-loop tile step exactly as tri_attn_core_v3_kernel issues it (3 Q K^T MFMAs -> 16 v_exp_f32 + row sum -> fp16 hi|lo
//      split -> 4 P V MFMAs; operands from LDS) at 1 .. 4 waves per SIMD, and variants: P V accumulator in AGPRs, two P V
//      accumulators, zero V_lo rows under the P_lo products, no row sum, the next tile's Q K^T issued before the split
//      (software pipelining), probabilities of different magnitude.
// Every line prints ticks of s_memtime per unit AND the wall time (hipEvent), i.e. the tick rate can be read off.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -o tools/ubench/issue_bench tools/ubench/issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

DEV f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// accumulator in AGPRs (the compiler keeps it in VGPRs when it can)
DEV void mfma_acc(u32x4 a, u32x4 b, f32x16& c) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
#define ACLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
DEV void mfma_hard(u32x4 a, u32x4 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : ACLOB);
}
DEV void split2h_rn(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, h16x2));
}
// lo part scaled by 2^11 (a normal fp16 number whenever hi is): r = (x - hi) * 2048 with the scale inside the two fma_mix
// (src1 = -2048 from an SGPR, src2 = x * 2048 needs one more multiply per value: the variant measures whether avoiding fp16
// denormals pays for it)
DEV void split2h_scaled(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    float ra, rb;
    const float a2 = a * 2048.0f, b2 = b * 2048.0f;
    asm("v_fma_mix_f32 %0, %1, %3, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a2), "s"(-2048.0f));
    asm("v_fma_mix_f32 %0, %1, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b2), "s"(-2048.0f));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, h16x2));
}
DEV void split2h_mixlo(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
template <int SPLIT>
DEV void split8(const f32x16& v, int base, u32x4& h, u32x4& l) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned a, b;
        if (SPLIT == 0) split2h_rn(v[base + 2 * w], v[base + 2 * w + 1], a, b);
        else if (SPLIT == 2) split2h_mixlo(v[base + 2 * w], v[base + 2 * w + 1], a, b);
        else split2h_scaled(v[base + 2 * w], v[base + 2 * w + 1], a, b);
        h[w] = a; l[w] = b;
    }
}

struct Out { unsigned long long ticks[16]; };

// ------------------------------------------------------------------------------------------------------------------ A
// 16 independent instructions per iteration; DATA selects the magnitude of the operands
template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(Out* out, float* sink, int iters, float mag) {
    const int lane = threadIdx.x & 63;
    float a[16], b[16];
    unsigned u[16];
    for (int e = 0; e < 16; ++e) {
        a[e] = mag * (1.0f + 0.01f * (e + lane));
        b[e] = mag * (1.5f + 0.01f * e);
        const h16x2 hh = {(_Float16)a[e], (_Float16)b[e]};
        u[e] = __builtin_bit_cast(unsigned, hh);
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define A_CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define A_MIX(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(b[i]) : "v"(u[i]), "v"(a[i]));
#define A_EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define A_ADD(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(b[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]));
#define A_DOT2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(b[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double*>(&b[(i) & 14])) : "v"(*reinterpret_cast<double*>(&a[(i) & 14])), "v"(*reinterpret_cast<double*>(&a[((i) + 2) & 14])));
        if (OP == 0) { REP16(A_CVT) }
        else if (OP == 1) { REP16(A_MIX) }
        else if (OP == 2) { REP16(A_EXP) }
        else if (OP == 3) { REP16(A_ADD) }
        else if (OP == 4) { REP16(A_DOT2) }
        else if (OP == 5) { REP16(A_PKMUL) }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += a[e] + b[e] + __uint_as_float(u[e]);
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x].ticks[threadIdx.x >> 6] = c1 - c0;
}

// ------------------------------------------------------------------------------------------------------------------ B
// waves with (wave >> 2) % 2 == 0 issue NM MFMAs per iteration (two accumulators), the others NV VALU instructions (KIND 0
// v_add_f32, 1 v_exp_f32, 2 v_cvt_pk_f16_f32); ROLES 0: every wave does both (MFMA first, then the VALU block), 1: split
template <int KIND, int NM, int NV, int ROLES, int AGPR>
__global__ __launch_bounds__(1024) void overlap_kernel(Out* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 o0, o1;
    float a[16];
    unsigned u[16];
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; a[e] = -0.01f * (e + lane) - 0.3f; u[e] = 0x3c003800u + e + lane; }
    u32x4 x = {0x3c003c00u + lane, 0x3a003c00u, 0x3c003800u, 0x39003c00u}, y = {0x30003100u, 0x32003000u + lane, 0x31003000u, 0x30003300u};
    const bool do_m = ROLES == 0 || ((wave >> 2) & 1) == 0, do_v = ROLES == 0 || ((wave >> 2) & 1) == 1;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int g = 0; g < NM; ++g) {
                if (AGPR) { if (g & 1) mfma_acc(x, y, o1); else mfma_acc(x, y, o0); }
                else { if (g & 1) o1 = mfma_h(x, y, o1); else o0 = mfma_h(x, y, o0); }
            }
        }
        FENCE();
        if (do_v) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = v & 15;
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 5) & 15]));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
                else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 5) & 15]));
            }
        }
        FENCE();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += a[e] + o0[e] + o1[e] + __uint_as_float(u[e]);
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x].ticks[wave] = c1 - c0;
}

// ------------------------------------------------------------------------------------------------------------------ C
// FLAGS: 1 P V accumulator in AGPRs | 2 two P V accumulators | 4 zero V_lo rows under the P_lo products | 8 no row sum |
//        16 Q K^T of the next tile issued before the split of this one | 32 scaled lo parts (no fp16 denormals)
//        64 no split at all (hi parts only: 2 P V MFMAs) -- the VALU floor without the lo parts
//        128 row sum on the matrix pipe: v_mfma_f32_4x4x4_16B_f16 with A = ones adds a lane's own four fp16 values to its accumulator
//            (8 per tile: hi and lo parts) | 256 the same on two accumulators | 512 row sum by v_dot2_f32_f16 (16 per tile)
//        1024 (not run: hipcc re-uses hard-coded AGPRs between asm statements) | 2048 lean zero V_lo rows under
//            P_lo: a second pair of V registers from a per-lane address that stays on a zero line for the lanes of the lo plane
//        4096 split with v_fma_mixlo/hi_f16 (3 instructions per pair, two of them at the transcendental rate)
template <int FLAGS>
__global__ __launch_bounds__(1024) void tile_kernel(Out* out, float* sink, int iters, float offs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    // K planes (hi | lo), V tiles: pseudo-random fp16 values in [0.06, 1) (hi parts) / 2^-12 of that (lo parts)
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        const unsigned hsh = (unsigned)i * 2654435761u;
        const unsigned short hv = (unsigned short)(0x2c00u + (hsh >> 20 & 0x0fffu)), lv = (unsigned short)(0x0400u + (hsh >> 8 & 0x03ffu));
        reinterpret_cast<unsigned short*>(lds)[i] = (i & 1024) ? lv : hv;
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<unsigned*>(lds + 32768)[i] = 0u;
    __syncthreads();
    u32x4 qh, ql;
    for (int w = 0; w < 4; ++w) { qh[w] = 0x30003400u + ((unsigned)lane * 2654435761u >> 22 & 0x03ff03ffu); ql[w] = 0x04000500u + (unsigned)lane; }
    f32x16 o0, o1, negm;
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; negm[e] = -offs; }
    float lsum = 0.f, d0 = 0.f, d1 = 0.f;
    f32x4 la0 = {0.f, 0.f, 0.f, 0.f}, la1 = {0.f, 0.f, 0.f, 0.f};
    bool big = false;
    const unsigned kbase = (unsigned)hi * 1024u + (unsigned)r * 16u;          // + 512 t, lo plane + 2048
    const unsigned vbase = 8192u + (unsigned)hi * 512u + (unsigned)r * 16u;     // + 2048 t
    const unsigned zbase = (FLAGS & 4) ? (r >= 16 ? 32768u + (unsigned)(r & 15) * 16u : 0xffffffffu) : 0u;
    __syncthreads();
    u32x4 kh = *reinterpret_cast<const u32x4*>(lds + kbase), kl = *reinterpret_cast<const u32x4*>(lds + kbase + 2048u);
    f32x16 snext;
    if (FLAGS & 16) { snext = mfma_h(kh, qh, negm); snext = mfma_h(kh, ql, snext); snext = mfma_h(kl, qh, snext); }
    if (FLAGS & 1024) {
        asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\t"
                     "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"
                     "v_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\t"
                     "v_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0" ::: ACLOB);
    }
    // lean zero rows: lanes of the lo plane (r >= 16) read a zero line at a fixed address, the others the tile's V registers again
    unsigned zaddr = (r >= 16) ? 32768u + (unsigned)(r & 15) * 16u : vbase;
    const unsigned zstep = (r >= 16) ? 0u : 2048u, zhalf = (r >= 16) ? 0u : 1024u;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned t = (unsigned)it & 3u;
        f32x16 s;
        if (FLAGS & 16) s = snext;
        else { s = mfma_h(kh, qh, negm); s = mfma_h(kh, ql, s); s = mfma_h(kl, qh, s); }
        const unsigned tn = (unsigned)(it + 1) & 3u;
        kh = *reinterpret_cast<const u32x4*>(lds + kbase + 512u * tn);
        kl = *reinterpret_cast<const u32x4*>(lds + kbase + 512u * tn + 2048u);
        u32x4 va0 = *reinterpret_cast<const u32x4*>(lds + vbase + 2048u * t), va1 = *reinterpret_cast<const u32x4*>(lds + vbase + 2048u * t + 1024u);
        u32x4 vz0 = va0, vz1 = va1;
        if (FLAGS & 2048) {
            const unsigned za = zaddr + (t == 0 ? 0u : zstep * t);
            vz0 = *reinterpret_cast<const u32x4*>(lds + za);
            vz1 = *reinterpret_cast<const u32x4*>(lds + za + zhalf);
        }
        if (FLAGS & 4) {        // the A operand of the P_lo products: V_hi rows as they are, V_lo rows (lanes 16-31 of a half) zero
            const unsigned za = zbase == 0xffffffffu ? vbase + 2048u * t : zbase;
            vz0 = *reinterpret_cast<const u32x4*>(lds + za);
            vz1 = *reinterpret_cast<const u32x4*>(lds + za + (zbase == 0xffffffffu ? 1024u : 0u));
        }
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            s[j] = __builtin_amdgcn_exp2f(s[j]);
            s[j + 1] = __builtin_amdgcn_exp2f(s[j + 1]);
            if (!(FLAGS & (8 | 128 | 512))) { t0 += s[j]; t1 += s[j + 1]; }
        }
        if (!(FLAGS & (8 | 128 | 512))) {
            const float ts = t0 + t1;
            big |= !(ts < 30000.0f);
            lsum += ts;
        }
        if (FLAGS & 16) { snext = mfma_h(kh, qh, negm); snext = mfma_h(kh, ql, snext); snext = mfma_h(kl, qh, snext); }
        u32x4 ph0, pl0, ph1, pl1;
        if (FLAGS & 64) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                ph0[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s[2 * w], s[2 * w + 1]}, h16x2));
                ph1[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s[8 + 2 * w], s[9 + 2 * w]}, h16x2));
            }
            o0 = mfma_h(va0, ph0, o0);
            o0 = mfma_h(va1, ph1, o0);
        } else {
            split8<(FLAGS & 32) ? 1 : (FLAGS & 4096) ? 2 : 0>(s, 0, ph0, pl0);
            split8<(FLAGS & 32) ? 1 : (FLAGS & 4096) ? 2 : 0>(s, 8, ph1, pl1);
            if (FLAGS & 128) {
                const h16x4 ones = {(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
#define SUM4(acc, v, w) acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(h16x4, u32x2{v[w], v[w + 1]}), acc, 0, 0, 0)
                if (FLAGS & 256) {
                    SUM4(la0, ph0, 0); SUM4(la1, ph0, 2); SUM4(la0, ph1, 0); SUM4(la1, ph1, 2);
                    SUM4(la0, pl0, 0); SUM4(la1, pl0, 2); SUM4(la0, pl1, 0); SUM4(la1, pl1, 2);
                } else {
                    SUM4(la0, ph0, 0); SUM4(la0, ph0, 2); SUM4(la0, ph1, 0); SUM4(la0, ph1, 2);
                    SUM4(la0, pl0, 0); SUM4(la0, pl0, 2); SUM4(la0, pl1, 0); SUM4(la0, pl1, 2);
                }
            }
            if (FLAGS & 512) {
                const h16x2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    d0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, ph0[w]), one2, d0, false);
                    d1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, ph1[w]), one2, d1, false);
                    d0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, pl0[w]), one2, d0, false);
                    d1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, pl1[w]), one2, d1, false);
                }
            }
            if (FLAGS & 1024) {
                mfma_hard(va0, ph0); mfma_hard(va1, ph1); mfma_hard(vz0, pl0); mfma_hard(vz1, pl1);
            } else if (FLAGS & 1) {
                if (FLAGS & 2) { mfma_acc(va0, ph0, o0); mfma_acc(va1, ph1, o1); mfma_acc(vz0, pl0, o0); mfma_acc(vz1, pl1, o1); }
                else { mfma_acc(va0, ph0, o0); mfma_acc(va1, ph1, o0); mfma_acc(vz0, pl0, o0); mfma_acc(vz1, pl1, o0); }
            } else if (FLAGS & 2) {
                o0 = mfma_h(va0, ph0, o0); o1 = mfma_h(va1, ph1, o1); o0 = mfma_h(vz0, pl0, o0); o1 = mfma_h(vz1, pl1, o1);
            } else {
                o0 = mfma_h(va0, ph0, o0); o0 = mfma_h(va1, ph1, o0); o0 = mfma_h(vz0, pl0, o0); o0 = mfma_h(vz1, pl1, o0);
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float acc = lsum + (big ? 1.f : 0.f) + d0 + d1 + la0[0] + la1[0];
    if (FLAGS & 1024) { float x; asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(x) :: ACLOB); acc += x; }
    for (int e = 0; e < 16; ++e) acc += o0[e] + o1[e] + ((FLAGS & 16) ? snext[e] : 0.f);
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
    if (lane == 0) out[blockIdx.x].ticks[threadIdx.x >> 6] = c1 - c0;
}

// ------------------------------------------------------------------------------------------------------------------ host
struct Res { double mean_wave, max_wave_mean, ms; };
template <typename L>
static Res launch(L&& fn, int threads, int iters) {
    Out* d; float* sink;
    hipMalloc(&d, 256 * sizeof(Out)); hipMalloc(&sink, 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn(d, sink);                         // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    fn(d, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Out> h(256);
    hipMemcpy(h.data(), d, 256 * sizeof(Out), hipMemcpyDeviceToHost);
    double s = 0, smx = 0; int n = 0;
    for (int b = 0; b < 256; ++b) {
        double mx = 0;
        for (int w = 0; w < threads / 64; ++w) { const double v = (double)h[b].ticks[w] / iters; s += v; ++n; if (v > mx) mx = v; }
        smx += mx;
    }
    hipFree(d); hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
    return {s / n, smx / 256, (double)ms};
}

template <int OP>
static void run_rate(const char* name, float mag) {
    const int iters = 4000;
    printf("A  %-34s mag %8.1e :", name, mag);
    for (int threads : {256, 512, 1024}) {
        Res r = launch([&](Out* d, float* s) { hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(threads), 0, 0, d, s, iters, mag); }, threads, iters);
        printf("  %6.2f", r.max_wave_mean / 16 / (threads / 256));
        if (threads == 1024) printf("   ticks per instruction per SIMD at 1 / 2 / 4 waves per SIMD;  %.3f GHz tick rate", r.max_wave_mean * iters / (r.ms * 1e6));
    }
    printf("\n");
}

template <int KIND, int NM, int NV, int ROLES, int AGPR>
static void run_overlap(const char* name) {
    const int iters = 4000;
    printf("B  %-58s:", name);
    for (int threads : {512, 1024}) {
        Out* dd = nullptr;
        Res r = launch([&](Out* d, float* s) { dd = d; hipLaunchKernelGGL((overlap_kernel<KIND, NM, NV, ROLES, AGPR>), dim3(256), dim3(threads), 0, 0, d, s, iters); }, threads, iters);
        printf("  %7.1f (%.3f ms)", r.max_wave_mean, r.ms);
    }
    printf("   ticks per iteration (slowest wave) at 2 / 4 waves per SIMD\n");
}

template <int FLAGS>
static void run_tile(const char* name, float offs) {
    const int iters = 3000;
    printf("C  %-44s p ~ 2^-%-4.0f:", name, offs);
    for (int threads : {256, 512, 768, 1024}) {
        Res r = launch([&](Out* d, float* s) { hipLaunchKernelGGL(tile_kernel<FLAGS>, dim3(256), dim3(threads), 34816, 0, d, s, iters, offs); }, threads, iters);
        printf("  %6.1f|%6.1f", r.max_wave_mean / (threads / 256), r.ms * 1e6 / iters / (threads / 256));
    }
    printf("   ticks|ns per tile step per SIMD at 1 / 2 / 3 / 4 waves per SIMD\n");
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "ABC";
    auto has = [&](char c) { for (const char* p = what; *p; ++p) if (*p == c) return true; return false; };
    if (has('A')) {
        for (float mag : {1.0f, 1e-3f, 3e-6f, 1e-9f}) run_rate<0>("v_cvt_pk_f16_f32 (result magnitude)", mag);
        for (float mag : {1.0f, 1e-3f, 3e-6f, 1e-9f}) run_rate<1>("v_fma_mix_f32 (fp16 source magnitude)", mag);
        for (float mag : {-1.0f, -20.0f, -140.0f}) run_rate<2>("v_exp_f32 (argument)", mag);
        run_rate<3>("v_add_f32", 1.0f);
        run_rate<4>("v_dot2_f32_f16", 1.0f);
        run_rate<5>("v_pk_mul_f32", 1.0f);
    }
    if (has('B')) {
        run_overlap<0, 7, 0, 0, 0>("7 MFMA, every wave");
        run_overlap<0, 7, 0, 0, 1>("7 MFMA (AGPR accumulators), every wave");
        run_overlap<0, 0, 56, 0, 0>("56 v_add, every wave");
        run_overlap<0, 7, 56, 0, 0>("7 MFMA then 56 v_add, every wave");
        run_overlap<0, 7, 56, 0, 1>("7 MFMA (AGPR) then 56 v_add, every wave");
        run_overlap<0, 7, 56, 1, 0>("7 MFMA | 56 v_add, one role per wave, both on a SIMD");
        run_overlap<0, 7, 56, 1, 1>("7 MFMA (AGPR) | 56 v_add, one role per wave");
        run_overlap<0, 7, 112, 1, 0>("7 MFMA | 112 v_add, one role per wave");
        run_overlap<0, 7, 112, 1, 1>("7 MFMA (AGPR) | 112 v_add, one role per wave");
        run_overlap<1, 0, 28, 0, 0>("28 v_exp, every wave");
        run_overlap<1, 7, 28, 1, 0>("7 MFMA | 28 v_exp, one role per wave");
        run_overlap<1, 7, 28, 1, 1>("7 MFMA (AGPR) | 28 v_exp, one role per wave");
        run_overlap<2, 0, 48, 0, 0>("48 v_cvt_pk, every wave");
        run_overlap<2, 7, 48, 1, 0>("7 MFMA | 48 v_cvt_pk, one role per wave");
        run_overlap<2, 7, 48, 1, 1>("7 MFMA (AGPR) | 48 v_cvt_pk, one role per wave");
    }
    if (has('C')) {
        for (float offs : {6.0f}) {
            run_tile<0>("tile step as tri_attn_core_v3 issues it", offs);
            run_tile<1>("  P V accumulator in AGPRs", offs);
            run_tile<2>("  two P V accumulators", offs);
            run_tile<3>("  two P V accumulators in AGPRs", offs);
            run_tile<4>("  zero V_lo rows under P_lo", offs);
            run_tile<8>("  no row sum", offs);
            run_tile<16>("  next Q K^T before the split", offs);
            run_tile<17>("  next Q K^T before the split + AGPR", offs);
            run_tile<32>("  lo parts x 2^11 (no fp16 denormals)", offs);
            run_tile<64>("  hi parts only (no split, 2 P V MFMAs)", offs);
            run_tile<128>("  row sum by 8 x mfma_4x4x4 (ones)", offs);
            run_tile<128 | 256>("  row sum by 8 x mfma_4x4x4, two accumulators", offs);
            run_tile<512>("  row sum by 16 x v_dot2_f32_f16", offs);
            run_tile<128 | 16>("  mfma_4x4x4 row sum + next Q K^T early", offs);
            run_tile<128 | 256 | 16>("  mfma_4x4x4 (2 acc) + next Q K^T early", offs);
            run_tile<512 | 16>("  v_dot2 row sum + next Q K^T early", offs);
            run_tile<8 | 16>("  no row sum + next Q K^T early", offs);
            run_tile<2048>("  lean zero V_lo rows under P_lo", offs);
            run_tile<2048 | 128 | 256>("  lean zero rows + mfma_4x4x4 row sum", offs);
            run_tile<2048 | 128 | 256 | 16>("  lean zero rows + mfma sum + early Q K^T", offs);
            run_tile<4096>("  split by v_fma_mixlo/hi_f16", offs);
            run_tile<4096 | 128 | 256>("  mixlo/hi split + mfma_4x4x4 row sum", offs);
        }
    }
    return 0;
}
