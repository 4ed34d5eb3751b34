// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ProteinReDiff denoiser.
//
// Conventions used by every pair-track kernel ("lane owns a pair row"):
//   * A wavefront (64 lanes) processes 32 pair rows at a time.  Lane l = (r, hi) with r = l & 31,
//     hi = l >> 5 owns HALF of the channels of row r.
//   * Canonical lane layout (CLL) of a C-channel row: lane (r, hi) holds KH = C/2 values; local
//     element s <-> channel ch(s, hi) = 8*(s>>2) + 4*hi + (s&3).  I.e. lane (r,0) holds the even
//     16-byte groups of the row and lane (r,1) the odd ones, so one global_load_dwordx4 per group.
//   * All row GEMMs are computed transposed, Out^T[Nout x 32 rows] = W[Nout x K] * X^T[K x 32 rows],
//     with v_mfma_f32_32x32x2_f32: A operand = weights (from LDS), B operand = the lane's own row
//     values.  The D fragment of output block nb, register q, is output channel
//     32*nb + (q&3) + 8*(q>>2) + 4*hi of the lane's row = CLL element s = 16*nb + q, so the result
//     of one GEMM is already in CLL and can be LayerNorm-ed (lane local + one cross-half exchange),
//     gated, fed to the next GEMM as B operand or stored with dwordx4 -- no shuffles, no LDS.
//   * Weights are staged into LDS with the K axis permuted to CLL order (stage_weight_cll) and a
//     row pitch of K+4 floats so that ds_read_b128 A-operand reads are bank-conflict free.
//   * f32-input MFMA is bit-exact fp32 FMA (MI355X_MICROARCH.md), which is what keeps the
//     1e-4 parity budget of BASELINE.json: no bf16/fp16 operands anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));      // operand of the packed fp32 VALU ops (v_pk_add/mul/fma_f32)

#define PRD_DEV __device__ __forceinline__

PRD_DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
PRD_DEV f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

PRD_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// ReLU that PROPAGATES NaN, like torch.relu in the reference (modules.py:306-326, model.py:110-122): fmaxf / v_max_f32 return the
// OTHER operand for a NaN, so a non-finite activation -- an operand beyond the fp16 range under PRD_ARITH_SPLIT16 -- would become a
// silent 0 in the next hidden layer (measured: ab_proj x 1e5 gave finite, WRONG coordinates through the coordinate head's ReLU).
// NaN < 0 is false: the NaN stays, reaches the outputs and trips the sticky flag of prd_step_boundary.
PRD_DEV float relu_nan(float v) { return v < 0.f ? 0.f : v; }
// gate sigmoid on the hardware transcendentals (v_exp_f32 + v_rcp_f32, ~1 ulp each): the gates multiply
// O(1) values, so their 1e-7 relative error is far inside the 1e-5 operator tolerance
PRD_DEV float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// Gates whose weights and bias were staged with the factor -log2(e) (stage_*_cll(..., NEG_LOG2E)): the MFMA result is
// y = -log2(e) * logit and sigmoid(logit) = 1 / (1 + 2^y) -- v_exp_f32, v_add_f32, v_rcp_f32, nothing else.
constexpr float NEG_LOG2E = -1.4426950408889634f;
PRD_DEV float gate_from_scaled(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }

// Buffer addressing: base in a scalar resource descriptor, per-lane byte offset in ONE VGPR that never changes,
// per-access byte offset in an SGPR -- strided stores / loads without a single VALU address instruction.
// The descriptor covers 2 GiB from `p`; callers re-base it (scalar work) to stay inside.
typedef __amdgpu_buffer_rsrc_t prd_rsrc;
PRD_DEV prd_rsrc make_rsrc(const void* p) {
    // the base must be wave-uniform; readfirstlane states it (a no-op when hipcc already knows) -- a base it believes
    // divergent turns every access into a waterfall loop
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
PRD_DEV void buf_store(float v, prd_rsrc rs, unsigned lane_bytes, unsigned uniform_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, lane_bytes, uniform_bytes, 0);
}
PRD_DEV float buf_load(prd_rsrc rs, unsigned lane_bytes, unsigned uniform_bytes) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_bytes, uniform_bytes, 0));
}
constexpr unsigned BUF_OOB = 0x80000000u;       // lane offset outside every descriptor: loads return 0, stores are dropped

// row index inside a 32x32 MFMA D fragment held in register q by a lane of half hi
PRD_DEV int drow32(int q, int hi) { return (q & 3) + 8 * (q >> 2) + 4 * hi; }
// CLL: channel of local element s for half hi
PRD_DEV int cll_ch(int s, int hi) { return 8 * (s >> 2) + 4 * hi + (s & 3); }

PRD_DEV float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
// sum over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1: no LDS crossbar), result in every lane of the row
PRD_DEV float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// ---- row load / store in CLL ------------------------------------------------------------------
template <int C>
PRD_DEV void load_row_cll(const float* __restrict__ row, int hi, bool valid, float (&x)[C / 2]) {
#pragma unroll
    for (int m = 0; m < C / 8; ++m) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(row + 8 * m + 4 * hi);
        x[4 * m + 0] = v.x; x[4 * m + 1] = v.y; x[4 * m + 2] = v.z; x[4 * m + 3] = v.w;
    }
}

// Same through a buffer descriptor: `lane_bytes` = byte offset of the lane's row (+ 16*hi) from the descriptor base, or
// BUF_OOB for a lane without a row (returns zeros) -- unconditional loads (exact vmcnt accounting, so that a prefetch
// can stay in flight), no 64-bit VALU address arithmetic and no per-element select.
template <int C>
PRD_DEV void load_row_cll_buf(prd_rsrc rs, unsigned lane_bytes, float (&x)[C / 2]) {
#pragma unroll
    for (int m = 0; m < C / 8; ++m) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, 32 * m, 0);
        x[4 * m + 0] = __uint_as_float(v[0]); x[4 * m + 1] = __uint_as_float(v[1]);
        x[4 * m + 2] = __uint_as_float(v[2]); x[4 * m + 3] = __uint_as_float(v[3]);
    }
}

template <int C>
PRD_DEV void store_row_cll(float* __restrict__ row, int hi, bool valid, const float (&x)[C / 2]) {
    if (!valid) return;
#pragma unroll
    for (int m = 0; m < C / 8; ++m)
        *reinterpret_cast<float4*>(row + 8 * m + 4 * hi) = make_float4(x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]);
}

// LayerNorm without affine over a CLL row (eps = 1e-5, biased variance; nn.LayerNorm semantics).
// fp32 MFMA and VALU instructions share a SIMD's issue time on gfx950, so every VALU instruction saved is MFMA
// time gained: the sums run on the packed fp32 ops (two elements per instruction, two partial sums).
template <int KH>
PRD_DEV void ln_cll(float (&x)[KH]) {
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KH; k += 2) s2 += f32x2{x[k], x[k + 1]};
    const float mean = xhalf_sum(s2.x + s2.y) * (1.0f / (2 * KH));
    const f32x2 m2 = {mean, mean};
    f32x2 v2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KH; k += 2) {
        const f32x2 d = f32x2{x[k], x[k + 1]} - m2;
        x[k] = d.x;
        x[k + 1] = d.y;
        v2 = __builtin_elementwise_fma(d, d, v2);
    }
    const float rstd = 1.0f / sqrtf(xhalf_sum(v2.x + v2.y) * (1.0f / (2 * KH)) + 1e-5f);
    const f32x2 r2 = {rstd, rstd};
#pragma unroll
    for (int k = 0; k < KH; k += 2) {
        const f32x2 y = f32x2{x[k], x[k + 1]} * r2;
        x[k] = y.x;
        x[k + 1] = y.y;
    }
}

// ---- LDS staging ------------------------------------------------------------------------------
// W: global [nout][K] (row pitch ldw floats, 16-byte aligned rows) -> Wl: LDS, row pitch K+4,
// K axis permuted to CLL order: 16-byte group f of a row goes to slot (f&1)*(K/8) + (f>>1).
// The same image from the TRANSPOSED matrix in memory: Wt [K][nout] (row pitch ldt), i.e. row o of the image is column o of Wt
// (the backward of a linear multiplies by W^T: no transposed copy of the weights has to be made first).  Consecutive threads
// take consecutive o: the four scalar loads of a 16-byte group are coalesced across the workgroup.
template <int K>
PRD_DEV void stage_weight_cll_t(float* Wl, const float* __restrict__ Wt, int nout, int ldt, int tid, int nthreads, float scale = 1.0f) {
    constexpr int F = K / 4;
    const int total = nout * F;
    for (int idx = tid; idx < total; idx += nthreads) {
        const int f = idx / nout, o = idx - f * nout;
        const float* src = Wt + (size_t)(4 * f) * ldt + o;
        const float4 v = make_float4(src[0], src[ldt], src[2 * (size_t)ldt], src[3 * (size_t)ldt]);
        *reinterpret_cast<float4*>(Wl + o * (K + 4) + (f & 1) * (K / 2) + (f >> 1) * 4) = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
    }
}

template <int K>
PRD_DEV void stage_weight_cll(float* Wl, const float* __restrict__ W, int nout, int ldw, int tid, int nthreads, float scale = 1.0f) {
    constexpr int F = K / 4, G = 8;
    // G loads in flight per thread before the first LDS write: a plain load -> write loop pays one L2 round trip per
    // 16 bytes (measured: 4.3 us to stage 70 KB, 6 dependent trips per thread)
    const int total = nout * F;
    for (int base = tid; base < total; base += G * nthreads) {
        float4 v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            const int ic = idx < total ? idx : tid;             // clamped: unconditional loads
            const int o = ic / F, f = ic - o * F;
            v[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 4 * f);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int o = idx / F, f = idx - o * F;
                *reinterpret_cast<float4*>(Wl + o * (K + 4) + (f & 1) * (K / 2) + (f >> 1) * 4) =
                    make_float4(scale * v[u].x, scale * v[u].y, scale * v[u].z, scale * v[u].w);
            }
        }
    }
}
// same, but K axis kept in plain order split in two contiguous halves (for generated B operands)
PRD_DEV void stage_weight_plain(float* Wl, const float* __restrict__ W, int nout, int K, int ldw, int k0, int tid, int nthreads) {
    const int F = K / 4;
    constexpr int G = 8;
    const int total = nout * F;
    for (int base = tid; base < total; base += G * nthreads) {
        float4 v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            const int ic = idx < total ? idx : tid;
            const int o = ic / F, f = ic - o * F;
            v[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + k0 + 4 * f);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int o = idx / F, f = idx - o * F;
                *reinterpret_cast<float4*>(Wl + o * (K + 4) + 4 * f) = v[u];
            }
        }
    }
}
// per-channel vector (bias, LN affine, 1-row weight) in CLL order: vl[hi*(C/2) + s] = v[ch(s,hi)]
PRD_DEV void stage_vec_cll(float* vl, const float* __restrict__ v, int C, int tid, int nthreads, float scale = 1.0f) {
    for (int c = tid; c < C; c += nthreads) {
        const int f = c >> 2, e = c & 3;
        vl[(f & 1) * (C / 2) + (f >> 1) * 4 + e] = v ? scale * v[c] : 0.f;
    }
}

// ---- the row GEMM: acc[nb] += W[32*nb .. 32*nb+31][:] * x  (x in CLL, K channels) --------------
// x holds CLL elements [4*M0, 4*M1) of a K-channel row (M counts 16-byte groups).  The A operand of
// step group m+1 is fetched from LDS before the MFMAs of group m are issued (software pipeline of
// depth 1); the sched_barrier keeps hipcc from hoisting every ds_read of the fully unrolled loop to
// the top, which otherwise costs >256 VGPRs and spills.
template <int K, int NB, int M0, int M1>
PRD_DEV void rowgemm_part(const float* Wl, const float (&x)[4 * (M1 - M0)], f32x16 (&acc)[NB], int r, int hi) {
    const float* wb = Wl + r * (K + 4) + hi * (K / 2);
    float4 w[NB], wn[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) w[nb] = *reinterpret_cast<const float4*>(wb + nb * 32 * (K + 4) + 4 * M0);
#pragma unroll
    for (int m = M0; m < M1; ++m) {
        if (m + 1 < M1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wn[nb] = *reinterpret_cast<const float4*>(wb + nb * 32 * (K + 4) + 4 * (m + 1));
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = mfma32(w[nb].x, x[4 * (m - M0) + 0], acc[nb]);
            acc[nb] = mfma32(w[nb].y, x[4 * (m - M0) + 1], acc[nb]);
            acc[nb] = mfma32(w[nb].z, x[4 * (m - M0) + 2], acc[nb]);
            acc[nb] = mfma32(w[nb].w, x[4 * (m - M0) + 3], acc[nb]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) w[nb] = wn[nb];
    }
}

template <int K, int NB>
PRD_DEV void rowgemm(const float* Wl, const float (&x)[K / 2], f32x16 (&acc)[NB], int r, int hi) {
    rowgemm_part<K, NB, 0, K / 8>(Wl, x, acc, r, hi);
}

// accumulators preloaded with a CLL-staged bias vector (16 consecutive floats per output block and lane half): the
// bias add of the epilogue becomes part of the MFMA chain; vl points at the lane's first element (vl + hi * C/2 + 16 * nb0)
template <int NB>
PRD_DEV void bias_acc(f32x16 (&acc)[NB], const float* vl) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(vl + 16 * nb + 4 * g);
            acc[nb][4 * g] = v.x; acc[nb][4 * g + 1] = v.y; acc[nb][4 * g + 2] = v.z; acc[nb][4 * g + 3] = v.w;
        }
}

// ---- experimental: the row GEMM on the bf16 matrix pipe, operands split three ways ---------------------------------
// x = hi + mid + lo, each part a bf16 obtained by TRUNCATION (exact for an fp32 value: 3 x 8 = 24 mantissa bits); the six
// products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 with fp32 accumulation reproduce the
// fp32 GEMM to ~1e-7 (tools/ubench/bf16x3_bench.hip: 1.16e-7 vs 1.39e-7 for the fp32 MFMA) at 2.1-2.4x its rate.  OPT-IN
// (prd_set_gemm_mode(1)): the default path and the reported numbers are plain fp32 MFMA arithmetic.
// One MFMA consumes 16 channels: lane (r, hi) supplies its CLL elements 8*step .. 8*step+7 (two 16-byte groups of the row).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

PRD_DEV unsigned pack_hi16(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
PRD_DEV void split3(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    const float ah = __uint_as_float(__float_as_uint(a) & 0xffff0000u), bh = __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float ar = a - ah, br = b - bh;
    const float am = __uint_as_float(__float_as_uint(ar) & 0xffff0000u), bm = __uint_as_float(__float_as_uint(br) & 0xffff0000u);
    ph = pack_hi16(ah, bh);
    pm = pack_hi16(am, bm);
    pl = pack_hi16(ar - am, br - bm);
}
// the lane's K/2 channels -> 3 planes x K/16 steps x 8 bf16
template <int K>
PRD_DEV void split3_cll(const float (&x)[K / 2], u32x4 (&p)[3][K / 16]) {
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned ph, pm, pl;
            split3(x[8 * s + 2 * q], x[8 * s + 2 * q + 1], ph, pm, pl);
            p[0][s][q] = ph;
            p[1][s][q] = pm;
            p[2][s][q] = pl;
        }
}
// W: global [nout][K] fp32 -> LDS image [3 planes][nout rows of (2*K/16 + 1) x 16 bytes] (the +1 pads the row pitch so that
// the b128 A-operand reads are bank-conflict free); returns nothing, rows_total = nout is the plane stride in rows
template <int K>
PRD_DEV void stage_weight_b3(u32x4* Wb, const float* __restrict__ W, int nout, int ldw, int tid, int nthreads, float scale = 1.0f) {
    constexpr int S = K / 16, PITCH = 2 * S + 1;
    for (int idx = tid; idx < nout * S * 2; idx += nthreads) {
        const int o = idx / (2 * S), rem = idx - o * (2 * S), st = rem >> 1, h = rem & 1;
        const float4 g0 = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 4 * h);        // CLL elements 8st .. 8st+3
        const float4 g1 = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 8 + 4 * h);    // 8st+4 .. 8st+7
        const float v[8] = {scale * g0.x, scale * g0.y, scale * g0.z, scale * g0.w, scale * g1.x, scale * g1.y, scale * g1.z, scale * g1.w};
        u32x4 pk[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned ph, pm, pl;
            split3(v[2 * q], v[2 * q + 1], ph, pm, pl);
            pk[0][q] = ph;
            pk[1][q] = pm;
            pk[2][q] = pl;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) Wb[((size_t)pl * nout + o) * PITCH + 2 * st + h] = pk[pl];
    }
}
// same, for `nrows` rows of W placed at rows row0.. of an image whose planes are `nout` rows apart
template <int K>
PRD_DEV void stage_weight_b3_rows(u32x4* Wb, int nout, int row0, const float* __restrict__ W, int nrows, int ldw, int tid, int nthreads,
                                  float scale) {
    constexpr int S = K / 16, PITCH = 2 * S + 1;
    for (int idx = tid; idx < nrows * S * 2; idx += nthreads) {
        const int o = idx / (2 * S), rem = idx - o * (2 * S), st = rem >> 1, h = rem & 1;
        const float4 g0 = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 4 * h);
        const float4 g1 = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 8 + 4 * h);
        const float v[8] = {scale * g0.x, scale * g0.y, scale * g0.z, scale * g0.w, scale * g1.x, scale * g1.y, scale * g1.z, scale * g1.w};
        u32x4 pk[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned ph, pm, pl;
            split3(v[2 * q], v[2 * q + 1], ph, pm, pl);
            pk[0][q] = ph;
            pk[1][q] = pm;
            pk[2][q] = pl;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) Wb[((size_t)pl * nout + row0 + o) * PITCH + 2 * st + h] = pk[pl];
    }
}
// acc[nb] += W[row0 + 32*nb .. +31][:] * x for the split row p; Wb / nout as staged by stage_weight_b3
template <int K, int NB>
PRD_DEV void rowgemm_b3(const u32x4* Wb, int nout, int row0, const u32x4 (&p)[3][K / 16], f32x16 (&acc)[NB], int r, int hi) {
    constexpr int S = K / 16, PITCH = 2 * S + 1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            u32x4 w[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[pl] = Wb[((size_t)pl * nout + row0 + nb * 32 + r) * PITCH + 2 * s + hi];
            const int wp[6] = {0, 0, 1, 0, 2, 1}, xp[6] = {0, 1, 0, 2, 0, 1};     // (weight plane, row plane) of the 6 products
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[wp[t]]), __builtin_bit_cast(bf16x8, p[xp[t]][s]),
                                                                   acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keeps hipcc from hoisting every LDS read of the unrolled loops (spills)
        }
}

// Same products with the operands SWAPPED: D = X * W^T, i.e. lane (r, hi) register q holds output channel row0 + 32 nb + r of
// pair row drow32(q, hi) -- four CONSECUTIVE rows per register quad, which is what a channel-major store wants
// (tri_mul_proj's bf16 x 3 operand planes: 8-byte stores of 4 bf16 instead of 2-byte scatters).
template <int K, int NB>
PRD_DEV void rowgemm_b3_t(const u32x4* Wb, int nout, int row0, const u32x4 (&p)[3][K / 16], f32x16 (&acc)[NB], int r, int hi) {
    constexpr int S = K / 16, PITCH = 2 * S + 1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            u32x4 w[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[pl] = Wb[((size_t)pl * nout + row0 + nb * 32 + r) * PITCH + 2 * s + hi];
            const int wp[6] = {0, 0, 1, 0, 2, 1}, xp[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p[xp[t]][s]), __builtin_bit_cast(bf16x8, w[wp[t]]),
                                                                   acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- the row GEMM on the fp16 matrix pipe, operands split in two ---------------------------------------------------------
// x = hi + lo with hi = RTZ_fp16(x) and lo = RTZ_fp16(x - hi) (the subtraction is exact): hi + lo carries 22 bits.  The three
// products hi*hi, hi*lo, lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation give a row GEMM whose error (2^-22 per
// operand) is of the size of the fp32 MFMA's own accumulation error, at 16/3 of its rate, and -- unlike the bf16 x 3 form --
// a weight image of the SAME size as fp32 (2 x 2 bytes), so the kernels whose fp32 weights fill the LDS can use it.
// Range: fp16 saturates at 65504 (RTZ never produces inf); LayerNorm-ed rows, gated attention outputs, ReLU hidden units and
// weights are far inside it.  Weights are staged x 16 (exact) so that their small components keep a normal lo part; the
// factor is taken back out in the epilogue (H2_WSCALE / H2_INV_WSCALE).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr float H2_WSCALE = 16.0f, H2_INV_WSCALE = 1.0f / 16.0f;

// hi = RN_fp16(x), lo = RN_fp16(x - hi): the residual is formed in fp32 by v_fma_mix_f32 (fp16 source half * -1 + fp32 source:
// exact) and the pair of residuals packed by a second v_cvt_pk_f16_f32.  Measured on gfx950 (tools/ubench/valu_rate_bench.hip,
// SIMD cycles per wave64 instruction): v_cvt_pk_f16_f32 / v_fma_mix_f32 4.6, v_fma_mixlo/hi_f16 8.4 (the rate of a
// transcendental) -- so 4 x 4.6 per pair of values beats the round-2 form (v_cvt_pkrtz + v_fma_mixlo_f16 + v_fma_mixhi_f16 =
// 4.6 + 2 x 8.4) although it is one instruction longer, and rounding hi to nearest leaves a signed residual: hi + lo carries
// 24 bits instead of 23.  VALU issue time is what the split kernels are bound by.
// Range: an operand of magnitude >= 65520 rounds to +-inf (the RTZ form saturated at 65504; either is wrong, this one loudly).
// The first conversion is left to the compiler: as the FIRST reader of a value that may come straight out of an MFMA or a
// transcendental it must be an instruction whose hazards hipcc pads (an asm statement's reads are not padded).
typedef _Float16 prd_h16x2 __attribute__((ext_vector_type(2)));
PRD_DEV void split2h(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, prd_h16x2));
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, prd_h16x2));
}
// NE CLL elements x[0 .. NE) (NE a multiple of 8) -> 2 planes x NE/8 operand registers of 8 fp16
template <int NE>
PRD_DEV void split2h_cll(const float (&x)[NE], u32x4 (&p)[2][NE / 8]) {
#pragma unroll
    for (int s = 0; s < NE / 8; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2h(x[8 * s + 2 * q], x[8 * s + 2 * q + 1], h, l);
            p[0][s][q] = h;
            p[1][s][q] = l;
        }
}
// LDS image of W [nout][K] fp32: plane pl (hi / lo) at Wh + pl * nout * (K/8) (in 16-byte units), row o = K/8 slots of 8 fp16
// without padding; slot j of row o is stored at h2_slot(o, j): j ^ (o & 15) for rows of >= 16 slots, j ^ ((o >> 1) & 7) for rows of 8
// slots (K = 64: two rows share a 256-byte bank line, the row's low bit already selects its half) -- the sixteen lanes of a
// ds_read_b128 group (non-contiguous lane sets, MI355X_MICROARCH.md: LDS) read one logical slot of sixteen rows that are distinct
// mod 16 -> sixteen different 16-byte bank groups (4 LDS cycles per read: tools/ubench/lds_pattern_bench.hip).
template <int K>
PRD_DEV int h2_slot(int row, int j) {
    constexpr int SL = K / 8;                     // slots per row (8: K = 64, 32: K = 256)
    if (SL >= 16) return j ^ (row & 15);
    return j ^ ((row >> 1) & (SL - 1));           // SL = 8: rows 2a, 2a+1 differ in bit 3 of the bank-group index already
}
// G row pieces (2 x 16 bytes each) in flight per thread before the first split / LDS write, as in stage_weight_cll.  (Measured:
// the prologue of a row kernel is NOT these round trips -- pair_tail 10.0k -> 8.8k cycles for 151 KB on 768 threads, while an
// L2-warm copy of the same bytes takes 4.1k, 3.2k with global_load_lds: tools/ubench/lds_fill_bench.hip.  The image is L2-cold
// at kernel start -- the previous kernel streamed 80-130 MB through the 4 MB L2s -- and 32 CUs of an XCD ask for the same lines.)
template <int K>
PRD_DEV void stage_weight_h2_rows(u32x4* Wh, int nout, int row0, const float* __restrict__ W, int nrows, int ldw, int tid, int nthreads,
                                  float scale) {
    constexpr int S = K / 16, G = 4;
    const int total = nrows * S * 2;
    for (int base = tid; base < total; base += G * nthreads) {
        float4 g0[G], g1[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            const int ic = idx < total ? idx : tid;             // clamped: unconditional loads
            const int o = ic / (2 * S), rem = ic - o * (2 * S), st = rem >> 1, h = rem & 1;
            g0[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 4 * h);        // CLL elements 8st .. 8st+3
            g1[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + 16 * st + 8 + 4 * h);    // 8st+4 .. 8st+7
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int o = idx / (2 * S), rem = idx - o * (2 * S), st = rem >> 1, h = rem & 1;
                const float v[8] = {scale * g0[u].x, scale * g0[u].y, scale * g0[u].z, scale * g0[u].w,
                                    scale * g1[u].x, scale * g1[u].y, scale * g1[u].z, scale * g1[u].w};
                u32x4 ph, pl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned a, b;
                    split2h(v[2 * q], v[2 * q + 1], a, b);
                    ph[q] = a;
                    pl[q] = b;
                }
                const int slot = h2_slot<K>(row0 + o, 2 * st + h);
                Wh[(size_t)(row0 + o) * (K / 8) + slot] = ph;
                Wh[(size_t)(nout + row0 + o) * (K / 8) + slot] = pl;
            }
        }
    }
}
template <int K>
PRD_DEV void stage_weight_h2(u32x4* Wh, const float* __restrict__ W, int nout, int ldw, int tid, int nthreads, float scale) {
    stage_weight_h2_rows<K>(Wh, nout, 0, W, nout, ldw, tid, nthreads, scale);
}
// ... and from the transposed matrix in memory, Wt [K][nout] (see stage_weight_cll_t)
template <int K>
PRD_DEV void stage_weight_h2_t(u32x4* Wh, const float* __restrict__ Wt, int nout, int ldt, int tid, int nthreads, float scale) {
    constexpr int S = K / 16;
    const int total = nout * S * 2;
    for (int idx = tid; idx < total; idx += nthreads) {
        const int rem = idx / nout, o = idx - rem * nout, st = rem >> 1, h = rem & 1;
        const float* s0 = Wt + (size_t)(16 * st + 4 * h) * ldt + o;        // CLL elements 8st .. 8st+3: k = 16 st + 4 h + e
        const float* s1 = s0 + (size_t)8 * ldt;                            // 8st+4 .. 8st+7: k = 16 st + 8 + 4 h + e
        const float v[8] = {scale * s0[0], scale * s0[ldt], scale * s0[2 * (size_t)ldt], scale * s0[3 * (size_t)ldt],
                            scale * s1[0], scale * s1[ldt], scale * s1[2 * (size_t)ldt], scale * s1[3 * (size_t)ldt]};
        u32x4 ph, pl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a, b;
            split2h(v[2 * q], v[2 * q + 1], a, b);
            ph[q] = a;
            pl[q] = b;
        }
        const int slot = h2_slot<K>(o, 2 * st + h);
        Wh[(size_t)o * (K / 8) + slot] = ph;
        Wh[(size_t)(nout + o) * (K / 8) + slot] = pl;
    }
}
// NATURAL K order (for operands generated per K step, not rows in CLL): slot j of row o holds W[o][k0 + 8 j .. + 7]; K must be a
// multiple of 128 (16 | K/8), slot j is stored at j ^ (o & 15).  K step s of lane (r, hi) then covers k = 16 s + 8 hi .. + 7.
PRD_DEV void stage_weight_h2_nat(u32x4* Wh, const float* __restrict__ W, int nout, int K, int ldw, int k0, int tid, int nthreads, float scale) {
    const int SL = K / 8;
    constexpr int G = 4;                                        // pieces in flight per thread (see stage_weight_h2_rows)
    const int total = nout * SL;
    for (int base = tid; base < total; base += G * nthreads) {
        float4 g0[G], g1[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            const int ic = idx < total ? idx : tid;
            const int o = ic / SL, j = ic - o * SL;
            g0[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + k0 + 8 * j);
            g1[u] = *reinterpret_cast<const float4*>(W + (size_t)o * ldw + k0 + 8 * j + 4);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int o = idx / SL, j = idx - o * SL;
                const float v[8] = {g0[u].x, g0[u].y, g0[u].z, g0[u].w, g1[u].x, g1[u].y, g1[u].z, g1[u].w};
                u32x4 ph, pl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned a_, b_;
                    split2h(scale * v[2 * q], scale * v[2 * q + 1], a_, b_);
                    ph[q] = a_;
                    pl[q] = b_;
                }
                const int slot = j ^ (o & 15);
                Wh[(size_t)o * SL + slot] = ph;
                Wh[(size_t)(nout + o) * SL + slot] = pl;
            }
        }
    }
}
// one K step of the natural-order form: acc[nb] += W[32 nb + r][16 st + 8 hi ..] * f (f = the lane's 8 generated operand values)
template <int NB>
PRD_DEV void h2_nat_step(const u32x4* Wh, int nout, int SL, int st, const float (&f)[8], f32x16 (&acc)[NB], int r, int hi) {
    u32x4 ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b;
        split2h(f[2 * q], f[2 * q + 1], a, b);
        ph[q] = a;
        pl[q] = b;
    }
    // rows 32 nb + r of both planes share the swizzle (o & 15 = r & 15): one slot per step, the row bases are loop invariants
    // (the K loops of the outer-linear / OPM / pair-init kernels are VALU-issue bound: every address instruction counts)
    const unsigned slot = (unsigned)(2 * st + hi) ^ (unsigned)(r & 15);
    const u32x4* wh0 = Wh + (unsigned)r * (unsigned)SL + slot;
    const u32x4* wl0 = Wh + ((unsigned)nout + (unsigned)r) * (unsigned)SL + slot;
    u32x4 wh[NB], wl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { wh[nb] = wh0[nb * 32 * SL]; wl[nb] = wl0[nb * 32 * SL]; }
    // the three products of an accumulator are issued NB accumulators apart (no back-to-back dependent MFMAs)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh[nb]), __builtin_bit_cast(f16x8_t, ph), acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh[nb]), __builtin_bit_cast(f16x8_t, pl), acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wl[nb]), __builtin_bit_cast(f16x8_t, ph), acc[nb], 0, 0, 0);
}
// acc[nb] += W[row0 + 32 nb .. +31][16 S0 .. 16 S1) * x for the split row p (K-steps S0 .. S1 of the image's K)
template <int K, int NB, int S0, int S1>
PRD_DEV void rowgemm_h2_part(const u32x4* Wh, int nout, int row0, const u32x4 (&p)[2][S1 - S0], f32x16 (&acc)[NB], int r, int hi) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = S0; s < S1; ++s) {
            const int o = row0 + nb * 32 + r;
            const int slot = h2_slot<K>(o, 2 * s + hi);
            const u32x4 wh = Wh[(size_t)o * (K / 8) + slot], wl = Wh[(size_t)(nout + o) * (K / 8) + slot];
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh), __builtin_bit_cast(f16x8_t, p[0][s - S0]), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh), __builtin_bit_cast(f16x8_t, p[1][s - S0]), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wl), __builtin_bit_cast(f16x8_t, p[0][s - S0]), acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keeps hipcc from hoisting every LDS read of the unrolled loops (spills)
        }
}
template <int K, int NB>
PRD_DEV void rowgemm_h2(const u32x4* Wh, int nout, int row0, const u32x4 (&p)[2][K / 16], f32x16 (&acc)[NB], int r, int hi) {
    rowgemm_h2_part<K, NB, 0, K / 16>(Wh, nout, row0, p, acc, r, hi);
}

template <int NB>
PRD_DEV void zero_acc(f32x16 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
}

// plain v_max3_f32 / v_max_f32: fmaxf() adds a canonicalising v_max per operand that comes out of an MFMA, and on
// gfx950 every VALU instruction costs matrix-pipe time (fp32 MFMA and VALU share the SIMD's issue cycles)
PRD_DEV float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
PRD_DEV float max2f(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// cross-lane reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the
// gfx950 half/row swap instructions: VALU only, no LDS round trip (ds_bpermute)
PRD_DEV float rows4_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max2f(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max2f(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
PRD_DEV float rows4_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// order LDS traffic of one wave against itself (cross-lane hand-off through LDS without s_barrier)
PRD_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- task distribution over persistent waves ----------------------------------------------------
// The row kernels run as ~one persistent workgroup per CU (weights staged into LDS once).  With a
// static round-robin the 3200 32-row tasks of an N=320 complex over 2048..3072 resident waves finish in
// ceil() rounds; a device-side queue lets every wave pull its next task instead.  `queue` points to 256
// zero-initialised ints (one counter per XCD, 128 B apart) owned by the caller; every wave performs exactly one failing fetch, so the wave
// that draws ticket ntask + nwaves - 1 is the last user and resets the counter for the next launch
// (launches sharing a counter must be stream-ordered).  queue == nullptr -> static round-robin.
struct WaveTasks {
    int* q;
    long ntask, cur, stride;
    int shard, nshard, count, last;
    PRD_DEV WaveTasks(int* queue, long ntask_, int waves_per_wg) : q(queue), ntask(ntask_) {
        stride = (long)gridDim.x * waves_per_wg;
        // static order is WAVE-major (slot = wave * #workgroups + workgroup): the tasks of a partial last round land
        // on as many different CUs as possible instead of filling the first few workgroups
        cur = (long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x;
        // one counter per XCD (workgroup b is observed to run on XCD b % 8; only speed depends on it):
        // a single word saturates near 90 dequeues/us, far below what 3000 waves ask for
        nshard = gridDim.x < 8 ? (int)gridDim.x : 8;
        shard = (int)(blockIdx.x % nshard);
        count = (int)((ntask - shard + nshard - 1) / nshard);                       // tasks shard, shard+nshard, ...
        const int wgs = (int)((gridDim.x - shard + nshard - 1) / nshard);           // workgroups feeding on this shard
        last = count + wgs * waves_per_wg - 1;                                       // ticket of the final (failing) fetch
    }
    PRD_DEV long next() {                       // task id, or -1 when the work is exhausted
        if (!q) {
            // readfirstlane: the id is the same in all 64 lanes, but only an SGPR value lets hipcc do the
            // task -> (row, block) index arithmetic on the scalar unit.  fp32 MFMA and VALU instructions share
            // the SIMD's issue time on gfx950 (tools/ubench/coissue_bench.hip), so VALU address math is not free.
            const long t = (long)__builtin_amdgcn_readfirstlane((int)cur);
            cur += stride;
            return t < ntask ? t : -1;
        }
        int t = 0;
        if ((threadIdx.x & 63) == 0) {
            t = atomicAdd(q + 32 * shard, 1);              // counters 128 B apart: one L2 line / channel each
            if (t == last) atomicExch(q + 32 * shard, 0);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        return t < count ? (long)t * nshard + shard : -1;
    }
};

// every entry point whose kernels depend on the arithmetic takes it as an argument (prd_hip.h: PRD_ARITH_* in the low byte,
// PRD_TUNE_* kernel-selection switches above it); the library keeps no state and reads no environment variable.
// PRD_SPLIT_ARITH(arith): validates the word, leaves the arithmetic in `arith`, the switches in `tune`, the whole word in `arith_full`
#define PRD_SPLIT_ARITH(a)                                                                       \
    const int arith_full = (a);                                                                  \
    const int tune = (a) >> 8;                                                                   \
    (void)tune; (void)arith_full;                                                                \
    if ((a) < 0 || ((a) & 0xff) > 1) return -1;                                                  \
    (a) &= 0xff
#define PRD_TGET_TA_VARIANT(t) ((t) & 15)                   /* first-generation attention core: 0 = default dispatch */
#define PRD_TGET_TA2_NO_V3(t) (((t) >> 4) & 1)
#define PRD_TGET_TA2_NO_LONG(t) (((t) >> 5) & 1)
#define PRD_TGET_TA2_FLAGS(t) ((((t) >> 6) & 1) ? (((t) >> 7) & 31) : -1)
#define PRD_TGET_TA2_NO_TAIL_SPLIT(t) (((t) >> 19) & 1)
#define PRD_TGET_TA2_NO_GV(t) (((t) >> 21) & 1)
#define PRD_TGET_TMP_NW16(t) (((t) >> 22) & 1)
#define PRD_TGET_TMS_D3(t) ((((t) >> 13) & 3) == 3)             /* 8 waves, three chunks of operands in flight */
#define PRD_TGET_OL_GEN2(t) (((t) >> 12) & 1)
#define PRD_TGET_TMS_NW(t) ((((t) >> 13) & 3) == 1 ? 12 : (((t) >> 13) & 3) == 2 ? 16 : 8)

static inline int prd_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int prd_round_up(int a, int b) { return prd_ceil_div(a, b) * b; }
