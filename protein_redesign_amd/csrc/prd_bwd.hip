// Backward kernels of the pair track, first set: triangle multiplication (reference modules.py:262-274 under autograd).
//
//   forward:  x = LN(pair);  ab = m2 * sigmoid(Wg x + bg) * (Wp x + bp)  -> operands A | B (channel-major, tri_mul_proj);
//             O = contraction(A, B) (tri_mul_contract);  y = sigmoid(Wog x + bog) * (Wo LN(O) + bo)   (tri_mul_out)
//   backward, given dy:
//     tri_mul_out_bwd   (row pass)   dz = dy * g, dgp = dy * z * g (1 - g)   [kept for the weight-gradient GEMMs],
//                                    dO = LN'(Wo^T dz) written channel-major, dx1 = Wog^T dgp
//     tri_mul_contract  (the forward contraction kernel, called on transposed operands)
//                                    dA[i][k] = sum_j dO[i][j] B[j][k],   dB[j][k] = sum_i dO[i][j] A[i][k]
//     tri_mul_proj_bwd  (row pass)   dpp = dAB * m2 * s, dpg = dAB * m2 * pp * s (1 - s)   [kept for the weight gradients],
//                                    dpair = LN'(Wp^T dpp + Wg^T dpg + dx1)
//   The weight gradients are reductions over all N^2 rows, dW = dOut^T In: plain tall-skinny GEMMs, left to the BLAS library on
//   the host side (training.py); the transposes of the channel-major operands between the passes are torch copies.
// Same "lane owns a pair row" scheme as the forward row kernels (prd_common.h); fp32 MFMA row GEMMs (gradients are not on the
// sampling hot path); the transposed weights are passed in by the caller.
#include "prd_common.h"
#include <type_traits>
#include "../../include/prd_hip.h"
#include <mutex>

namespace {

// LayerNorm (no affine) of a CLL row in place, returning 1/std (needed by the backward formula)
template <int KH>
PRD_DEV float ln_cll_rstd(float (&x)[KH]) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) s += x[k];
    const float mean = xhalf_sum(s) * (1.0f / (2 * KH));
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) { x[k] -= mean; v += x[k] * x[k]; }
    const float rstd = 1.0f / sqrtf(xhalf_sum(v) * (1.0f / (2 * KH)) + 1e-5f);
#pragma unroll
    for (int k = 0; k < KH; ++k) x[k] *= rstd;
    return rstd;
}
// d/dx of y = LN(x): dx = rstd * (dy - mean(dy) - y * mean(dy * y))   (y = the normalised row)
template <int KH>
PRD_DEV void ln_cll_bwd(float (&dy)[KH], const float (&y)[KH], float rstd) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) { s1 += dy[k]; s2 += dy[k] * y[k]; }
    const float m1 = xhalf_sum(s1) * (1.0f / (2 * KH)), m2 = xhalf_sum(s2) * (1.0f / (2 * KH));
#pragma unroll
    for (int k = 0; k < KH; ++k) dy[k] = rstd * (dy[k] - m1 - y[k] * m2);
}

int grid_for(long tasks, int per_wg, int cap) {
    long g = (tasks + per_wg - 1) / per_wg;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_mul_out_bwd_kernel(
    float* __restrict__ dz_out, float* __restrict__ dgp_out, float* __restrict__ dO, float* __restrict__ dx1,
    const float* __restrict__ dy, const float* __restrict__ pair, const float* __restrict__ O,
    const float* __restrict__ wo, const float* __restrict__ bo, const float* __restrict__ wog, const float* __restrict__ bog,
    const float* __restrict__ woT, const float* __restrict__ wogT, int b, int N, int ldn,
    float* __restrict__ x_out, float* __restrict__ lo_out, int dO_bch) {
    constexpr int KH = P / 2, NB = P / 32, WSZ = P * (P + 4);
    extern __shared__ __attribute__((aligned(16))) float smem_b1[];
    float* Wol = smem_b1;
    float* Wgl = Wol + WSZ;
    float* WoTl = Wgl + WSZ;
    float* WgTl = WoTl + WSZ;
    float* bol = WgTl + WSZ;
    float* bgl = bol + P;
    const int NT = NW * 64;
    stage_weight_cll<P>(Wol, wo, P, P, threadIdx.x, NT);
    stage_weight_cll<P>(Wgl, wog, P, P, threadIdx.x, NT);
    // W^T images: from the caller's transposed copies, or (null) straight from W read column-wise
    if (woT) stage_weight_cll<P>(WoTl, woT, P, P, threadIdx.x, NT); else stage_weight_cll_t<P>(WoTl, wo, P, P, threadIdx.x, NT);
    if (wogT) stage_weight_cll<P>(WgTl, wogT, P, P, threadIdx.x, NT); else stage_weight_cll_t<P>(WgTl, wog, P, P, threadIdx.x, NT);
    stage_vec_cll(bol, bo, P, threadIdx.x, NT);
    stage_vec_cll(bgl, bog, P, threadIdx.x, NT);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;
        const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const long off = (bi * N + jj) * P;
        float x[KH], g[KH];
        load_row_cll<P>(pair + off, hi, valid, x);
        ln_cll<KH>(x);
        if (x_out) store_row_cll<P>(x_out + off, hi, valid, x);        // LN(pair) rows for the caller's weight gradients
        {
            f32x16 ag[NB];
            zero_acc(ag);
            rowgemm<P, NB>(Wgl, x, ag, r, hi);
#pragma unroll
            for (int s = 0; s < KH; ++s) g[s] = sigmoidf_(ag[s >> 4][s & 15] + bgl[hi * KH + s]);
        }
        // O / dO through buffer descriptors: the lane part of the address once, the channel stride as a scalar offset (as the
        // forward's tri_mul_out_kernel; 64 scattered 64-bit address computations per task otherwise)
        const unsigned cbytes = (unsigned)N * (unsigned)ldn * 4u;
        const unsigned lane_o = valid ? (unsigned)r * 4u + (unsigned)(4 * hi) * cbytes : BUF_OOB;
        float lo[KH];
        {
            const prd_rsrc ro = make_rsrc(O + (((long)bb * P) * N + i) * ldn + vb * 32);
#pragma unroll
            for (int s = 0; s < KH; ++s) lo[s] = buf_load(ro, lane_o, (unsigned)(8 * (s >> 2) + (s & 3)) * cbytes);
        }
        const float rstd_o = ln_cll_rstd<KH>(lo);
        if (lo_out) store_row_cll<P>(lo_out + off, hi, valid, lo);      // LN(O) rows, likewise
        float dz[KH], dgp[KH];
        {
            f32x16 az[NB];
            zero_acc(az);
            rowgemm<P, NB>(Wol, lo, az, r, hi);
            float d[KH];
            load_row_cll<P>(dy + off, hi, valid, d);
#pragma unroll
            for (int s = 0; s < KH; ++s) {
                const float z = az[s >> 4][s & 15] + bol[hi * KH + s];
                dz[s] = d[s] * g[s];
                dgp[s] = d[s] * z * g[s] * (1.0f - g[s]);
            }
        }
        store_row_cll<P>(dz_out + off, hi, valid, dz);
        store_row_cll<P>(dgp_out + off, hi, valid, dgp);
        {   // dO = LN'(Wo^T dz), channel-major like O
            f32x16 a[NB];
            zero_acc(a);
            rowgemm<P, NB>(WoTl, dz, a, r, hi);
            float dlo[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) dlo[s] = a[s >> 4][s & 15];
            ln_cll_bwd<KH>(dlo, lo, rstd_o);
            {
                const prd_rsrc rdo = make_rsrc(dO + (((long)bb * dO_bch) * N + i) * ldn + vb * 32);
#pragma unroll
                for (int s = 0; s < KH; ++s) buf_store(dlo[s], rdo, lane_o, (unsigned)(8 * (s >> 2) + (s & 3)) * cbytes);
            }
        }
        {   // gate path of dx
            f32x16 a[NB];
            zero_acc(a);
            rowgemm<P, NB>(WgTl, dgp, a, r, hi);
            float d1[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) d1[s] = a[s >> 4][s & 15];
            store_row_cll<P>(dx1 + off, hi, valid, d1);
        }
    }
}

template <int P, int NW, bool B3>                   // B3: the eight row GEMMs in the fp16 x 2 split form (gemm mode 1), weights x 16
__global__ __launch_bounds__(NW * 64) void tri_mul_proj_bwd_kernel(
    float* __restrict__ dpair, float* __restrict__ dpp_out, float* __restrict__ dpg_out,
    const float* __restrict__ dAB, const float* __restrict__ dx1, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ wpT, const float* __restrict__ wgT, int b, int N, int ldn, int incoming) {
    constexpr int KH = P / 2, NB = P / 32, OUT = 2 * P;
    extern __shared__ __attribute__((aligned(16))) float smem_b3[];
    constexpr int WSZ = B3 ? OUT * P : OUT * (P + 4), WTSZ = B3 ? P * OUT : P * (OUT + 4);      // floats per staged matrix
    float* Wpl = smem_b3;                        // [2P][P+4] (fp32) or fp16 hi | lo planes of the same size as the fp32 matrix
    float* Wgl = Wpl + WSZ;
    float* WpTl = Wgl + WSZ;                     // [P][2P+4]: rows = input channels of the projection, K = its 2P outputs
    float* WgTl = WpTl + WTSZ;
    float* bpl = WgTl + WTSZ;                    // [2P] CLL
    float* bgl = bpl + OUT;
    const int NT = NW * 64;
    constexpr float ASC = B3 ? H2_INV_WSCALE : 1.0f;
    if (B3) {
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wpl), wp, OUT, P, threadIdx.x, NT, H2_WSCALE);
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wgl), wg, OUT, P, threadIdx.x, NT, H2_WSCALE);
        if (wpT) stage_weight_h2<OUT>(reinterpret_cast<u32x4*>(WpTl), wpT, P, OUT, threadIdx.x, NT, H2_WSCALE);
        else stage_weight_h2_t<OUT>(reinterpret_cast<u32x4*>(WpTl), wp, P, P, threadIdx.x, NT, H2_WSCALE);
        if (wgT) stage_weight_h2<OUT>(reinterpret_cast<u32x4*>(WgTl), wgT, P, OUT, threadIdx.x, NT, H2_WSCALE);
        else stage_weight_h2_t<OUT>(reinterpret_cast<u32x4*>(WgTl), wg, P, P, threadIdx.x, NT, H2_WSCALE);
    } else {
        stage_weight_cll<P>(Wpl, wp, OUT, P, threadIdx.x, NT);
        stage_weight_cll<P>(Wgl, wg, OUT, P, threadIdx.x, NT);
        if (wpT) stage_weight_cll<OUT>(WpTl, wpT, P, OUT, threadIdx.x, NT); else stage_weight_cll_t<OUT>(WpTl, wp, P, P, threadIdx.x, NT);
        if (wgT) stage_weight_cll<OUT>(WgTl, wgT, P, OUT, threadIdx.x, NT); else stage_weight_cll_t<OUT>(WgTl, wg, P, P, threadIdx.x, NT);
    }
    stage_vec_cll(bpl, bp, OUT, threadIdx.x, NT);
    stage_vec_cll(bgl, bg, OUT, threadIdx.x, NT);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const int vb = (int)(task % nvb);
        const long bu = task / nvb;
        const int bb = (int)(bu / N), u = (int)(bu - (long)bb * N);
        const int v = vb * 32 + r;
        const bool valid = v < N;
        const int vv = valid ? v : 0;
        // operand position [u][v] <-> pair position (u, v) (outgoing) or (v, u) (incoming), as in tri_mul_proj
        const long prow = (long)bb * N * N + (incoming ? (long)vv * N + u : (long)u * N + vv);
        float x[KH];
        load_row_cll<P>(pair + prow * P, hi, valid, x);
        const float rstd_x = ln_cll_rstd<KH>(x);
        const float m2 = valid ? mask[bu] * mask[(long)bb * N + vv] : 0.f;
        f32x16 adx[NB];
        zero_acc(adx);
        u32x4 xs[2][P / 16];
        if (B3) split2h_cll<KH>(x, xs);
        if constexpr (B3) {
            // dAB through a buffer descriptor: the lane part of the address once, the channel stride as a scalar offset (64
            // scattered 64-bit address computations per task otherwise)
            const prd_rsrc rdab = make_rsrc(dAB + (((long)bb * OUT) * N + u) * ldn + vb * 32);
            const unsigned cbytes = (unsigned)N * (unsigned)ldn * 4u;
            const unsigned lo_dab = valid ? (unsigned)r * 4u + (unsigned)(4 * hi) * cbytes : BUF_OOB;
            // Split form: one 32-channel block (16 CLL elements per lane) of one operand at a time.  The whole-operand form below
            // keeps ap, ag, dpp, dpg and both split copies of 32 elements alive next to x, its split and adx: > 256 VGPRs, 592 bytes
            // of scratch per lane at P = 64.
#define PRD_PB_PART(H_, NB_)                                                                                            \
            {                                                                                                           \
                __builtin_amdgcn_sched_barrier(0);      /* no loads / LDS reads of this part hoisted into the previous one */ \
                f32x16 ap1[1], ag1[1];                                                                                  \
                zero_acc(ap1);                                                                                          \
                zero_acc(ag1);                                                                                          \
                rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wpl), OUT, (H_) * P + (NB_) * 32, xs, ap1, r, hi);      \
                rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wgl), OUT, (H_) * P + (NB_) * 32, xs, ag1, r, hi);      \
                float dpp[16], dpg[16];                                                                                 \
                _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                        \
                    const int s_ = 16 * (NB_) + q;                                                                      \
                    const float dab = buf_load(rdab, lo_dab, (unsigned)((H_) * P + 8 * (s_ >> 2) + (s_ & 3)) * cbytes); \
                    const float pp = ap1[0][q] * ASC + bpl[hi * P + (H_) * KH + s_];                                    \
                    const float sg = sigmoidf_(ag1[0][q] * ASC + bgl[hi * P + (H_) * KH + s_]);                         \
                    dpp[q] = dab * m2 * sg;                                                                             \
                    dpg[q] = dab * m2 * pp * sg * (1.0f - sg);                                                          \
                }                                                                                                       \
                if (valid) {        /* CLL elements 16 nb .. +15 = four 16-byte groups of the row: channels 32 nb + 8 g + 4 hi */ \
                    float* d0 = dpp_out + prow * OUT + (H_) * P + 32 * (NB_) + 4 * hi;                                  \
                    float* d1 = dpg_out + prow * OUT + (H_) * P + 32 * (NB_) + 4 * hi;                                  \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                     \
                        *reinterpret_cast<float4*>(d0 + 8 * g) = make_float4(dpp[4 * g], dpp[4 * g + 1], dpp[4 * g + 2], dpp[4 * g + 3]); \
                        *reinterpret_cast<float4*>(d1 + 8 * g) = make_float4(dpg[4 * g], dpg[4 * g + 1], dpg[4 * g + 2], dpg[4 * g + 3]); \
                    }                                                                                                   \
                }                                                                                                       \
                u32x4 ps[2][2], gs[2][2];                                                                               \
                split2h_cll<16>(dpp, ps);                                                                               \
                split2h_cll<16>(dpg, gs);                                                                               \
                constexpr int S0_ = (H_) * (KH / 8) + 2 * (NB_);      /* K steps of 16 on the 2P-wide K axis of W^T */   \
                rowgemm_h2_part<OUT, NB, S0_, S0_ + 2>(reinterpret_cast<const u32x4*>(WpTl), P, 0, ps, adx, r, hi);     \
                rowgemm_h2_part<OUT, NB, S0_, S0_ + 2>(reinterpret_cast<const u32x4*>(WgTl), P, 0, gs, adx, r, hi);     \
            }
            PRD_PB_PART(0, 0)
            if constexpr (NB == 2) PRD_PB_PART(0, 1)
            PRD_PB_PART(1, 0)
            if constexpr (NB == 2) PRD_PB_PART(1, 1)
#undef PRD_PB_PART
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // h = 0: the a operand (output channels 0 .. P-1), h = 1: the b operand
                f32x16 ap[NB], ag[NB];
                zero_acc(ap);
                zero_acc(ag);
                if (B3) {
                    rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wpl), OUT, h * P, xs, ap, r, hi);
                    rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wgl), OUT, h * P, xs, ag, r, hi);
                } else {
                    rowgemm<P, NB>(Wpl + h * P * (P + 4), x, ap, r, hi);
                    rowgemm<P, NB>(Wgl + h * P * (P + 4), x, ag, r, hi);
                }
                float dpp[KH], dpg[KH];
    #pragma unroll
                for (int s = 0; s < KH; ++s) {
                    const int ch = h * P + cll_ch(s, hi);
                    const float dab = valid ? dAB[(((long)bb * OUT + ch) * N + u) * ldn + vv] : 0.f;
                    const float pp = ap[s >> 4][s & 15] * ASC + bpl[hi * P + h * KH + s];
                    const float sg = sigmoidf_(ag[s >> 4][s & 15] * ASC + bgl[hi * P + h * KH + s]);
                    dpp[s] = dab * m2 * sg;
                    dpg[s] = dab * m2 * pp * sg * (1.0f - sg);
                }
                // kept for the weight-gradient GEMMs: row layout [pair position][2P], this half at columns h P ..
                store_row_cll<P>(dpp_out + prow * OUT + h * P, hi, valid, dpp);
                store_row_cll<P>(dpg_out + prow * OUT + h * P, hi, valid, dpg);
                // dx += Wp[h]^T dpp + Wg[h]^T dpg: the half is CLL elements [32 h, 32 h + 32) of the 2P-wide K axis = groups [8h, 8h+8)
                if (B3) {                            // K steps of 16 (8 per lane): the half is steps [KH/8 h, KH/8 (h + 1)) of the 2P-wide K axis
                    u32x4 ps[2][KH / 8], gs[2][KH / 8];
                    split2h_cll<KH>(dpp, ps);
                    split2h_cll<KH>(dpg, gs);
                    if (h == 0) {
                        rowgemm_h2_part<OUT, NB, 0, KH / 8>(reinterpret_cast<const u32x4*>(WpTl), P, 0, ps, adx, r, hi);
                        rowgemm_h2_part<OUT, NB, 0, KH / 8>(reinterpret_cast<const u32x4*>(WgTl), P, 0, gs, adx, r, hi);
                    } else {
                        rowgemm_h2_part<OUT, NB, KH / 8, KH / 4>(reinterpret_cast<const u32x4*>(WpTl), P, 0, ps, adx, r, hi);
                        rowgemm_h2_part<OUT, NB, KH / 8, KH / 4>(reinterpret_cast<const u32x4*>(WgTl), P, 0, gs, adx, r, hi);
                    }
                } else if (h == 0) {
                    rowgemm_part<OUT, NB, 0, KH / 4>(WpTl, dpp, adx, r, hi);
                    rowgemm_part<OUT, NB, 0, KH / 4>(WgTl, dpg, adx, r, hi);
                } else {
                    rowgemm_part<OUT, NB, KH / 4, KH / 2>(WpTl, dpp, adx, r, hi);
                    rowgemm_part<OUT, NB, KH / 4, KH / 2>(WgTl, dpg, adx, r, hi);
                }
            }
        }
        float dx[KH];
        load_row_cll<P>(dx1 + prow * P, hi, valid, dx);
#pragma unroll
        for (int s = 0; s < KH; ++s) dx[s] += adx[s >> 4][s & 15] * ASC;
        ln_cll_bwd<KH>(dx, x, rstd_x);
        store_row_cll<P>(dpair + prow * P, hi, valid, dx);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Triangle attention backward, core (reference modules.py:236-243 -> 185-225 under autograd).  One workgroup per (pair row, head),
// persistent.  Given dog = d(gated per-head output) [b,N,N,64] (= W_o^T d(update), a row GEMM done by the caller) it recomputes
// q, k, v, gate of the row (fp32 MFMA row GEMMs into LDS), then
//   pass A (wave = 16-query tile):  sweep 1 over the keys: softmax statistics m, l and o = P v;  delta = do . o;  d(gate
//                                   pre-activation);  sweep 2: p = exp(s - m) / l,  dS = p (do . v_j - delta),  dq += dS k_j
//   pass B (wave = 16-key tile):    sweep over the queries: p and dS again from the stored m, l, delta;  dk += dS q,  dv += p do
// on v_mfma_f32_16x16x4_f32 in the "swapped" form of the forward kernels: a logit tile comes out as (keys in registers, query =
// lane) in pass A and (queries in registers, key = lane) in pass B, which is the B-operand layout of the products that contract
// over the tile's register index (P v, dS k, dS^T q, P^T do) -- no transposes; their A operands are read column-wise from the
// same [position][16] LDS arrays (pitch 20: both the row-wise float4 reads and the column-wise scalar reads are conflict-free).
// The logits are recomputed three times (sweep 1, sweep 2, pass B).  Masked keys (mask_2d < 0.5) have their logit REPLACED by
// -2^15 in the forward: no gradient flows into q, k through them, while v still receives p * do.
// Output: dqkvg[b,N,N,4,64] by pair position = d(W_q x) | d(W_k x) | d(W_v x) | d(gate pre-act.), channels head-major; the
// projections' input gradient, the LayerNorm backward and the weight gradients are row GEMMs / slab reductions on the caller's side.
constexpr int TB_PITCH = 20;        // LDS pitch (floats) of the [N][16] arrays
constexpr float LOG2E = 1.4426950408889634f;
template <int P>
__global__ __launch_bounds__(512) void tri_attn_bwd_core_kernel(
    float* __restrict__ dqkvg, const float* __restrict__ dog, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv, const float* __restrict__ wg,
    const float* __restrict__ bg, int b, int N, int npad, int H, int ending) {
    constexpr int C = 16, HC = 64, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem_tb[];
    float* Wl = smem_tb;                         // [64][P+4]: rows 0-15 k_h, 16-31 v_h, 32-47 q_h, 48-63 g_h
    float* Ql = Wl + 64 * (P + 4);               // [npad][17]   (scaled by 1/sqrt(c))
    float* Kl = Ql + npad * TB_PITCH;
    float* Vl = Kl + npad * TB_PITCH;
    // (the sigmoid gate of the row does not live in LDS: it is parked in the d(gate) slot of dqkvg -- which this workgroup
    // overwrites with the gradient afterwards -- so that rows up to 416 positions fit, BASELINE configs[3] draws N <= 384)
    float* Dl = Vl + npad * TB_PITCH;            // dog, then do = dog * gate
    float* Ml = Dl + npad * TB_PITCH;            // [npad] running max
    float* Ll = Ml + npad;                       // [npad] softmax denominator
    float* El = Ll + npad;                       // [npad] delta = do . o
    float* kml = El + npad;                      // [npad] key mask of this row (1 keep / 0 masked)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x, NWV = NT >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const float scale = 0.25f;                   // 1 / sqrt(head_dim)
    const long nrows = (long)b * N;
    const long nwork = nrows * H;
    int h_staged = -1;
    for (long w = blockIdx.x; w < nwork; w += gridDim.x) {
        // consecutive workgroups take the H heads of one row (same XCD neighbourhood is not needed for correctness)
        const int h = (int)(w % H);
        const long bu = w / H;
        const int bb = (int)(bu / N);
        const long u = bu - (long)bb * N;
        auto row_pos = [&](int v) -> long { return ending ? (((long)bb * N + v) * N + u) : (bu * N + v); };
        __syncthreads();                         // previous work item fully done with the LDS
        if (h != h_staged) {
            stage_weight_cll<P>(Wl, wk + (long)h * C * P, C, P, tid, NT);
            stage_weight_cll<P>(Wl + C * (P + 4), wv + (long)h * C * P, C, P, tid, NT);
            stage_weight_cll<P>(Wl + 2 * C * (P + 4), wq + (long)h * C * P, C, P, tid, NT, scale);
            stage_weight_cll<P>(Wl + 3 * C * (P + 4), wg + (long)h * C * P, C, P, tid, NT);
            h_staged = h;
            __syncthreads();
        }
        const float mu = mask[bu];
        // ---- projections of the row: 4 waves x 32-position blocks ----
        for (int vb = wave; vb * 32 < N; vb += NWV) {
            const int v = vb * 32 + r;
            const bool valid = v < N;
            float x[KH];
            load_row_cll<P>(pair + row_pos(valid ? v : 0) * P, hi, valid, x);
            ln_cll<KH>(x);
            f32x16 a0[1], a1[1];
            zero_acc(a0);
            zero_acc(a1);
            rowgemm<P, 1>(Wl, x, a0, r, hi);                         // [k; v]
            rowgemm<P, 1>(Wl + 2 * C * (P + 4), x, a1, r, hi);       // [q; g]
            // D rows of a 32-output block: channels {4hi+e} in registers 0-3, {8+4hi+e} in 4-7; second 16 outputs in 8-15.
            // Positions past N (the arrays are read in whole 16-position tiles) hold zeros: they meet p = 0 in the MFMAs.
            const float z = valid ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Kl[v * TB_PITCH + 4 * hi + e] = z * a0[0][e];
                Kl[v * TB_PITCH + 8 + 4 * hi + e] = z * a0[0][4 + e];
                Vl[v * TB_PITCH + 4 * hi + e] = z * a0[0][8 + e];
                Vl[v * TB_PITCH + 8 + 4 * hi + e] = z * a0[0][12 + e];
                Ql[v * TB_PITCH + 4 * hi + e] = z * a1[0][e];
                Ql[v * TB_PITCH + 8 + 4 * hi + e] = z * a1[0][4 + e];
            }
            if (valid) {
                float* gp = dqkvg + row_pos(v) * (4 * HC) + 3 * HC + h * C;
                *reinterpret_cast<float4*>(gp + 4 * hi) = make_float4(sigmoidf_(a1[0][8] + bg[h * C + 4 * hi]), sigmoidf_(a1[0][9] + bg[h * C + 4 * hi + 1]),
                                                                      sigmoidf_(a1[0][10] + bg[h * C + 4 * hi + 2]), sigmoidf_(a1[0][11] + bg[h * C + 4 * hi + 3]));
                *reinterpret_cast<float4*>(gp + 8 + 4 * hi) = make_float4(sigmoidf_(a1[0][12] + bg[h * C + 8 + 4 * hi]), sigmoidf_(a1[0][13] + bg[h * C + 8 + 4 * hi + 1]),
                                                                          sigmoidf_(a1[0][14] + bg[h * C + 8 + 4 * hi + 2]), sigmoidf_(a1[0][15] + bg[h * C + 8 + 4 * hi + 3]));
            }
            if (hi == 0) kml[v] = (valid && mu * mask[(long)bb * N + v] >= 0.5f) ? 1.f : 0.f;
        }
        for (int idx = tid; idx < npad * C; idx += NT) {             // dog of this head, row by row (zeros past N)
            const int v = idx >> 4, c = idx & 15;
            Dl[v * TB_PITCH + c] = v < N ? dog[row_pos(v) * HC + h * C + c] : 0.f;
        }
        __syncthreads();
        const int ql = lane & 15, g4 = lane >> 4;
        const int ntile = (N + 15) / 16;
        constexpr float FILL2 = -32768.0f * LOG2E;                   // the masked-key logit in the exp2 domain
        // ---- pass A: a 16-query tile per wave; lane (ql, g4) = query ql, keys / channels 4 g4 + e in registers ----
        for (int qt = wave; qt < ntile; qt += NWV) {
            const int q = qt * 16 + ql;
            const bool qok = q < N;
            const float4 qf = *reinterpret_cast<const float4*>(Ql + q * TB_PITCH + 4 * g4);
            float m_run = -1e30f, l_run = 0.f;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < npad; k0 += 32) {                  // sweep 1: statistics and o = P v, two 16-key tiles per update
                f32x4 sv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 kf = *reinterpret_cast<const float4*>(Kl + (k0 + 16 * j + ql) * TB_PITCH + 4 * g4);
                    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    z4 = mfma16(kf.x, qf.x, z4);
                    z4 = mfma16(kf.y, qf.y, z4);
                    z4 = mfma16(kf.z, qf.z, z4);
                    z4 = mfma16(kf.w, qf.w, z4);
                    sv[j] = z4;
                }
                float tmax = -INFINITY;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 km = *reinterpret_cast<const float4*>(kml + k0 + 16 * j + 4 * g4);
                    const float kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sv[j][e] = (k0 + 16 * j + 4 * g4 + e >= N) ? -INFINITY : (kmv[e] != 0.f ? sv[j][e] * LOG2E : FILL2);
                        tmax = fmaxf(tmax, sv[j][e]);
                    }
                }
                tmax = rows4_max(tmax);
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                float psum = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sv[j][e] = __builtin_amdgcn_exp2f(sv[j][e] - m_new);
                        psum += sv[j][e];
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= alpha;
                l_run = l_run * alpha + psum;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o = mfma16(Vl[(k0 + 16 * j + 4 * g4 + e) * TB_PITCH + ql], sv[j][e], o);      // o^T[ch][query]
            }
            const float il = 1.0f / rows4_sum(l_run);
            float* outp = dqkvg + row_pos(qok ? q : 0) * (4 * HC) + h * C + 4 * g4;
            // the gate parked there by the projection phase (another wave of this workgroup: ordered by the barrier above, and a
            // CU's L1 sees the CU's own stores)
            const float4 gv = qok ? *reinterpret_cast<const float4*>(outp + 3 * HC) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 dg = *reinterpret_cast<const float4*>(Dl + q * TB_PITCH + 4 * g4);
            const float gvv[4] = {gv.x, gv.y, gv.z, gv.w}, dgv[4] = {dg.x, dg.y, dg.z, dg.w};
            float dov[4], dgp[4], dsum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float oe = o[e] * il;
                dgp[e] = dgv[e] * oe * gvv[e] * (1.0f - gvv[e]);      // d(gate pre-activation)
                dov[e] = dgv[e] * gvv[e];                             // do = dog * gate
                dsum += dov[e] * oe;
            }
            const float delta = rows4_sum(dsum);
            if (qok) *reinterpret_cast<float4*>(outp + 3 * HC) = make_float4(dgp[0], dgp[1], dgp[2], dgp[3]);
            *reinterpret_cast<float4*>(Dl + q * TB_PITCH + 4 * g4) = make_float4(dov[0], dov[1], dov[2], dov[3]);   // pass B reads do
            if (g4 == 0) { Ml[q] = m_run; Ll[q] = il; El[q] = delta; }
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < npad; k0 += 32) {                  // sweep 2: dS and dq
                f32x4 sv[2], dp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 kf = *reinterpret_cast<const float4*>(Kl + (k0 + 16 * j + ql) * TB_PITCH + 4 * g4);
                    const float4 vf = *reinterpret_cast<const float4*>(Vl + (k0 + 16 * j + ql) * TB_PITCH + 4 * g4);
                    f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
                    z4 = mfma16(kf.x, qf.x, z4);
                    d4 = mfma16(vf.x, dov[0], d4);                    // dP[key][query] = v_key . do_query
                    z4 = mfma16(kf.y, qf.y, z4);
                    d4 = mfma16(vf.y, dov[1], d4);
                    z4 = mfma16(kf.z, qf.z, z4);
                    d4 = mfma16(vf.z, dov[2], d4);
                    z4 = mfma16(kf.w, qf.w, z4);
                    d4 = mfma16(vf.w, dov[3], d4);
                    sv[j] = z4;
                    dp[j] = d4;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 km = *reinterpret_cast<const float4*>(kml + k0 + 16 * j + 4 * g4);
                    const float kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool keep = kmv[e] != 0.f;              // keys past N have kml = 0
                        const float pe = __builtin_amdgcn_exp2f(sv[j][e] * LOG2E - m_run) * il;
                        sv[j][e] = keep ? pe * (dp[j][e] - delta) : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq = mfma16(Kl[(k0 + 16 * j + 4 * g4 + e) * TB_PITCH + ql], sv[j][e], dq);    // dq^T[ch][query]
            }
            if (qok) *reinterpret_cast<float4*>(outp) = make_float4(scale * dq[0], scale * dq[1], scale * dq[2], scale * dq[3]);   // q = scale * W_q x
        }
        __syncthreads();
        // ---- pass B: a 16-key tile per wave; lane (ql, g4) = key ql, queries / channels 4 g4 + e in registers ----
        for (int kt = wave; kt < ntile; kt += NWV) {
            const int j = kt * 16 + ql;
            const bool jok = j < N;
            const float4 kf = *reinterpret_cast<const float4*>(Kl + j * TB_PITCH + 4 * g4);
            const float4 vf = *reinterpret_cast<const float4*>(Vl + j * TB_PITCH + 4 * g4);
            const bool keep = kml[j] != 0.f;
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < npad; q0 += 32) {                  // two 16-query tiles per step (independent MFMA chains)
                f32x4 sv[2], ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 qa = *reinterpret_cast<const float4*>(Ql + (q0 + 16 * t + ql) * TB_PITCH + 4 * g4);
                    const float4 da = *reinterpret_cast<const float4*>(Dl + (q0 + 16 * t + ql) * TB_PITCH + 4 * g4);
                    f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
                    z4 = mfma16(qa.x, kf.x, z4);                      // S[query][key]
                    d4 = mfma16(da.x, vf.x, d4);                      // dP[query][key]
                    z4 = mfma16(qa.y, kf.y, z4);
                    d4 = mfma16(da.y, vf.y, d4);
                    z4 = mfma16(qa.z, kf.z, z4);
                    d4 = mfma16(da.z, vf.z, d4);
                    z4 = mfma16(qa.w, kf.w, z4);
                    d4 = mfma16(da.w, vf.w, d4);
                    sv[t] = z4;
                    ds[t] = d4;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 m4 = *reinterpret_cast<const float4*>(Ml + q0 + 16 * t + 4 * g4);
                    const float4 l4 = *reinterpret_cast<const float4*>(Ll + q0 + 16 * t + 4 * g4);
                    const float4 e4 = *reinterpret_cast<const float4*>(El + q0 + 16 * t + 4 * g4);
                    const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool qin = q0 + 16 * t + 4 * g4 + e < N;         // the statistics of queries past N are not defined
                        const float pe = qin ? __builtin_amdgcn_exp2f((keep ? sv[t][e] * LOG2E : FILL2) - mv[e]) * lv[e] : 0.f;
                        sv[t][e] = pe;
                        ds[t][e] = (keep && qin) ? pe * (ds[t][e] - ev[e]) : 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv = mfma16(Dl[(q0 + 16 * t + 4 * g4 + e) * TB_PITCH + ql], sv[t][e], dv);      // dv^T[ch][key] += do[query][ch] p
                        dk = mfma16(Ql[(q0 + 16 * t + 4 * g4 + e) * TB_PITCH + ql], ds[t][e], dk);      // dk^T[ch][key] += q[query][ch] dS
                    }
            }
            if (jok) {
                float* outp = dqkvg + row_pos(j) * (4 * HC) + h * C + 4 * g4;
                *reinterpret_cast<float4*>(outp + HC) = make_float4(dk[0], dk[1], dk[2], dk[3]);
                *reinterpret_cast<float4*>(outp + 2 * HC) = make_float4(dv[0], dv[1], dv[2], dv[3]);
            }
        }
    }
}

// d/dx of LayerNorm (no affine) over the last axis: one wave per row, C a multiple of 64
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ res, long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    const float* gr = dy + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / C + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float y = (xr[c] - mean) * rstd; s1 += gr[c]; s2 += gr[c] * y; }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 / C, m2 = s2 / C;
    for (int c = lane; c < C; c += 64) {
        const float y = (xr[c] - mean) * rstd;
        dx[row * C + c] = rstd * (gr[c] - m1 - y * m2) + (res ? res[row * C + c] : 0.f);
    }
}


// The same for rows of 64 channels (the pair track): 16 lanes x float4 per row, four rows per wave, reductions inside a DPP row --
// the row kernel above moves 4 bytes per lane and load and spends six LDS-crossbar shuffles per reduction (74 us for the 157 MB of
// a b = 2, N = 320 call; this form runs at the speed of the three streams).  res (may be null): added to the result (the
// residual path of the update whose backward this is).
__global__ __launch_bounds__(256) void ln_rows_bwd64_kernel(float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res, long rows) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    const long off = row * 64 + 4 * (threadIdx.x & 15);
    const float4 xv = *reinterpret_cast<const float4*>(x + off), gv = *reinterpret_cast<const float4*>(dy + off);
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = *reinterpret_cast<const float4*>(res + off);
    const float mean = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64);
    const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
    const float rstd = 1.0f / sqrtf(row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 64) + 1e-5f);
    const float y0 = d0 * rstd, y1 = d1 * rstd, y2 = d2 * rstd, y3 = d3 * rstd;
    const float m1 = row16_sum((gv.x + gv.y) + (gv.z + gv.w)) * (1.0f / 64);
    const float m2 = row16_sum((gv.x * y0 + gv.y * y1) + (gv.z * y2 + gv.w * y3)) * (1.0f / 64);
    *reinterpret_cast<float4*>(dx + off) = make_float4(rstd * (gv.x - m1 - y0 * m2) + rv.x, rstd * (gv.y - m1 - y1 * m2) + rv.y,
                                                       rstd * (gv.z - m1 - y2 * m2) + rv.z, rstd * (gv.w - m1 - y3 * m2) + rv.w);
}

// Backward of an attention-bias head over the pair rows (modules.py:300-304, AF2_modules.py:454-459): bias[b, h, i, j] = sum_c W'[h][c]
// LN(pair[b, i, j])[c] (+ c_h).  ONE pass over the pair rows instead of a permute copy of dbias, a K = H GEMM, a LayerNorm-backward
// pass and a LayerNorm pass for the weight gradient's operand:  dLN = sum_h dbias[b, h, p] W'[h][:]  (H <= 8 FMAs per channel, dbias read
// in its own [b, H, N N] layout),  dx = LN'(dLN; x),  and on the side xn = LN(x) and d2[row][h] = dbias by position -- the two
// operands of the weight-gradient reduction (prd_linear_wgrad).  Thread layout of ln_rows_bwd64_kernel: 16 lanes x float4 per row.
template <int H>
__global__ __launch_bounds__(256) void pair_bias_bwd64_kernel(float* __restrict__ dx, float* __restrict__ xn, float* __restrict__ d2,
                                                              const float* __restrict__ dbias, const float* __restrict__ wf,
                                                              const float* __restrict__ x, long rows, long nn) {
    const int l16 = threadIdx.x & 15;
    float4 w4[H];
#pragma unroll
    for (int h = 0; h < H; ++h) w4[h] = *reinterpret_cast<const float4*>(wf + h * 64 + 4 * l16);
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    const long bb = row / nn, p = row - bb * nn;
    const float* dp = dbias + bb * H * nn + p;
    float d[H];
#pragma unroll
    for (int h = 0; h < H; ++h) d[h] = dp[(long)h * nn];
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < H; ++h) { gv.x += d[h] * w4[h].x; gv.y += d[h] * w4[h].y; gv.z += d[h] * w4[h].z; gv.w += d[h] * w4[h].w; }
    const long off = row * 64 + 4 * l16;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    const float mean = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64);
    const float d0 = xv.x - mean, d1 = xv.y - mean, d2_ = xv.z - mean, d3 = xv.w - mean;
    const float rstd = 1.0f / sqrtf(row16_sum((d0 * d0 + d1 * d1) + (d2_ * d2_ + d3 * d3)) * (1.0f / 64) + 1e-5f);
    const float y0 = d0 * rstd, y1 = d1 * rstd, y2 = d2_ * rstd, y3 = d3 * rstd;
    const float m1 = row16_sum((gv.x + gv.y) + (gv.z + gv.w)) * (1.0f / 64);
    const float m2 = row16_sum((gv.x * y0 + gv.y * y1) + (gv.z * y2 + gv.w * y3)) * (1.0f / 64);
    *reinterpret_cast<float4*>(dx + off) = make_float4(rstd * (gv.x - m1 - y0 * m2), rstd * (gv.y - m1 - y1 * m2),
                                                       rstd * (gv.z - m1 - y2 * m2), rstd * (gv.w - m1 - y3 * m2));
    if (xn) *reinterpret_cast<float4*>(xn + off) = make_float4(y0, y1, y2, y3);
    if (d2 && l16 < H) {
        float v = d[0];
#pragma unroll
        for (int h = 1; h < H; ++h) v = l16 == h ? d[h] : v;
        d2[row * H + l16] = v;
    }
}

// ---- weight gradient of a pair-position linear: dW[o][i] = sum_rows dy[row][o] * x[row][i] --------------------------------
// rows = b N N (2e5 at N = 320), O, I <= 256: a reduction over a huge K with a tiny output, for which the BLAS picks an
// 8-workgroup kernel (450 us per call, 35 % of the first-cut training step).  Here the rows are dealt to ~256 slabs; a
// workgroup holds 64x64 blocks of dW in fp32 MFMA accumulators (v_mfma_f32_32x32x2_f32: two rows per instruction, lane r
// supplies dy[row][o0 + 2r + ea] and x[row][i0 + 2r + eb], so a wave reads 256 contiguous bytes per row and operand), waves
// that share a block interleave the rows and are merged in LDS; the slab partials are summed by a second kernel in slab
// order (deterministic, no atomics).
template <int NBLK>                                     // 64x64 blocks of dW per workgroup: 1, 2 or 4 (4 / NBLK waves per block)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(float* __restrict__ part, const float* __restrict__ dy, const float* __restrict__ x,
                                                           long rows, int O, int I, int lddy, int ldx, int rows_per_wg, int want_db) {
    constexpr int WPB = 4 / NBLK, U = 8;
    __shared__ float red[WPB > 1 ? 4 : 1][64][64];
    __shared__ float redb[4][2][32];                        // column sums of dy (the bias gradient) of the waves of a block
    const int pn = O * I + (want_db ? O : 0);               // floats per slab partial: dW, then db
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int ibn = I / 64;
    const int blk = blockIdx.y * NBLK + wave / WPB, sub = wave % WPB;
    const int ob = blk / ibn, ib = blk - ob * ibn;
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    const float* dyp = dy + ob * 64 + 2 * r;
    const float* xp = x + ib * 64 + 2 * r;
    f32x16 acc[2][2];
#pragma unroll
    for (int ea = 0; ea < 2; ++ea)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ea][eb][q] = 0.f;
    float sdb0 = 0.f, sdb1 = 0.f;                           // db partial of columns ob * 64 + 2 r, + 1 (rows of this lane's parity)
    float2 ca[U], cb[U], na[U], nb[U];
    auto load = [&](float2 (&a)[U], float2 (&b)[U], long row) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = row + 2 * u + hi;
            const long rc = rr < r1 ? rr : r1 - 1;          // clamped (rows past the slab are multiplied by zero)
            const float z = rr < r1 ? 1.f : 0.f;
            const float2 t = *reinterpret_cast<const float2*>(dyp + rc * lddy);
            a[u] = make_float2(t.x * z, t.y * z);
            b[u] = *reinterpret_cast<const float2*>(xp + rc * ldx);
        }
    };
    const long step = 2L * U * WPB;
    long row = r0 + 2L * U * sub;
    if (row < r1) load(ca, cb, row);
    for (; row < r1; row += step) {
        const long nx = row + step;
        load(na, nb, nx < r1 ? nx : row);                   // unconditional prefetch (the last one re-reads the current group)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0][0] = mfma32(ca[u].x, cb[u].x, acc[0][0]);
            acc[0][1] = mfma32(ca[u].x, cb[u].y, acc[0][1]);
            acc[1][0] = mfma32(ca[u].y, cb[u].x, acc[1][0]);
            acc[1][1] = mfma32(ca[u].y, cb[u].y, acc[1][1]);
            sdb0 += ca[u].x;
            sdb1 += ca[u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
    }
    if (WPB > 1) {                                          // merge the waves of a block (fixed order)
        if (sub > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[wave][e * 16 + q][lane] = acc[e >> 1][e & 1][q];
        }
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int w = 1; w < WPB; ++w)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[e >> 1][e & 1][q] += red[wave + w][e * 16 + q][lane];
        }
    }
    if (want_db && ib == 0) {                               // block-uniform per wave; the waves of a block add up in wave order
        sdb0 += __shfl_xor(sdb0, 32);
        sdb1 += __shfl_xor(sdb1, 32);
        if (hi == 0) { redb[wave][0][r] = sdb0; redb[wave][1][r] = sdb1; }
    }
    if (want_db) __syncthreads();
    if (want_db && ib == 0 && sub == 0 && hi == 0) {
        float b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { b0 += redb[wave + w][0][r]; b1 += redb[wave + w][1][r]; }
        *reinterpret_cast<float2*>(part + (size_t)blockIdx.x * pn + (size_t)O * I + ob * 64 + 2 * r) = make_float2(b0, b1);
    }
    if (sub == 0) {
        float* pt = part + (size_t)blockIdx.x * pn;
#pragma unroll
        for (int ea = 0; ea < 2; ++ea)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = ob * 64 + 2 * drow32(q, hi) + ea;
                *reinterpret_cast<float2*>(pt + (size_t)o * I + ib * 64 + 2 * r) = make_float2(acc[ea][0][q], acc[ea][1][q]);
            }
    }
}

// The same slab reduction on the 16-bit matrix pipe (gemm mode 1): v_mfma_f32_32x32x16_f16 takes SIXTEEN rows per instruction
// where the fp32 form takes two, and the three split products hi*hi, hi*lo, lo*hi still leave a 64x64 block at 12 matrix
// instructions per 16 rows instead of 32 twice as long ones.  Lane (r, hi) loads dy[row0 + 8 hi + u][o0 + 2r + ea] and
// x[row0 + 8 hi + u][i0 + 2r + eb], u < 8: its eight values of one column ARE the eight contraction entries an operand lane
// holds, so the rows go from the float2 loads through the hi | lo split straight into the MFMA, no LDS.
// Range: x (activations) is O(1); dy is a gradient of unknown magnitude and fp16 has five exponent bits, so dy is multiplied by
// a wave-uniform power of two `cur` before the split and the accumulators carry that factor.  `cur` is set from the first
// non-zero 16-row group (largest |dy| -> [64, 128)) and lowered, with the accumulators rescaled, whenever a group would
// pass 2^14: groups far below the running scale lose relative precision exactly in proportion to how little they contribute.
template <int NBLK>
__global__ __launch_bounds__(256) void linear_wgrad_h2_kernel(float* __restrict__ part, const float* __restrict__ dy, const float* __restrict__ x,
                                                              long rows, int O, int I, int lddy, int ldx, int rows_per_wg, int want_db) {
    constexpr int WPB = 4 / NBLK, U = 8;
    __shared__ float red[WPB > 1 ? 4 : 1][64][64];
    __shared__ float redb[4][2][32];
    const int pn = O * I + (want_db ? O : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
    const int ibn = I / 64;
    const int blk = blockIdx.y * NBLK + wave / WPB, sub = wave % WPB;
    const int ob = blk / ibn, ib = blk - ob * ibn;
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    const float* dyp = dy + ob * 64 + 2 * r;
    const float* xp = x + ib * 64 + 2 * r;
    f32x16 acc[2][2];
#pragma unroll
    for (int ea = 0; ea < 2; ++ea)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ea][eb][q] = 0.f;
    float sdb0 = 0.f, sdb1 = 0.f;
    float cur = 0.f;                                        // the power of two dy is multiplied by (0: not set yet)
    float2 ca[U], cb[U], na[U], nb[U];
    auto load = [&](float2 (&a)[U], float2 (&b)[U], long row) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = row + 8 * hi + u;
            const long rc = rr < r1 ? rr : r1 - 1;          // clamped (rows past the slab are multiplied by zero)
            const float z = rr < r1 ? 1.f : 0.f;
            const float2 t = *reinterpret_cast<const float2*>(dyp + rc * lddy);
            a[u] = make_float2(t.x * z, t.y * z);
            b[u] = *reinterpret_cast<const float2*>(xp + rc * ldx);
        }
    };
    auto mfma_h = [](u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    };
    const long step = 2L * U * WPB;
    long row = r0 + 2L * U * sub;
    if (row < r1) load(ca, cb, row);
    for (; row < r1; row += step) {
        const long nx = row + step;
        load(na, nb, nx < r1 ? nx : row);                   // unconditional prefetch (the last one re-reads the current group)
        float mx = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(ca[u].x), __builtin_fabsf(ca[u].y)));
            sdb0 += ca[u].x;
            sdb1 += ca[u].y;
        }
        // (the test is per lane, one ballot; the wave maximum is only formed when some lane asks for a new scale)
        if (__any(mx > 0.f && mx < 3.0e38f && (cur == 0.f || mx * cur > 16384.f))) {      // rare after the first group
            mx = mx < 3.0e38f ? mx : 0.f;
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o_));
            mx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mx)));
            int e = 7 - __builtin_amdgcn_frexp_expf(mx);    // mx * 2^e in [64, 128)
            e = e > 120 ? 120 : (e < -120 ? -120 : e);
            const float nw = __builtin_ldexpf(1.0f, e);
            if (cur != 0.f) {
                const float f = nw / cur;                   // a power of two: exact
#pragma unroll
                for (int ea = 0; ea < 2; ++ea)
#pragma unroll
                    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[ea][eb][q] *= f;
            }
            cur = nw;
        }
        u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned h, l;
            split2h(ca[2 * w].x * cur, ca[2 * w + 1].x * cur, h, l); ah[0][w] = h; al[0][w] = l;
            split2h(ca[2 * w].y * cur, ca[2 * w + 1].y * cur, h, l); ah[1][w] = h; al[1][w] = l;
            split2h(cb[2 * w].x, cb[2 * w + 1].x, h, l); bh[0][w] = h; bl[0][w] = l;
            split2h(cb[2 * w].y, cb[2 * w + 1].y, h, l); bh[1][w] = h; bl[1][w] = l;
        }
#pragma unroll
        for (int ea = 0; ea < 2; ++ea)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                acc[ea][eb] = mfma_h(ah[ea], bh[eb], acc[ea][eb]);
                acc[ea][eb] = mfma_h(ah[ea], bl[eb], acc[ea][eb]);
                acc[ea][eb] = mfma_h(al[ea], bh[eb], acc[ea][eb]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
    }
    {   // back to the scale of dy
        const float inv = cur != 0.f ? 1.0f / cur : 0.f;
#pragma unroll
        for (int ea = 0; ea < 2; ++ea)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ea][eb][q] *= inv;
    }
    if (WPB > 1) {                                          // merge the waves of a block (fixed order)
        if (sub > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[wave][e * 16 + q][lane] = acc[e >> 1][e & 1][q];
        }
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int w = 1; w < WPB; ++w)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[e >> 1][e & 1][q] += red[wave + w][e * 16 + q][lane];
        }
    }
    if (want_db && ib == 0) {
        sdb0 += __shfl_xor(sdb0, 32);
        sdb1 += __shfl_xor(sdb1, 32);
        if (hi == 0) { redb[wave][0][r] = sdb0; redb[wave][1][r] = sdb1; }
    }
    if (want_db) __syncthreads();
    if (want_db && ib == 0 && sub == 0 && hi == 0) {
        float b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { b0 += redb[wave + w][0][r]; b1 += redb[wave + w][1][r]; }
        *reinterpret_cast<float2*>(part + (size_t)blockIdx.x * pn + (size_t)O * I + ob * 64 + 2 * r) = make_float2(b0, b1);
    }
    if (sub == 0) {
        float* pt = part + (size_t)blockIdx.x * pn;
#pragma unroll
        for (int ea = 0; ea < 2; ++ea)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = ob * 64 + 2 * drow32(q, hi) + ea;
                *reinterpret_cast<float2*>(pt + (size_t)o * I + ib * 64 + 2 * r) = make_float2(acc[ea][0][q], acc[ea][1][q]);
            }
    }
}

// The same reduction for a NARROW output side (O <= 16: the attention-bias and coordinate-head linears, modules.py:300-304,
// model.py:364-369): one lane per input column, the O values of dy per row broadcast, rows dealt to slabs and to the four waves.
template <int OMAX>
__global__ __launch_bounds__(256) void linear_wgrad_narrow_kernel(float* __restrict__ part, const float* __restrict__ dy, const float* __restrict__ x,
                                                                  long rows, int O, int I, int lddy, int ldx, int rows_per_wg, int want_db) {
    __shared__ float red[4][OMAX][64];
    __shared__ float redb[4][OMAX];
    const int pn = O * I + (want_db ? O : 0);
    float sdb[OMAX];
#pragma unroll
    for (int o = 0; o < OMAX; ++o) sdb[o] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.y * 64 + lane;
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float acc[OMAX];
#pragma unroll
    for (int o = 0; o < OMAX; ++o) acc[o] = 0.f;
    constexpr int U = OMAX <= 4 ? 32 : 8;                  // rows in flight per wave: 256 B each -- the narrow form is a pure stream of x
    const bool vec4 = O == 4 && lddy == 4 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    for (long row = r0 + (long)U * wave; row < r1; row += 4 * U) {
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                           // unconditional loads from clamped rows (all in flight); rows past the slab x 0
            const long rc = row + u < r1 ? row + u : r1 - 1;
            xv[u] = x[rc * ldx + i] * (row + u < r1 ? 1.f : 0.f);
        }
        if (OMAX == 4 && vec4) {                                // the four values of a row in one 16-byte broadcast load
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 d4 = *reinterpret_cast<const float4*>(dy + (row + u < r1 ? row + u : r1 - 1) * 4);
                const float z = row + u < r1 ? 1.f : 0.f;
                const float dv[4] = {d4.x * z, d4.y * z, d4.z * z, d4.w * z};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    acc[o] += dv[o] * xv[u];
                    sdb[o] += dv[o];
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* dr = dy + (row + u < r1 ? row + u : r1 - 1) * lddy;         // wave-uniform address: a broadcast load
            const float z = row + u < r1 ? 1.f : 0.f;
#pragma unroll
            for (int o = 0; o < OMAX; ++o)
                if (o < O) {
                    const float dv = dr[o] * z;
                    acc[o] += dv * xv[u];
                    sdb[o] += dv;                                // the same value in every lane
                }
        }
    }
    if (want_db && blockIdx.y == 0) {
        if (lane == 0)
#pragma unroll
            for (int o = 0; o < OMAX; ++o) redb[wave][o] = sdb[o];
    }
#pragma unroll
    for (int o = 0; o < OMAX; ++o) red[wave][o][lane] = acc[o];
    __syncthreads();
    if (wave == 0) {
        float* pt = part + (size_t)blockIdx.x * pn;
#pragma unroll
        for (int o = 0; o < OMAX; ++o)
            if (o < O) pt[(size_t)o * I + i] = ((red[0][o][lane] + red[1][o][lane]) + red[2][o][lane]) + red[3][o][lane];
        if (want_db && blockIdx.y == 0 && lane < O) pt[(size_t)O * I + lane] = ((redb[0][lane] + redb[1][lane]) + redb[2][lane]) + redb[3][lane];
    }
}

// Gradient of a small embedding table looked up at every pair position (bond-type / bond-distance / relative-position tables of
// the input stage, modules.py:35-71): dtable[c][:] = sum of dy over the rows with idx == c.  The BLAS form (one-hot^T dy) is the
// same long-K, tiny-output GEMM as the linear weight gradients (475 us per table); here a wave keeps a private [card][C] table in
// LDS (lane = channel, one row at a time: no conflicts, fixed order), the four waves and then the slabs are summed in order.
__global__ __launch_bounds__(256) void embed_wgrad_kernel(float* __restrict__ part, const long long* __restrict__ idx, const float* __restrict__ dy,
                                                          const float* __restrict__ row_scale, long rows, int card, int C, int lddy, int rows_per_wg) {
    extern __shared__ float etab[];                          // [4 waves][card][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = etab + (size_t)wave * card * 64;
    for (int e = lane; e < card * 64; e += 64) mine[e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    constexpr int U = 8;
    for (long row = r0 + (long)U * wave; row < r1; row += 4 * U) {
        float v[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = row + u < r1 ? row + u : r1 - 1;
            c[u] = (int)idx[rr];                             // wave-uniform
            v[u] = (row + u < r1 && lane < C) ? dy[rr * lddy + lane] * (row_scale ? row_scale[rr] : 1.0f) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c[u] >= 0 && c[u] < card) mine[c[u] * 64 + lane] += v[u];
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * card * C;
    for (int e = threadIdx.x; e < card * 64; e += 256) {
        const int cc = e >> 6, ch = e & 63;
        if (ch < C) pt[cc * C + ch] = ((etab[e] + etab[(size_t)card * 64 + e]) + etab[(size_t)2 * card * 64 + e]) + etab[(size_t)3 * card * 64 + e];
    }
}

// Several small tables looked up at the same pair positions (the input stage: three bond-feature tables, bond distance,
// relative position): their gradients in ONE pass over dy -- a wave keeps one private [total rows][64] table in LDS, every row of
// dy is loaded once and added, times the set's row scale, at the K table rows its K indices name.
struct EmbedMulti { const long long* idx[8]; const float* scale[8]; int off[8]; int card[8]; int K; int total; };
__global__ __launch_bounds__(256) void embed_wgrad_multi_kernel(float* __restrict__ part, EmbedMulti em, const float* __restrict__ dy,
                                                                long rows, int C, int lddy, int rows_per_wg) {
    extern __shared__ float etabm[];                         // [4 waves][total][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = etabm + (size_t)wave * em.total * 64;
    for (int e = lane; e < em.total * 64; e += 64) mine[e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    // 64 rows per wave and round: the lanes fetch the K indices and scales of 64 rows with coalesced loads (one row per lane) and
    // hand them out with v_readlane -- a wave-uniform 8-byte load per (row, table) made the single-table form latency-bound
    const int lc = lane < C ? lane : 0;
    const float lz = lane < C ? 1.f : 0.f;
    for (long row0 = r0 + 64L * wave; row0 < r1; row0 += 256) {
        const long rl = row0 + lane;
        const bool ok = rl < r1;
        const long rcl = ok ? rl : r1 - 1;
        int ci[8];
        float si[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ci[k] = -1;
            si[k] = 0.f;
            if (k < em.K) {
                ci[k] = ok ? (int)em.idx[k][rcl] : -1;
                si[k] = em.scale[k] ? em.scale[k][rcl] : 1.0f;
            }
        }
        for (int u0 = 0; u0 < 64 && row0 + u0 < r1; u0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long rr = row0 + u0 + j;
                v[j] = dy[(rr < r1 ? rr : r1 - 1) * lddy + lc] * (rr < r1 ? lz : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < em.K) {
                        const int c = __builtin_amdgcn_readlane(ci[k], u0 + j);
                        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, si[k]), u0 + j));
                        if (c >= 0 && c < em.card[k]) mine[(em.off[k] + c) * 64 + lane] += v[j] * sc;
                    }
        }
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * em.total * C;
    for (int e = threadIdx.x; e < em.total * 64; e += 256) {
        const int cc = e >> 6, ch = e & 63;
        if (ch < C) pt[cc * C + ch] = ((etabm[e] + etabm[(size_t)em.total * 64 + e]) + etabm[(size_t)2 * em.total * 64 + e]) + etabm[(size_t)3 * em.total * 64 + e];
    }
}

// 64 elements per workgroup, the slabs dealt in four contiguous quarters to the four waves (eight loads in flight per lane),
// quarter sums combined in wave order: the summation order is fixed.
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(float* __restrict__ dw, float* __restrict__ db, const float* __restrict__ part,
                                                                  int n, int nw, int slabs) {     // n floats per slab: nw of dW, then db
    __shared__ float qs[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int per = (slabs + 3) / 4, k0 = wave * per, k1 = (k0 + per < slabs) ? k0 + per : slabs;
    float s = 0.f;
    if (e < n) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(k + u) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; k < k1; ++k) s += part[(size_t)k * n + e];
    }
    qs[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && e < n) {
        const float v = ((qs[0][lane] + qs[1][lane]) + qs[2][lane]) + qs[3][lane];
        if (e < nw) dw[e] = v;
        else db[e - nw] = v;
    }
}

}  // namespace

#define PRD_BWD_SET_LDS(kernel)                                                                                 \
    do {                                                                                                        \
        static std::once_flag once_;                                                                            \
        std::call_once(once_, [] {                                                                              \
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                     \
    } while (0)

// Operands of the two gradient contractions of TriangleMultiplication as ONE stacked contraction (prd_tri_mul_contract with 2P
// channel pairs): ops [b][4P][N][ldn] = dO | dO^T | B^T | A^T by channel block, so that
//   out[c]     = dO[c]   (B^T[c])^T  = dA[c]      (dA[i][k] = sum_j dO[i][j] B[j][k])
//   out[P + c] = dO^T[c] (A^T[c])^T  = dB[c]      (dB[j][k] = sum_i dO[i][j] A[i][k])
// lands directly in the channel-major dAB the projection-stage backward reads.  dO is already in block 0 (written there by
// prd_tri_mul_out_bwd); this kernel writes the three transposes through 64 x 64 LDS tiles (coalesced on both sides) and zeroes
// the padding columns N .. ldn of every block, dO's included.
__global__ __launch_bounds__(256) void tri_mul_bwd_operands_kernel(float* __restrict__ ops, const float* __restrict__ AB,
                                                                   int N, int ldn, int P, int T, long ntask) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x, cl = 4 * (t & 15), rl = t >> 4;
    for (long task = blockIdx.x; task < ntask; task += gridDim.x) {
        long q = task;
        const int src = (int)(q % 3); q /= 3;
        const int tj = (int)(q % T); q /= T;
        const int ti = (int)(q % T); q /= T;
        const int c = (int)(q % P);
        const int bb = (int)(q / P);
        const size_t plane = (size_t)N * ldn;
        const float* S = src == 0 ? ops + ((size_t)bb * 4 * P + c) * plane
                       : AB + ((size_t)bb * 2 * P + (src == 1 ? P : 0) + c) * plane;
        float* D = ops + ((size_t)bb * 4 * P + (size_t)(src + 1) * P + c) * plane;
        const int i0 = 64 * ti, j0 = 64 * tj;
        __syncthreads();                                // the previous task's tile has been read
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = i0 + rl + 16 * k, col = j0 + cl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < N) {
                const float* sp = S + (size_t)row * ldn + col;
                if (col + 3 < N) v = *reinterpret_cast<const float4*>(sp);
                else {
                    if (col < N) v.x = sp[0];
                    if (col + 1 < N) v.y = sp[1];
                    if (col + 2 < N) v.z = sp[2];
                }
                if (src == 0 && col + 3 >= N && col < ldn) {      // dO's own padding columns
                    float* zp = ops + ((size_t)bb * 4 * P + c) * plane + (size_t)row * ldn + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e >= N && col + e < ldn) zp[e] = 0.f;
                }
            }
            tile[rl + 16 * k][cl] = v.x;
            tile[rl + 16 * k][cl + 1] = v.y;
            tile[rl + 16 * k][cl + 2] = v.z;
            tile[rl + 16 * k][cl + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = j0 + rl + 16 * k, col = i0 + cl;    // D[row][col] = S[col][row]
            if (row < N && col < ldn) {
                const float4 v = make_float4(tile[cl][rl + 16 * k], tile[cl + 1][rl + 16 * k], tile[cl + 2][rl + 16 * k], tile[cl + 3][rl + 16 * k]);
                *reinterpret_cast<float4*>(D + (size_t)row * ldn + col) = v;
            }
        }
    }
}

// out[b][i][p][j] = dy[b][i][j][p] + dy[b][j][i][p]: the symmetrised gradient of the outer-linear update (modules.py:283-287:
// out[i][j] depends on x_i x_j and on u_i - u_j) in the [b, N P, N] row layout its backward GEMM contracts over j.  One workgroup
// per (b, i, 64 positions j): both reads are 4 P-byte rows, the transpose goes through a 64 x 65 LDS tile.
__global__ __launch_bounds__(256) void sym_transpose_kernel(float* __restrict__ out, const float* __restrict__ dy, int N, int P, int T, long ntask) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x, c4 = 4 * (t & 15), rl = t >> 4;
    for (long task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int tj = (int)(task % T);
        const long bi = task / T;                       // b * N + i
        const long bb = bi / N;
        const int i = (int)(bi - bb * N), j0 = 64 * tj;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl = rl + 16 * k, j = j0 + jl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < N && c4 < P) {
                const float4 a = *reinterpret_cast<const float4*>(dy + (bi * N + j) * P + c4);
                const float4 c = *reinterpret_cast<const float4*>(dy + ((bb * N + j) * N + i) * P + c4);
                v = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
            }
            tile[jl][c4] = v.x; tile[jl][c4 + 1] = v.y; tile[jl][c4 + 2] = v.z; tile[jl][c4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pl = rl + 16 * k, j = j0 + c4;    // out[(bi P + pl) N + j .. j + 3]
            if (pl < P && j < N) {
                float* dst = out + (bi * P + pl) * N + j;
                const float v[4] = {tile[c4][pl], tile[c4 + 1][pl], tile[c4 + 2][pl], tile[c4 + 3][pl]};
                if (j + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j + e < N) dst[e] = v[e];
            }
        }
    }
}

// rbf[b][i][j][r] = mask_i mask_j exp(-scale (|z_i - z_j| - center_r)^2), scale = (R - 1) / 2: the radial-basis features of
// the pair distances (modules.py:73-82, model.py:352-356) as rows, for the weight gradient of the distance embedding
// (prd_pair_init generates them on the fly and never stores them).  One thread per (pair position, four centres).
__global__ __launch_bounds__(256) void rbf_rows_kernel(float* __restrict__ out, const float* __restrict__ z, const float* __restrict__ centers,
                                                       const float* __restrict__ mask, long npos, int N, int R) {
    const int r4 = R / 4;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= npos * r4) return;
    const long pos = tid / r4;
    const int q = (int)(tid - pos * r4);
    const long bi = pos / N;
    const int j = (int)(pos - bi * N);
    const long bb = bi / N;
    const float* zi = z + bi * 3;
    const float* zj = z + (bb * N + j) * 3;
    const float dx = zi[0] - zj[0], dy = zi[1] - zj[1], dz = zi[2] - zj[2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const float m = mask[bi] * mask[bb * N + j];
    const float scale = (R - 1) * 0.5f;
    const float4 c = *reinterpret_cast<const float4*>(centers + 4 * q);
    const float a0 = dist - c.x, a1 = dist - c.y, a2 = dist - c.z, a3 = dist - c.w;
    *reinterpret_cast<float4*>(out + pos * R + 4 * q) =
        make_float4(m * expf(-scale * a0 * a0), m * expf(-scale * a1 * a1), m * expf(-scale * a2 * a2), m * expf(-scale * a3 * a3));
}

extern "C" int prd_rbf_rows(float* out, const float* z, const float* centers, const float* mask, int b, int N, int R, hipStream_t stream) {
    if (!out || !z || !centers || !mask || b <= 0 || N <= 0 || R <= 0) return PRD_ERR_ARG;
    if (R % 4) return PRD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(centers)) & 15) return PRD_ERR_ALIGN;
    const long npos = (long)b * N * N, nthr = npos * (R / 4);
    hipLaunchKernelGGL(rbf_rows_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, out, z, centers, mask, npos, N, R);
    return (int)hipGetLastError();
}

// out[b][i][j][:] = scale (x[b][i][j][:] + x[b][j][i][:]): the pair symmetrisation in front of the coordinate head (modules.py:403)
// and its backward, in the row layout (both reads are whole 4 P-byte rows: no transpose needed).  One thread per float4.
__global__ __launch_bounds__(256) void sym_rows_kernel(float* __restrict__ out, const float* __restrict__ x, long npos, int N, int P4, float scale) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= npos * P4) return;
    const long pos = tid / P4;
    const int q = (int)(tid - pos * P4);
    const long bi = pos / N;
    const int j = (int)(pos - bi * N);
    const long bb = bi / N;
    const int i = (int)(bi - bb * N);
    const float4 a = *reinterpret_cast<const float4*>(x + (pos * P4 + q) * 4);
    const float4 c = *reinterpret_cast<const float4*>(x + ((((bb * N + j) * N + i) * P4) + q) * 4);
    *reinterpret_cast<float4*>(out + (pos * P4 + q) * 4) = make_float4(scale * (a.x + c.x), scale * (a.y + c.y), scale * (a.z + c.z), scale * (a.w + c.w));
}

// The two reductions of the outer-linear backward over T[r][p][s] (r = (b, i) node rows, p = pair channels, s = single channels;
// T = (dy + dy^T) LN(single), modules.py:283-287 under autograd):  dx[r][s] = sum_p T[r][p][s] w1[p][s]  and
// dw1[c][p][s] = sum over the rows of chunk c of T[r][p][s] x[r][s]  (the caller adds the few chunks).  Neither is a GEMM (s is
// elementwise); torch needs a multiply that writes a T-sized temporary and a sum for each.  One thread per s column, T read
// once per kernel with unit stride along s.
__global__ __launch_bounds__(256) void outer_linear_bwd_dx_kernel(float* __restrict__ dx, const float* __restrict__ T, const float* __restrict__ w1,
                                                                  long R, int P, int S) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= R * S) return;
    const long r = tid / S;
    const int s = (int)(tid - r * S);
    const float* t = T + r * P * S + s;
    float a0 = 0.f, a1 = 0.f;
    int p = 0;
    for (; p + 1 < P; p += 2) {
        a0 += t[(long)p * S] * w1[(long)p * S + s];
        a1 += t[(long)(p + 1) * S] * w1[(long)(p + 1) * S + s];
    }
    if (p < P) a0 += t[(long)p * S] * w1[(long)p * S + s];
    dx[tid] = a0 + a1;
}

__global__ __launch_bounds__(256) void outer_linear_bwd_dw1_kernel(float* __restrict__ part, const float* __restrict__ T, const float* __restrict__ x,
                                                                   long R, int P, int S, int rows_per_chunk) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;          // p * S + s
    if (col >= (long)P * S) return;
    const int s = (int)(col % S);
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long r = r0;
    for (; r + 3 < r1; r += 4) {
        a0 += T[r * P * S + col] * x[r * S + s];
        a1 += T[(r + 1) * P * S + col] * x[(r + 1) * S + s];
        a2 += T[(r + 2) * P * S + col] * x[(r + 2) * S + s];
        a3 += T[(r + 3) * P * S + col] * x[(r + 3) * S + s];
    }
    for (; r < r1; ++r) a0 += T[r * P * S + col] * x[r * S + s];
    part[(long)blockIdx.y * P * S + col] = (a0 + a1) + (a2 + a3);
}

extern "C" int prd_outer_linear_bwd_reduce(float* dx, float* dw1_part, int chunks, const float* T, const float* w1, const float* x,
                                           long long R, int P, int S, hipStream_t stream) {
    if (!dx || !dw1_part || !T || !w1 || !x || R <= 0 || P <= 0 || S <= 0 || chunks <= 0) return PRD_ERR_ARG;
    hipLaunchKernelGGL(outer_linear_bwd_dx_kernel, dim3((unsigned)((R * S + 255) / 256)), dim3(256), 0, stream, dx, T, w1, (long)R, P, S);
    const int rpc = (int)((R + chunks - 1) / chunks);
    hipLaunchKernelGGL(outer_linear_bwd_dw1_kernel, dim3((unsigned)(((long)P * S + 255) / 256), chunks), dim3(256), 0, stream, dw1_part, T, x,
                       (long)R, P, S, rpc);
    return (int)hipGetLastError();
}

extern "C" int prd_sym_rows(float* out, const float* x, float scale, int b, int N, int P, hipStream_t stream) {
    if (!out || !x || b <= 0 || N <= 0 || P <= 0) return PRD_ERR_ARG;
    if (P % 4 || out == x) return PRD_ERR_UNSUPPORTED;                  // (not in place: position (j, i) is read by another thread)
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15) return PRD_ERR_ALIGN;
    const long npos = (long)b * N * N, nthr = npos * (P / 4);
    hipLaunchKernelGGL(sym_rows_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, out, x, npos, N, P / 4, scale);
    return (int)hipGetLastError();
}

extern "C" int prd_sym_transpose(float* out, const float* dy, int b, int N, int P, hipStream_t stream) {
    if (!out || !dy || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dy)) & 15) return PRD_ERR_ALIGN;
    const int T = prd_ceil_div(N, 64);
    const long ntask = (long)b * N * T;
    const int grid = (int)(ntask < 256 * 8 ? ntask : 256 * 8);
    hipLaunchKernelGGL(sym_transpose_kernel, dim3(grid), dim3(256), 0, stream, out, dy, N, P, T, ntask);
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_bwd_operands(float* ops, const float* AB, int b, int N, int P, hipStream_t stream) {
    if (!ops || !AB || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    const int ldn = prd_round_up(N, 32), T = prd_ceil_div(ldn, 64);
    const long ntask = (long)b * P * T * T * 3;
    const int grid = (int)(ntask < 256 * 8 ? ntask : 256 * 8);
    hipLaunchKernelGGL(tri_mul_bwd_operands_kernel, dim3(grid), dim3(256), 0, stream, ops, AB, N, ldn, P, T, ntask);
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_out_bwd(float* dz, float* dgp, float* dO, float* dx1, const float* dy, const float* pair, const float* O,
                                   const float* w_out, const float* b_out, const float* w_ogate, const float* b_ogate,
                                   const float* w_out_t, const float* w_ogate_t, float* x_out, float* lo_out, int dO_batch_channels,
                                   int b, int N, int P, hipStream_t stream) {
    if (!dz || !dgp || !dO || !dx1 || !dy || !pair || !O || !w_out || !b_out || !w_ogate || !b_ogate || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    if (dO_batch_channels == 0) dO_batch_channels = P;
    if (dO_batch_channels < P) return PRD_ERR_ARG;
    constexpr int NWB = 8;
    const int ldn = prd_round_up(N, 32);
    const size_t lds = ((size_t)4 * P * (P + 4) + 2 * P) * sizeof(float);
    const int grid = grid_for((long)b * N * prd_ceil_div(N, 32), 4, 256);
    if (P == 64) {
        PRD_BWD_SET_LDS((tri_mul_out_bwd_kernel<64, NWB>));
        hipLaunchKernelGGL((tri_mul_out_bwd_kernel<64, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dz, dgp, dO, dx1, dy, pair, O, w_out,
                           b_out, w_ogate, b_ogate, w_out_t, w_ogate_t, b, N, ldn, x_out, lo_out, dO_batch_channels);
    } else {
        PRD_BWD_SET_LDS((tri_mul_out_bwd_kernel<32, NWB>));
        hipLaunchKernelGGL((tri_mul_out_bwd_kernel<32, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dz, dgp, dO, dx1, dy, pair, O, w_out,
                           b_out, w_ogate, b_ogate, w_out_t, w_ogate_t, b, N, ldn, x_out, lo_out, dO_batch_channels);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_proj_bwd(float* dpair, float* dpp, float* dpg, const float* dAB, const float* dx1, const float* pair,
                                    const float* mask, const float* w_proj, const float* b_proj, const float* w_gate, const float* b_gate,
                                    const float* w_proj_t, const float* w_gate_t, int incoming, int b, int N, int P, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!dpair || !dpp || !dpg || !dAB || !dx1 || !pair || !mask || !w_proj || !b_proj || !w_gate || !b_gate ||
        b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    constexpr int NWB = 8;
    const int ldn = prd_round_up(N, 32);
    const bool b3 = arith == PRD_ARITH_SPLIT16;
    const size_t lds = b3 ? ((size_t)8 * P * P + 4 * P) * sizeof(float)
                          : ((size_t)2 * 2 * P * (P + 4) + (size_t)2 * P * (2 * P + 4) + 4 * P) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const int grid = grid_for((long)b * N * prd_ceil_div(N, 32), 4, 256);
#define PRD_PBW(PP, BB)                                                                                              \
    do {                                                                                                             \
        PRD_BWD_SET_LDS((tri_mul_proj_bwd_kernel<PP, NWB, BB>));                                                     \
        hipLaunchKernelGGL((tri_mul_proj_bwd_kernel<PP, NWB, BB>), dim3(grid), dim3(NWB * 64), lds, stream, dpair, dpp, dpg, dAB, dx1, pair, \
                           mask, w_proj, b_proj, w_gate, b_gate, w_proj_t, w_gate_t, b, N, ldn, incoming);           \
    } while (0)
    if (P == 64) { if (b3) PRD_PBW(64, true); else PRD_PBW(64, false); }
    else { if (b3) PRD_PBW(32, true); else PRD_PBW(32, false); }
#undef PRD_PBW
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn_bwd_core(float* dqkvg, const float* dog, const float* pair, const float* mask, const float* wq,
                                     const float* wk, const float* wv, const float* wg, const float* bg, int ending,
                                     int b, int N, int P, int H, int c, hipStream_t stream) {
    if (!dqkvg || !dog || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    const int npad = prd_round_up(N, 32);
    const size_t lds = ((size_t)64 * (P + 4) + (size_t)4 * npad * TB_PITCH + 4 * (size_t)npad) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;          // rows beyond N = 416: not in this cut
    const long nwork = (long)b * N * H;
    const int grid = (int)(nwork < 256 ? nwork : 256);
    const int nthreads = 512;                                   // 8 waves: two per SIMD cover each other's dependent MFMA chains
    if (P == 64) {
        PRD_BWD_SET_LDS(tri_attn_bwd_core_kernel<64>);
        hipLaunchKernelGGL(tri_attn_bwd_core_kernel<64>, dim3(grid), dim3(nthreads), lds, stream, dqkvg, dog, pair, mask, wq, wk, wv, wg, bg, b, N, npad, H, ending);
    } else {
        PRD_BWD_SET_LDS(tri_attn_bwd_core_kernel<32>);
        hipLaunchKernelGGL(tri_attn_bwd_core_kernel<32>, dim3(grid), dim3(nthreads), lds, stream, dqkvg, dog, pair, mask, wq, wk, wv, wg, bg, b, N, npad, H, ending);
    }
    return (int)hipGetLastError();
}

// ---- a linear at every pair position as a ROW kernel (the GEMMs of the training backward with 2e5 rows and K, N <= 256) --------
// prd_gemm's tile kernels give one 64 x 64 tile (one K chunk!) to a workgroup: at K = 64 that is a load -> split -> LDS -> MFMA ->
// store chain without anything to overlap it with, 2 TB/s for K = 64 -> N = 256.  Here the weights (<= 64 KB as fp16 hi | lo) stay
// in LDS for the life of a persistent workgroup and every wave streams 32-row blocks through them: y = act(LN?(x) W^T + bias),
// optionally zeroed where mask_pos <= 0 (the ReLU backward from recomputed activations).  Shapes: (K, OUT) = (64, 64), (64, 256),
// (256, 64).  (A fused LayerNorm-backward tail for the 256 -> 64 form was measured no faster than this kernel + ln_rows_bwd64 --
// 77.8 vs 47.6 + 30 us -- and is not built.)
namespace {
// Un-normalised operand rows (gradients: 1e-4 ... 1e-9 per element near a minimum, include/prd_hip.h OPERAND RANGE) are brought to
// [1, 2) by an EXACT power of two before the fp16 hi | lo split and the factor is taken out of the accumulator again: biased
// exponent of the largest magnitude of a row piece (both lane halves), clamped so that both factors are normal numbers.
template <int KH>
PRD_DEV unsigned row_exp_biased(const float (&x)[KH]) {
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int k = 0; k < KH; k += 2) { m0 = fmaxf(m0, fabsf(x[k])); m1 = fmaxf(m1, fabsf(x[k + 1])); }
    float m = fmaxf(m0, m1);
    m = fmaxf(m, __shfl_xor(m, 32));
    unsigned eb = (__float_as_uint(m) >> 23) & 255u;
    return eb < 1u ? 1u : (eb > 253u ? 253u : eb);      // an all-zero (or denormal) row: factor 2^126 of zeros; inf / NaN stay loud
}
PRD_DEV float pow2_from_biased(unsigned eb) { return __uint_as_float(eb << 23); }             // 2^(eb - 127)
PRD_DEV float inv_pow2_from_biased(unsigned eb) { return __uint_as_float((254u - eb) << 23); } // 2^(127 - eb)

template <int K, int NW>
__global__ __launch_bounds__(NW * 64) void pair_linear_rows_kernel(
    float* __restrict__ out, float* __restrict__ xn_out, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ mask_pos, long rows, int OUT, int ln_in, int act, int w_kn) {
    extern __shared__ __attribute__((aligned(16))) float smem_pl[];
    u32x4* Wimg = reinterpret_cast<u32x4*>(smem_pl);
    float* bl = smem_pl + (size_t)OUT * K;
    const int NT = NW * 64;
    if (w_kn) stage_weight_h2_t<K>(Wimg, w, OUT, OUT, threadIdx.x, NT, H2_WSCALE);        // w given as [K][OUT]
    else if (OUT == 64) stage_weight_h2<K>(Wimg, w, 64, K, threadIdx.x, NT, H2_WSCALE);
    else stage_weight_h2<K>(Wimg, w, 256, K, threadIdx.x, NT, H2_WSCALE);
    stage_vec_cll(bl, bias, OUT, threadIdx.x, NT);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long ntask = (rows + 31) / 32;
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const long row = task * 32 + r;
        const bool valid = row < rows;
        const long rowc = valid ? row : 0;
        if constexpr (K == 64) {
            float xv[32];
            load_row_cll<64>(x + rowc * 64, hi, valid, xv);
            float back = H2_INV_WSCALE;                 // accumulator -> result
            if (ln_in) {
                ln_cll<32>(xv);
                if (xn_out) store_row_cll<64>(xn_out + rowc * 64, hi, valid, xv);
            } else {                                    // a raw row: normalise by a power of two (see row_exp_biased)
                const unsigned eb = row_exp_biased<32>(xv);
                const float dn = inv_pow2_from_biased(eb);
#pragma unroll
                for (int k = 0; k < 32; ++k) xv[k] *= dn;
                back = H2_INV_WSCALE * pow2_from_biased(eb);
            }
            u32x4 ps[2][4];
            split2h_cll<32>(xv, ps);
            for (int j = 0; j < OUT / 64; ++j) {
                f32x16 acc[2];
                zero_acc(acc);
                rowgemm_h2<64, 2>(Wimg, OUT, 64 * j, ps, acc, r, hi);
                float y[32];
#pragma unroll
                for (int s_ = 0; s_ < 32; ++s_) {
                    float v = acc[s_ >> 4][s_ & 15] * back + bl[hi * (OUT / 2) + 32 * j + s_];
                    y[s_] = act == 1 ? relu_nan(v) : v;
                }
                if (mask_pos) {
                    float m[32];
                    load_row_cll<64>(mask_pos + rowc * OUT + 64 * j, hi, valid, m);
#pragma unroll
                    for (int s_ = 0; s_ < 32; ++s_) y[s_] = m[s_] > 0.f ? y[s_] : 0.f;
                }
                store_row_cll<64>(out + rowc * OUT + 64 * j, hi, valid, y);
            }
        } else {                                        // K = 256 -> 64 outputs, the row in four 64-channel pieces
            f32x16 acc[2];
            zero_acc(acc);
            // running power-of-two scale of the row over its four pieces: a piece with a larger exponent than any before it
            // rescales the accumulators (exactly), every piece is split at the scale of the largest so far
            unsigned eb_run = 1u;
            auto piece = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                float xv[32];
                load_row_cll<64>(x + rowc * 256 + 64 * c, hi, valid, xv);
                const unsigned eb = row_exp_biased<32>(xv);
                if (c == 0) eb_run = eb;
                else if (eb > eb_run) {
                    const float f = (eb - eb_run) >= 126u ? 0.f : __uint_as_float((127u - (eb - eb_run)) << 23);     // 2^(eb_run - eb)
#pragma unroll
                    for (int s_ = 0; s_ < 32; ++s_) acc[s_ >> 4][s_ & 15] *= f;
                    eb_run = eb;
                }
                const float dn = inv_pow2_from_biased(eb_run);
#pragma unroll
                for (int k = 0; k < 32; ++k) xv[k] *= dn;
                u32x4 ps[2][4];
                split2h_cll<32>(xv, ps);
                rowgemm_h2_part<256, 2, 4 * c, 4 * c + 4>(Wimg, 64, 0, ps, acc, r, hi);
            };
            piece(std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 1>{});
            piece(std::integral_constant<int, 2>{});
            piece(std::integral_constant<int, 3>{});
            const float back = H2_INV_WSCALE * pow2_from_biased(eb_run);
            float y[32];
#pragma unroll
            for (int s_ = 0; s_ < 32; ++s_) {
                const float v = acc[s_ >> 4][s_ & 15] * back + bl[hi * 32 + s_];
                y[s_] = act == 1 ? relu_nan(v) : v;
            }
            if (mask_pos) {
                float m[32];
                load_row_cll<64>(mask_pos + rowc * 64, hi, valid, m);
#pragma unroll
                for (int s_ = 0; s_ < 32; ++s_) y[s_] = m[s_] > 0.f ? y[s_] : 0.f;
            }
            store_row_cll<64>(out + rowc * 64, hi, valid, y);
        }
    }
}

}  // namespace

extern "C" int prd_pair_linear_supported(int K, int OUT, int arith) {
    if (arith < 0 || (arith & 0xff) != PRD_ARITH_SPLIT16) return 0;
    return ((K == 64 && (OUT == 64 || OUT == 256)) || (K == 256 && OUT == 64)) ? 1 : 0;
}

extern "C" int prd_pair_linear(float* out, const float* x, const float* w, const float* bias, long long rows, int K, int OUT,
                               int ln_in, float* xn_out, int act, const float* mask_pos, int w_kn, int arith, hipStream_t stream) {
    if (!out || !x || !w || rows <= 0 || act < 0 || act > 1) return PRD_ERR_ARG;
    if (!prd_pair_linear_supported(K, OUT, arith)) return PRD_ERR_UNSUPPORTED;
    if ((ln_in && K != 64) || (xn_out && !ln_in)) return PRD_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xn_out) |
         reinterpret_cast<uintptr_t>(mask_pos)) & 15)
        return PRD_ERR_ALIGN;
    constexpr int NWP = 8;
    const size_t lds = ((size_t)OUT * K + OUT) * sizeof(float);
    const long ntask = (rows + 31) / 32;
    const int grid = grid_for(ntask, NWP, 256);          // one persistent workgroup per CU (the register budget allows two waves per SIMD)
    if (K == 64) {
        PRD_BWD_SET_LDS((pair_linear_rows_kernel<64, NWP>));
        hipLaunchKernelGGL((pair_linear_rows_kernel<64, NWP>), dim3(grid), dim3(NWP * 64), lds, stream, out, xn_out, x, w, bias, mask_pos,
                           (long)rows, OUT, ln_in, act, w_kn);
    } else {
        PRD_BWD_SET_LDS((pair_linear_rows_kernel<256, NWP>));
        hipLaunchKernelGGL((pair_linear_rows_kernel<256, NWP>), dim3(grid), dim3(NWP * 64), lds, stream, out, xn_out, x, w, bias, mask_pos,
                           (long)rows, OUT, ln_in, act, w_kn);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_ln_rows_bwd(float* dx, const float* dy, const float* x, const float* res, long long rows, int C, hipStream_t stream) {
    if (!dx || !dy || !x || rows <= 0 || C <= 0) return PRD_ERR_ARG;
    const bool al16 = ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
    if (C == 64 && al16)
        hipLaunchKernelGGL(ln_rows_bwd64_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, dx, dy, x, res, (long)rows);
    else
        hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dx, dy, x, res, (long)rows, C);
    return (int)hipGetLastError();
}

extern "C" int prd_pair_bias_bwd(float* dx, float* xn, float* d2, const float* dbias, const float* wf, const float* x, int b, long long nn,
                                 int H, int P, hipStream_t stream) {
    if (!dx || !dbias || !wf || !x || b <= 0 || nn <= 0) return PRD_ERR_ARG;
    if (P != 64 || (H != 4 && H != 8)) return PRD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wf)) & 15) return PRD_ERR_ALIGN;
    const long rows = (long)b * nn;
    const dim3 grid((unsigned)((rows + 15) / 16));
    if (H == 4) hipLaunchKernelGGL(pair_bias_bwd64_kernel<4>, grid, dim3(256), 0, stream, dx, xn, d2, dbias, wf, x, rows, (long)nn);
    else hipLaunchKernelGGL(pair_bias_bwd64_kernel<8>, grid, dim3(256), 0, stream, dx, xn, d2, dbias, wf, x, rows, (long)nn);
    return (int)hipGetLastError();
}

static long wgrad_slabs(long long rows) {
    long slabs = rows / 512 < 256 ? (long)((rows + 511) / 512) : 256;
    return slabs < 1 ? 1 : slabs;
}

extern "C" size_t prd_linear_wgrad_workspace(long long rows, int O, int I) {
    if (rows <= 0 || O <= 0 || I <= 0) return 0;
    return (size_t)wgrad_slabs(rows) * ((size_t)O * I + O) * sizeof(float);        // dW and db partials of every slab
}

extern "C" int prd_linear_wgrad(float* dw, float* db, const float* dy, const float* x, long long rows, int O, int I, int lddy, int ldx,
                                float* ws, size_t ws_bytes, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    (void)tune;
    if (!dw || !dy || !x || !ws || rows <= 0 || O <= 0 || I <= 0) return PRD_ERR_ARG;
    const bool narrow = O <= 16;
    if ((!narrow && (O % 64)) || (I % 64) || O > 256 || I > 256) return PRD_ERR_UNSUPPORTED;
    // the wide kernel reads dy / x and writes its partials with 8-byte accesses: even leading dimensions AND 8-byte aligned pointers
    if (!narrow && ((lddy & 1) || (ldx & 1) || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws)) & 7)))
        return PRD_ERR_ALIGN;
    if (lddy < O || ldx < I) return PRD_ERR_ARG;
    if (ws_bytes < prd_linear_wgrad_workspace(rows, O, I)) return PRD_ERR_WORKSPACE;
    const long slabs = wgrad_slabs(rows);
    const int rows_per_wg = (int)((rows + slabs - 1) / slabs);
    const int want_db = db ? 1 : 0;
    const int nw = O * I, n = nw + (want_db ? O : 0);
    if (narrow) {
        dim3 grid((unsigned)slabs, I / 64);
        if (O <= 4) hipLaunchKernelGGL(linear_wgrad_narrow_kernel<4>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
        else hipLaunchKernelGGL(linear_wgrad_narrow_kernel<16>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
    } else {
        const int nblk = (O / 64) * (I / 64);
        const int per = (nblk % 4 == 0) ? 4 : ((nblk % 2 == 0) ? 2 : 1);     // 64x64 blocks per workgroup
        dim3 grid((unsigned)slabs, nblk / per);
        if (arith == PRD_ARITH_SPLIT16) {
            if (per == 4) hipLaunchKernelGGL(linear_wgrad_h2_kernel<4>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
            else if (per == 2) hipLaunchKernelGGL(linear_wgrad_h2_kernel<2>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
            else hipLaunchKernelGGL(linear_wgrad_h2_kernel<1>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
        } else if (per == 4) hipLaunchKernelGGL(linear_wgrad_kernel<4>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
        else if (per == 2) hipLaunchKernelGGL(linear_wgrad_kernel<2>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
        else hipLaunchKernelGGL(linear_wgrad_kernel<1>, grid, dim3(256), 0, stream, ws, dy, x, (long)rows, O, I, lddy, ldx, rows_per_wg, want_db);
    }
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, dw, db, ws, n, nw, (int)slabs);
    return (int)hipGetLastError();
}

extern "C" int prd_embed_wgrad_multi(float* dtables, const long long* const* idx, const float* const* row_scale, const int* card, int K,
                                     const float* dy, long long rows, int C, int lddy, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (!dtables || !idx || !card || !dy || !ws || K <= 0 || rows <= 0 || C <= 0 || lddy < C) return PRD_ERR_ARG;
    if (K > 8 || C > 64) return PRD_ERR_UNSUPPORTED;
    EmbedMulti em{};
    em.K = K;
    int total = 0;
    for (int k = 0; k < K; ++k) {
        if (!idx[k] || card[k] <= 0) return PRD_ERR_ARG;
        em.idx[k] = idx[k];
        em.scale[k] = row_scale ? row_scale[k] : nullptr;
        em.off[k] = total;
        em.card[k] = card[k];
        total += card[k];
    }
    em.total = total;
    if (total > 128) return PRD_ERR_UNSUPPORTED;
    if (ws_bytes < prd_embed_wgrad_workspace(rows, total, C)) return PRD_ERR_WORKSPACE;
    const long slabs = wgrad_slabs(rows);
    const int rows_per_wg = (int)((rows + slabs - 1) / slabs);
    const size_t lds = (size_t)4 * total * 64 * sizeof(float);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)embed_wgrad_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(embed_wgrad_multi_kernel, dim3((unsigned)slabs), dim3(256), lds, stream, ws, em, dy, (long)rows, C, lddy, rows_per_wg);
    const int n = total * C;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, dtables, (float*)nullptr, ws, n, n, (int)slabs);
    return (int)hipGetLastError();
}

extern "C" size_t prd_embed_wgrad_workspace(long long rows, int card, int C) {
    if (rows <= 0 || card <= 0 || C <= 0) return 0;
    return (size_t)wgrad_slabs(rows) * card * C * sizeof(float);
}

extern "C" int prd_embed_wgrad(float* dtable, const long long* idx, const float* dy, const float* row_scale, long long rows, int card, int C,
                               int lddy, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (!dtable || !idx || !dy || !ws || rows <= 0 || card <= 0 || C <= 0 || lddy < C) return PRD_ERR_ARG;
    if (C > 64 || card > 128) return PRD_ERR_UNSUPPORTED;
    if (ws_bytes < prd_embed_wgrad_workspace(rows, card, C)) return PRD_ERR_WORKSPACE;
    const long slabs = wgrad_slabs(rows);
    const int rows_per_wg = (int)((rows + slabs - 1) / slabs);
    const size_t lds = (size_t)4 * card * 64 * sizeof(float);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)embed_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(embed_wgrad_kernel, dim3((unsigned)slabs), dim3(256), lds, stream, ws, idx, dy, row_scale, (long)rows, card, C, lddy, rows_per_wg);
    const int n = card * C;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, dtable, (float*)nullptr, ws, n, n, (int)slabs);
    return (int)hipGetLastError();
}
