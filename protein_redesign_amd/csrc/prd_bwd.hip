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
    const float* __restrict__ woT, const float* __restrict__ wogT, int b, int N, int ldn) {
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
    stage_weight_cll<P>(WoTl, woT, P, P, threadIdx.x, NT);
    stage_weight_cll<P>(WgTl, wogT, P, P, threadIdx.x, NT);
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
        {
            f32x16 ag[NB];
            zero_acc(ag);
            rowgemm<P, NB>(Wgl, x, ag, r, hi);
#pragma unroll
            for (int s = 0; s < KH; ++s) g[s] = sigmoidf_(ag[s >> 4][s & 15] + bgl[hi * KH + s]);
        }
        float lo[KH];
#pragma unroll
        for (int s = 0; s < KH; ++s) lo[s] = valid ? O[(((long)bb * P + cll_ch(s, hi)) * N + i) * ldn + jj] : 0.f;
        const float rstd_o = ln_cll_rstd<KH>(lo);
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
            if (valid) {
#pragma unroll
                for (int s = 0; s < KH; ++s) dO[(((long)bb * P + cll_ch(s, hi)) * N + i) * ldn + jj] = dlo[s];
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

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_mul_proj_bwd_kernel(
    float* __restrict__ dpair, float* __restrict__ dpp_out, float* __restrict__ dpg_out,
    const float* __restrict__ dAB, const float* __restrict__ dx1, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ wpT, const float* __restrict__ wgT, int b, int N, int ldn, int incoming) {
    constexpr int KH = P / 2, NB = P / 32, OUT = 2 * P;
    extern __shared__ __attribute__((aligned(16))) float smem_b3[];
    float* Wpl = smem_b3;                        // [2P][P+4]
    float* Wgl = Wpl + OUT * (P + 4);
    float* WpTl = Wgl + OUT * (P + 4);           // [P][2P+4]: rows = input channels of the projection, K = its 2P outputs
    float* WgTl = WpTl + P * (OUT + 4);
    float* bpl = WgTl + P * (OUT + 4);           // [2P] CLL
    float* bgl = bpl + OUT;
    const int NT = NW * 64;
    stage_weight_cll<P>(Wpl, wp, OUT, P, threadIdx.x, NT);
    stage_weight_cll<P>(Wgl, wg, OUT, P, threadIdx.x, NT);
    stage_weight_cll<OUT>(WpTl, wpT, P, OUT, threadIdx.x, NT);
    stage_weight_cll<OUT>(WgTl, wgT, P, OUT, threadIdx.x, NT);
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
#pragma unroll
        for (int h = 0; h < 2; ++h) {            // h = 0: the a operand (output channels 0 .. P-1), h = 1: the b operand
            f32x16 ap[NB], ag[NB];
            zero_acc(ap);
            zero_acc(ag);
            rowgemm<P, NB>(Wpl + h * P * (P + 4), x, ap, r, hi);
            rowgemm<P, NB>(Wgl + h * P * (P + 4), x, ag, r, hi);
            float dpp[KH], dpg[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) {
                const int ch = h * P + cll_ch(s, hi);
                const float dab = valid ? dAB[(((long)bb * OUT + ch) * N + u) * ldn + vv] : 0.f;
                const float pp = ap[s >> 4][s & 15] + bpl[hi * P + h * KH + s];
                const float sg = sigmoidf_(ag[s >> 4][s & 15] + bgl[hi * P + h * KH + s]);
                dpp[s] = dab * m2 * sg;
                dpg[s] = dab * m2 * pp * sg * (1.0f - sg);
            }
            // kept for the weight-gradient GEMMs: row layout [pair position][2P], this half at columns h P ..
            store_row_cll<P>(dpp_out + prow * OUT + h * P, hi, valid, dpp);
            store_row_cll<P>(dpg_out + prow * OUT + h * P, hi, valid, dpg);
            // dx += Wp[h]^T dpp + Wg[h]^T dpg: the half is CLL elements [32 h, 32 h + 32) of the 2P-wide K axis = groups [8h, 8h+8)
            if (h == 0) {
                rowgemm_part<OUT, NB, 0, KH / 4>(WpTl, dpp, adx, r, hi);
                rowgemm_part<OUT, NB, 0, KH / 4>(WgTl, dpg, adx, r, hi);
            } else {
                rowgemm_part<OUT, NB, KH / 4, KH / 2>(WpTl, dpp, adx, r, hi);
                rowgemm_part<OUT, NB, KH / 4, KH / 2>(WgTl, dpg, adx, r, hi);
            }
        }
        float dx[KH];
        load_row_cll<P>(dx1 + prow * P, hi, valid, dx);
#pragma unroll
        for (int s = 0; s < KH; ++s) dx[s] += adx[s >> 4][s & 15];
        ln_cll_bwd<KH>(dx, x, rstd_x);
        store_row_cll<P>(dpair + prow * P, hi, valid, dx);
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

extern "C" int prd_tri_mul_out_bwd(float* dz, float* dgp, float* dO, float* dx1, const float* dy, const float* pair, const float* O,
                                   const float* w_out, const float* b_out, const float* w_ogate, const float* b_ogate,
                                   const float* w_out_t, const float* w_ogate_t, int b, int N, int P, hipStream_t stream) {
    if (!dz || !dgp || !dO || !dx1 || !dy || !pair || !O || !w_out || !b_out || !w_ogate || !b_ogate || !w_out_t || !w_ogate_t ||
        b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    constexpr int NWB = 8;
    const int ldn = prd_round_up(N, 32);
    const size_t lds = ((size_t)4 * P * (P + 4) + 2 * P) * sizeof(float);
    const int grid = grid_for((long)b * N * prd_ceil_div(N, 32), 4, 256);
    if (P == 64) {
        PRD_BWD_SET_LDS((tri_mul_out_bwd_kernel<64, NWB>));
        hipLaunchKernelGGL((tri_mul_out_bwd_kernel<64, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dz, dgp, dO, dx1, dy, pair, O, w_out,
                           b_out, w_ogate, b_ogate, w_out_t, w_ogate_t, b, N, ldn);
    } else {
        PRD_BWD_SET_LDS((tri_mul_out_bwd_kernel<32, NWB>));
        hipLaunchKernelGGL((tri_mul_out_bwd_kernel<32, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dz, dgp, dO, dx1, dy, pair, O, w_out,
                           b_out, w_ogate, b_ogate, w_out_t, w_ogate_t, b, N, ldn);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_proj_bwd(float* dpair, float* dpp, float* dpg, const float* dAB, const float* dx1, const float* pair,
                                    const float* mask, const float* w_proj, const float* b_proj, const float* w_gate, const float* b_gate,
                                    const float* w_proj_t, const float* w_gate_t, int incoming, int b, int N, int P, hipStream_t stream) {
    if (!dpair || !dpp || !dpg || !dAB || !dx1 || !pair || !mask || !w_proj || !b_proj || !w_gate || !b_gate || !w_proj_t || !w_gate_t ||
        b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    constexpr int NWB = 8;
    const int ldn = prd_round_up(N, 32);
    const size_t lds = ((size_t)2 * 2 * P * (P + 4) + (size_t)2 * P * (2 * P + 4) + 4 * P) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const int grid = grid_for((long)b * N * prd_ceil_div(N, 32), 4, 256);
    if (P == 64) {
        PRD_BWD_SET_LDS((tri_mul_proj_bwd_kernel<64, NWB>));
        hipLaunchKernelGGL((tri_mul_proj_bwd_kernel<64, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dpair, dpp, dpg, dAB, dx1, pair, mask,
                           w_proj, b_proj, w_gate, b_gate, w_proj_t, w_gate_t, b, N, ldn, incoming);
    } else {
        PRD_BWD_SET_LDS((tri_mul_proj_bwd_kernel<32, NWB>));
        hipLaunchKernelGGL((tri_mul_proj_bwd_kernel<32, NWB>), dim3(grid), dim3(NWB * 64), lds, stream, dpair, dpp, dpg, dAB, dx1, pair, mask,
                           w_proj, b_proj, w_gate, b_gate, w_proj_t, w_gate_t, b, N, ldn, incoming);
    }
    return (int)hipGetLastError();
}
