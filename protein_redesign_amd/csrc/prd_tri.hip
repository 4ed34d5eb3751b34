// Pair-track kernels, part 2: triangle multiplication and triangle attention.
//
// Triangle multiplication (reference modules.py:262-274) = three launches:
//   tri_mul_proj : p = LN(pair); ab = m2 * sigmoid(Wg p + bg) * (Wp p + bp), stored CHANNEL-MAJOR
//                  AB[b][2P][N][ldn] (ldn = round_up(N,32), zero padded) so that the contraction is
//                  2P... P independent, K-contiguous N x N x N GEMMs.  "incoming" writes the
//                  transposed operand (a[k,i] -> A[i][k]) so both modes share one contraction.
//   prd_gemm     : O[b][d][i][j] = sum_k A[d][i][k] B[d][j][k]      (batched, MFMA)
//   tri_mul_out  : pair += sigmoid(Wog LN(pair) + bog) * (Wo LN_d(O) + bo)
// Triangle attention (modules.py:236-243 -> 185-225) = two launches:
//   tri_attn_core: one workgroup per (b, row, head): K_h / V_h of the whole row live in LDS, the
//                  N x N logits never exist in memory (flash-style online softmax in registers,
//                  16x16x4 f32 MFMA for QK^T and PV in the "swapped" form so that every softmax
//                  quantity of a query is lane-local); writes the gated per-head output
//                  og[b,N,N,H*c].
//   tri_attn_out : pair += Wo og + bo.
#include "prd_common.h"
#include "../../include/prd_hip.h"

namespace {

constexpr int WG = 256;

// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(WG) void tri_mul_proj_kernel(float* __restrict__ AB, const float* __restrict__ pair,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ wp, const float* __restrict__ bp,
                                                          const float* __restrict__ wg, const float* __restrict__ bg,
                                                          int b, int N, int ldn, int incoming) {
    constexpr int KH = P / 2, OUT = 2 * P, OB = OUT / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wpl = smem;                       // [2P][P+4]
    float* Wgl = Wpl + OUT * (P + 4);
    float* bpl = Wgl + OUT * (P + 4);        // [2P] CLL
    float* bgl = bpl + OUT;
    stage_weight_cll<P>(Wpl, wp, OUT, P, threadIdx.x, WG);
    stage_weight_cll<P>(Wgl, wg, OUT, P, threadIdx.x, WG);
    stage_vec_cll(bpl, bp, OUT, threadIdx.x, WG);
    stage_vec_cll(bgl, bg, OUT, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = ldn / 32;
    const long ntask = (long)b * N * nvb;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const int vb = (int)(task % nvb);
        const long bu = task / nvb;                 // bb*N + u
        const int bb = (int)(bu / N), u = (int)(bu - (long)bb * N);
        const int v = vb * 32 + r;
        const bool valid = v < N;
        const int vv = valid ? v : 0;
        // outgoing: operand row u, contraction index v <-> pair[u, v]; incoming: pair[v, u]
        const long pos = incoming ? (((long)bb * N + vv) * N + u) : (bu * N + vv);
        float x[KH];
        load_row_cll<P>(pair + pos * P, hi, valid, x);
        ln_cll<KH>(x);
        const float m2 = valid ? mask[bu] * mask[(long)bb * N + vv] : 0.f;
#pragma unroll 1
        for (int ob = 0; ob < OB; ++ob) {
            f32x16 ap[1], ag[1];
            zero_acc(ap);
            zero_acc(ag);
            rowgemm<P, 1>(Wpl + ob * 32 * (P + 4), x, ap, r, hi);
            rowgemm<P, 1>(Wgl + ob * 32 * (P + 4), x, ag, r, hi);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int s = ob * 16 + q;                       // CLL element of the 2P-wide output
                const int c = 32 * ob + drow32(q, hi);           // output channel
                const float val = m2 * sigmoidf_(ag[0][q] + bgl[hi * P + s]) * (ap[0][q] + bpl[hi * P + s]);
                AB[(((long)bb * OUT + c) * N + u) * ldn + v] = valid ? val : 0.f;
            }
        }
    }
}

template <int P>
__global__ __launch_bounds__(WG) void tri_mul_out_kernel(float* out, const float* pair, const float* __restrict__ O,
                                                         const float* __restrict__ wo, const float* __restrict__ bo,
                                                         const float* __restrict__ wog, const float* __restrict__ bog,
                                                         int b, int N, int ldn, int residual) {
    constexpr int KH = P / 2, NB = P / 32;
    __shared__ __attribute__((aligned(16))) float Wol[P * (P + 4)];
    __shared__ __attribute__((aligned(16))) float Wgl[P * (P + 4)];
    __shared__ __attribute__((aligned(16))) float bol[P];
    __shared__ __attribute__((aligned(16))) float bgl[P];
    stage_weight_cll<P>(Wol, wo, P, P, threadIdx.x, WG);
    stage_weight_cll<P>(Wgl, wog, P, P, threadIdx.x, WG);
    stage_vec_cll(bol, bo, P, threadIdx.x, WG);
    stage_vec_cll(bgl, bog, P, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;
        const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const long off = (bi * N + jj) * P;
        float raw[KH], x[KH];
        load_row_cll<P>(pair + off, hi, valid, raw);
#pragma unroll
        for (int s = 0; s < KH; ++s) x[s] = raw[s];
        ln_cll<KH>(x);
        f32x16 ag[NB];
        zero_acc(ag);
        rowgemm<P, NB>(Wgl, x, ag, r, hi);
        // contraction output of this (i,j) for the lane's channels (coalesced over j per channel)
#pragma unroll
        for (int s = 0; s < KH; ++s)
            x[s] = valid ? O[(((long)bb * P + cll_ch(s, hi)) * N + i) * ldn + jj] : 0.f;
        ln_cll<KH>(x);
        f32x16 ao[NB];
        zero_acc(ao);
        rowgemm<P, NB>(Wol, x, ao, r, hi);
#pragma unroll
        for (int s = 0; s < KH; ++s)
            raw[s] = (residual ? raw[s] : 0.f) + sigmoidf_(ag[s >> 4][s & 15] + bgl[hi * KH + s]) * (ao[s >> 4][s & 15] + bol[hi * KH + s]);
        store_row_cll<P>(out + off, hi, valid, raw);
    }
}

// ---------------------------------------------------------------------------------------------
// triangle attention core.  HC = H*c = 64, c = 16.  Workgroup = 8 waves, one (b, row u, head h).
// ---------------------------------------------------------------------------------------------
constexpr int KP = 20;          // LDS pitch (floats) of the [*, 16] K / Q / G tiles

template <int P, int TA_WAVES>
__global__ __launch_bounds__(TA_WAVES * 64) void tri_attn_core_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int npad, int H, int ending) {
    constexpr int KH = P / 2, C = 16, HC = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wkv = smem;                         // [32][P+4]: rows 0-15 = k_h, 16-31 = v_h
    float* Wqg = Wkv + 32 * (P + 4);           // [32][P+4]: rows 0-15 = q_h, 16-31 = g_h
    float* Kl = Wqg + 32 * (P + 4);            // [npad][KP]
    float* Vt = Kl + npad * KP;                // [16][npad+4]
    float* km = Vt + C * (npad + 4);           // [npad] key mask value of this row
    float* scratch = km + npad;                // per wave: Q [32][KP], G [32][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    float* Qs = scratch + wave * (2 * 32 * KP);
    float* Gs = Qs + 32 * KP;
    const int nvb = npad / 32;
    const long ntask = (long)b * N * H;
    for (long task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int h = (int)(task % H);
        const long bu = task / H;
        const int bb = (int)(bu / N), u = (int)(bu - (long)bb * N);
        __syncthreads();                        // previous task's LDS fully consumed
        stage_weight_cll<P>(Wkv, wk + (long)h * C * P, C, P, tid, TA_WAVES * 64);
        stage_weight_cll<P>(Wkv + C * (P + 4), wv + (long)h * C * P, C, P, tid, TA_WAVES * 64);
        stage_weight_cll<P>(Wqg, wq + (long)h * C * P, C, P, tid, TA_WAVES * 64);
        stage_weight_cll<P>(Wqg + C * (P + 4), wg + (long)h * C * P, C, P, tid, TA_WAVES * 64);
        const float mu = mask[bu];
        for (int k = tid; k < npad; k += TA_WAVES * 64) km[k] = (k < N) ? mu * mask[(long)bb * N + k] : 0.f;
        __syncthreads();
        // ---- phase 1: K_h, V_h of every position of the row ----
        for (int vb = wave; vb < nvb; vb += TA_WAVES) {
            const int v = vb * 32 + r;
            const bool valid = v < N;
            const int vv = valid ? v : 0;
            const long pos = ending ? (((long)bb * N + vv) * N + u) : (bu * N + vv);
            float x[KH];
            load_row_cll<P>(pair + pos * P, hi, valid, x);
            ln_cll<KH>(x);
            f32x16 acc[1];
            zero_acc(acc);
            rowgemm<P, 1>(Wkv, x, acc, r, hi);
            // D rows 0-15 = k channels {4hi+e, 8+4hi+e}; rows 16-31 = v channels likewise
            *reinterpret_cast<float4*>(Kl + v * KP + 4 * hi) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            *reinterpret_cast<float4*>(Kl + v * KP + 8 + 4 * hi) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(4 * hi + e) * (npad + 4) + v] = acc[0][8 + e];
                Vt[(8 + 4 * hi + e) * (npad + 4) + v] = acc[0][12 + e];
            }
        }
        __syncthreads();
        // ---- phase 2: queries in blocks of 32 per wave ----
        const int ql = lane & 15, g4 = lane >> 4;
        for (int qb = wave; qb < nvb; qb += TA_WAVES) {
            {
                const int v = qb * 32 + r;
                const bool valid = v < N;
                const int vv = valid ? v : 0;
                const long pos = ending ? (((long)bb * N + vv) * N + u) : (bu * N + vv);
                float x[KH];
                load_row_cll<P>(pair + pos * P, hi, valid, x);
                ln_cll<KH>(x);
                f32x16 acc[1];
                zero_acc(acc);
                rowgemm<P, 1>(Wqg, x, acc, r, hi);
                const float sc = 0.25f;                    // 1/sqrt(c), c = 16 (modules.py:176, 216)
                *reinterpret_cast<float4*>(Qs + r * KP + 4 * hi) = make_float4(sc * acc[0][0], sc * acc[0][1], sc * acc[0][2], sc * acc[0][3]);
                *reinterpret_cast<float4*>(Qs + r * KP + 8 + 4 * hi) = make_float4(sc * acc[0][4], sc * acc[0][5], sc * acc[0][6], sc * acc[0][7]);
                const float* bgh = bg + h * C;
                *reinterpret_cast<float4*>(Gs + r * KP + 4 * hi) =
                    make_float4(sigmoidf_(acc[0][8] + bgh[4 * hi]), sigmoidf_(acc[0][9] + bgh[4 * hi + 1]),
                                sigmoidf_(acc[0][10] + bgh[4 * hi + 2]), sigmoidf_(acc[0][11] + bgh[4 * hi + 3]));
                *reinterpret_cast<float4*>(Gs + r * KP + 8 + 4 * hi) =
                    make_float4(sigmoidf_(acc[0][12] + bgh[8 + 4 * hi]), sigmoidf_(acc[0][13] + bgh[8 + 4 * hi + 1]),
                                sigmoidf_(acc[0][14] + bgh[8 + 4 * hi + 2]), sigmoidf_(acc[0][15] + bgh[8 + 4 * hi + 3]));
            }
            wave_lds_fence();
#pragma unroll 1
            for (int qt = 0; qt < 2; ++qt) {
                const int qrow = qt * 16 + ql;                          // row inside the 32-block
                const float4 qf = *reinterpret_cast<const float4*>(Qs + qrow * KP + 4 * g4);
                float m_run = -1e30f, l_run = 0.f;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};                         // O^T[c = 4*g4 + e][q = ql]
                for (int kt = 0; kt < npad / 16; ++kt) {
                    const int key0 = kt * 16;
                    const float4 kf = *reinterpret_cast<const float4*>(Kl + (key0 + ql) * KP + 4 * g4);
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(kf.x, qf.x, s);                          // S^T[key = key0 + 4*g4 + e][q = ql]
                    s = mfma16(kf.y, qf.y, s);
                    s = mfma16(kf.z, qf.z, s);
                    s = mfma16(kf.w, qf.w, s);
                    const float4 mk = *reinterpret_cast<const float4*>(km + key0 + 4 * g4);
                    const int kbase = key0 + 4 * g4;
                    float sv[4] = {s[0], s[1], s[2], s[3]};
                    const float mv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (mv[e] < 0.5f) sv[e] = -32768.0f;            // masked_fill(-2**15), modules.py:220
                        if (kbase + e >= N) sv[e] = -INFINITY;          // padding beyond the sequence: excluded
                    }
                    float tmax = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = expf(m_run - m_new);
                    float p[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[e] = expf(sv[e] - m_new);
                    l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
                    m_run = m_new;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= alpha;
                    const float4 vf = *reinterpret_cast<const float4*>(Vt + ql * (npad + 4) + key0 + 4 * g4);
                    o = mfma16(vf.x, p[0], o);                          // O^T += V^T[c = ql][key] * P^T[key][q]
                    o = mfma16(vf.y, p[1], o);
                    o = mfma16(vf.z, p[2], o);
                    o = mfma16(vf.w, p[3], o);
                }
                float l_tot = l_run + __shfl_xor(l_run, 16);
                l_tot = l_tot + __shfl_xor(l_tot, 32);
                const int v = qb * 32 + qrow;
                if (v < N) {
                    const float4 gf = *reinterpret_cast<const float4*>(Gs + qrow * KP + 4 * g4);
                    const long pos = ending ? (((long)bb * N + v) * N + u) : (bu * N + v);
                    *reinterpret_cast<float4*>(og + pos * HC + h * C + 4 * g4) =
                        make_float4(gf.x * (o[0] / l_tot), gf.y * (o[1] / l_tot), gf.z * (o[2] / l_tot), gf.w * (o[3] / l_tot));
                }
            }
            wave_lds_fence();
        }
    }
}

template <int P>
__global__ __launch_bounds__(WG) void tri_attn_out_kernel(float* out, const float* pair, const float* __restrict__ og,
                                                          const float* __restrict__ wo, const float* __restrict__ bo, long rows, int residual) {
    constexpr int KH = P / 2, NB = P / 32, HC = 64;
    __shared__ __attribute__((aligned(16))) float Wl[P * (HC + 4)];
    __shared__ __attribute__((aligned(16))) float bl[P];
    stage_weight_cll<HC>(Wl, wo, P, HC, threadIdx.x, WG);
    stage_vec_cll(bl, bo, P, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long ntask = (rows + 31) / 32;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float x[HC / 2];
        load_row_cll<HC>(og + pos * HC, hi, valid, x);
        f32x16 acc[NB];
        zero_acc(acc);
        rowgemm<HC, NB>(Wl, x, acc, r, hi);
        float pr[KH];
        load_row_cll<P>(pair + pos * P, hi, valid && residual, pr);
#pragma unroll
        for (int s = 0; s < KH; ++s) pr[s] = pr[s] + (acc[s >> 4][s & 15] + bl[hi * KH + s]);
        store_row_cll<P>(out + pos * P, hi, valid, pr);
    }
}

int grid_for(long tasks, int per_wg, int cap) {
    long g = (tasks + per_wg - 1) / per_wg;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define PRD_SET_LDS(kernel, bytes)                                                                              \
    do {                                                                                                        \
        static size_t prd_lds_set = 0;                                                                          \
        if ((size_t)(bytes) > prd_lds_set) {                                                                    \
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            prd_lds_set = (size_t)(bytes);                                                                      \
        }                                                                                                       \
    } while (0)

extern "C" size_t prd_workspace_bytes(const char* op, int b, int N, int S, int P) {
    (void)S;
    if (!op || b <= 0 || N <= 0) return 0;
    const size_t ldn = (size_t)prd_round_up(N, 32);
    if (op[0] == 't' && op[4] == 'm') return (size_t)3 * b * P * N * ldn * sizeof(float);   // "tri_mul"
    if (op[0] == 't' && op[4] == 'a') return (size_t)b * N * N * 64 * sizeof(float);        // "tri_attn"
    return 0;
}

extern "C" int prd_tri_mul(float* out, const float* pair, const float* mask, const float* w_proj, const float* b_proj,
                           const float* w_gate, const float* b_gate, const float* w_out, const float* b_out,
                           const float* w_ogate, const float* b_ogate, int incoming, int residual,
                           int b, int N, int P, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (!out || !pair || !mask || !w_proj || !b_proj || !w_gate || !b_gate || !w_out || !b_out || !w_ogate || !b_ogate || !ws ||
        b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    if (ws_bytes < prd_workspace_bytes("tri_mul", b, N, 0, P)) return PRD_ERR_WORKSPACE;
    const int ldn = prd_round_up(N, 32);
    float* AB = ws;                                   // [b][2P][N][ldn]
    float* O = ws + (size_t)2 * b * P * N * ldn;      // [b][P][N][ldn]
    {
        const size_t lds = ((size_t)2 * 2 * P * (P + 4) + 4 * P) * sizeof(float);
        const long ntask = (long)b * N * (ldn / 32);
        const int grid = grid_for(ntask, 4, 1024);
        if (P == 64) {
            PRD_SET_LDS(tri_mul_proj_kernel<64>, lds);
            hipLaunchKernelGGL(tri_mul_proj_kernel<64>, dim3(grid), dim3(WG), lds, stream, AB, pair, mask, w_proj, b_proj, w_gate, b_gate, b, N, ldn, incoming);
        } else {
            PRD_SET_LDS(tri_mul_proj_kernel<32>, lds);
            hipLaunchKernelGGL(tri_mul_proj_kernel<32>, dim3(grid), dim3(WG), lds, stream, AB, pair, mask, w_proj, b_proj, w_gate, b_gate, b, N, ldn, incoming);
        }
        int e = (int)hipGetLastError();
        if (e) return e;
    }
    {
        PrdGemm g = {};
        g.A = AB; g.B = AB + (size_t)P * N * ldn; g.C = O;
        g.M = N; g.N = N; g.K = N;
        g.lda = ldn; g.ldb = ldn; g.ldc = ldn;
        g.G1 = b; g.G2 = P;
        g.sa1 = (long long)2 * P * N * ldn; g.sa2 = (long long)N * ldn;
        g.sb1 = g.sa1; g.sb2 = g.sa2;
        g.sc1 = (long long)P * N * ldn; g.sc2 = (long long)N * ldn;
        g.alpha = 1.f;
        int e = prd_gemm(&g, stream);
        if (e) return e;
    }
    {
        const long ntask = (long)b * N * prd_ceil_div(N, 32);
        const int grid = grid_for(ntask, 4, 2048);
        if (P == 64) hipLaunchKernelGGL(tri_mul_out_kernel<64>, dim3(grid), dim3(WG), 0, stream, out, pair, O, w_out, b_out, w_ogate, b_ogate, b, N, ldn, residual);
        else hipLaunchKernelGGL(tri_mul_out_kernel<32>, dim3(grid), dim3(WG), 0, stream, out, pair, O, w_out, b_out, w_ogate, b_ogate, b, N, ldn, residual);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn_core(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                                 const float* wv, const float* wg, const float* bg, int ending,
                                 int b, int N, int P, int H, int c, hipStream_t stream) {
    if (!og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    const int npad = prd_round_up(N, 32);
    const size_t lds_fixed = (size_t)2 * 32 * (P + 4) + (size_t)npad * KP + 16 * (npad + 4) + npad;
    // 8 waves (2 per SIMD) while the row's K/V fit next to 8 query scratch tiles, else 4 waves
    const int nw = ((lds_fixed + 8 * 2 * 32 * KP) * sizeof(float) <= 120 * 1024) ? 8 : 4;
    const size_t lds = (lds_fixed + (size_t)nw * 2 * 32 * KP) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const long ntask = (long)b * N * H;
    const int grid = grid_for(ntask, 1, 4096);
#define PRD_TA_LAUNCH(PP, NW)                                                                                          \
    do {                                                                                                               \
        PRD_SET_LDS((tri_attn_core_kernel<PP, NW>), lds);                                                              \
        hipLaunchKernelGGL((tri_attn_core_kernel<PP, NW>), dim3(grid), dim3(NW * 64), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N, npad, H, ending); \
    } while (0)
    if (P == 64) { if (nw == 8) PRD_TA_LAUNCH(64, 8); else PRD_TA_LAUNCH(64, 4); }
    else { if (nw == 8) PRD_TA_LAUNCH(32, 8); else PRD_TA_LAUNCH(32, 4); }
#undef PRD_TA_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn_out(float* out, const float* pair, const float* og, const float* wo, const float* bo,
                                int residual, int b, int N, int P, hipStream_t stream) {
    if (!out || !pair || !og || !wo || !bo || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    const long rows = (long)b * N * N;
    const int grid2 = grid_for((rows + 31) / 32, 4, 2048);
    if (P == 64) hipLaunchKernelGGL(tri_attn_out_kernel<64>, dim3(grid2), dim3(WG), 0, stream, out, pair, og, wo, bo, rows, residual);
    else hipLaunchKernelGGL(tri_attn_out_kernel<32>, dim3(grid2), dim3(WG), 0, stream, out, pair, og, wo, bo, rows, residual);
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn(float* out, const float* pair, const float* mask, const float* wq, const float* wk, const float* wv,
                            const float* wg, const float* bg, const float* wo, const float* bo, int ending, int residual,
                            int b, int N, int P, int H, int c, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (!ws) return PRD_ERR_ARG;
    if (ws_bytes < prd_workspace_bytes("tri_attn", b, N, 0, P)) return PRD_ERR_WORKSPACE;
    int e = prd_tri_attn_core(ws, pair, mask, wq, wk, wv, wg, bg, ending, b, N, P, H, c, stream);
    if (e) return e;
    return prd_tri_attn_out(out, pair, ws, wo, bo, residual, b, N, P, stream);
}
