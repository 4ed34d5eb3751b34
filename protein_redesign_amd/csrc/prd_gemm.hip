// Generic fp32 building blocks of the single track and of the triangle-multiplication contraction:
//   * prd_gemm:     batched C = epilogue(A * B^T) (or A * B) on v_mfma_f32_32x32x2_f32, LDS tiled
//   * prd_ln_rows:  LayerNorm over the last axis (optional affine)
//   * prd_softmax_rows: in-place row softmax with zero fill of the padded tail
// Replaces the ATen linear / bmm / layer_norm / softmax calls of the reference
// (ProteinReDiff/modules.py:185-225, 306-311; models/AF2_modules.py:251-293, 613-628).
#include "prd_common.h"
#include "../../include/prd_hip.h"
#include <mutex>

namespace {

constexpr int KC = 32;          // K chunk staged in LDS per iteration
constexpr int LDT = KC + 4;     // LDS row pitch (floats): conflict-free ds_read_b128 down a column

// epilogue of one output element (order documented in include/prd_hip.h)
PRD_DEV void epilogue_store(const PrdGemm& g, int g1, int g2, int m, int n, float v, float* __restrict__ C) {
    v = v * g.alpha;
    if (g.colscale) v *= g.colscale[n];
    if (g.bias) v += g.bias[n];
    if (g.addmat) v += g.addmat[g1 * g.sad1 + g2 * g.sad2 + (size_t)m * g.ldadd + n];
    if (g.colmask && g.colmask[g1 * g.scm1 + n] < 0.5f) v = g.fill;
    const int act = (n >= g.act_from) ? g.act : 0;
    if (act == 1) v = relu_nan(v);
    else if (act == 2) v = sigmoidf_(v);
    if (g.rowmask && (g.rowmask_cols <= 0 || n < g.rowmask_cols)) v *= g.rowmask[g1 * g.srm1 + m];
    if (g.mulmat) {
        const float mv = g.mulmat[g1 * g.smu1 + g2 * g.smu2 + (size_t)m * g.ldmul + n];
        v = g.mul_pos ? (mv > 0.f ? v : 0.f) : v * mv;
    }
    if (g.resid) {
        const float rv = g.resid[g1 * g.sr1 + g2 * g.sr2 + (size_t)m * g.ldr + n];
        v += g.rscale ? rv * g.rscale[n] : rv;
    }
    if (g.C2 && n >= g.n_split) g.C2[(size_t)m * g.ldc2 + (n - g.n_split)] = v;
    else C[(size_t)m * g.ldc + n] = v;
}

template <int WM, int WN>       // wave tile (multiples of 32); workgroup = 2 x 2 waves
__global__ __launch_bounds__(256) void gemm_kernel(PrdGemm g) {
    constexpr int TM = 2 * WM, TN = 2 * WN, MI = WM / 32, NI = WN / 32;
    __shared__ __attribute__((aligned(16))) float As[TM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TN * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int tiles_n = (g.N + TN - 1) / TN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int gb = blockIdx.y, g1 = gb / g.G2, g2 = gb - g1 * g.G2;

    const float* __restrict__ A = g.A + g1 * g.sa1 + g2 * g.sa2;
    const float* __restrict__ B = g.B + g1 * g.sb1 + g2 * g.sb2;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

    // Operand staging: thread -> (row = tid>>3 (+32 per rep), 16-byte group f = tid&7).  For B given as [N][K] the global
    // loads of chunk k0+KC are issued right after chunk k0 went to LDS and fly over its MFMAs (unconditional loads from
    // clamped addresses; rows / columns past the edge are zeroed when they are written to LDS).
    constexpr int RA = TM / 32, RB = TN / 32;
    const int srow = tid >> 3, sf = tid & 7;
    float4 pa[RA], pb[RB];
    auto fetch = [&](int k0) {
        const int k = k0 + 4 * sf;
        const int kc = k < g.K ? k : 0;                     // lda / ldb are multiples of 4: a 16-byte load at k < K stays in the row
#pragma unroll
        for (int rep = 0; rep < RA; ++rep) {
            const int m = m0 + srow + 32 * rep;
            pa[rep] = *reinterpret_cast<const float4*>(A + (size_t)(m < g.M ? m : 0) * g.lda + kc);
        }
        if (!g.b_kn) {
#pragma unroll
            for (int rep = 0; rep < RB; ++rep) {
                const int n = n0 + srow + 32 * rep;
                pb[rep] = *reinterpret_cast<const float4*>(B + (size_t)(n < g.N ? n : 0) * g.ldb + kc);
            }
        }
    };
    auto edge = [&](float4 v, bool row_ok, int k) {
        if (!row_ok || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 1 >= g.K) v.y = 0.f;
        if (k + 2 >= g.K) v.z = 0.f;
        if (k + 3 >= g.K) v.w = 0.f;
        return v;
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += KC) {
        {
            const int k = k0 + 4 * sf;
#pragma unroll
            for (int rep = 0; rep < RA; ++rep)
                *reinterpret_cast<float4*>(&As[(srow + 32 * rep) * LDT + 4 * sf]) = edge(pa[rep], m0 + srow + 32 * rep < g.M, k);
            if (!g.b_kn) {
#pragma unroll
                for (int rep = 0; rep < RB; ++rep)
                    *reinterpret_cast<float4*>(&Bs[(srow + 32 * rep) * LDT + 4 * sf]) = edge(pb[rep], n0 + srow + 32 * rep < g.N, k);
            }
        }
        if (g.b_kn) {
            // B is [K][N]: read 16 B along n, scatter transposed into Bs[n][k]
            constexpr int F = TN / 4;                 // float4 per k row of the tile
            for (int idx = tid; idx < KC * F; idx += 256) {
                const int kk = idx / F, f = idx - kk * F;
                const int k = k0 + kk, n = n0 + 4 * f;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K && n < g.N) {
                    v = *reinterpret_cast<const float4*>(B + (size_t)k * g.ldb + n);
                    if (n + 1 >= g.N) v.y = 0.f;
                    if (n + 2 >= g.N) v.z = 0.f;
                    if (n + 3 >= g.N) v.w = 0.f;
                }
                Bs[(4 * f + 0) * LDT + kk] = v.x;
                Bs[(4 * f + 1) * LDT + kk] = v.y;
                Bs[(4 * f + 2) * LDT + kk] = v.z;
                Bs[(4 * f + 3) * LDT + kk] = v.w;
            }
        }
        fetch(k0 + KC < g.K ? k0 + KC : k0);                // next chunk (the last iteration re-reads its own)
        __syncthreads();
        // ---- 16 k-steps of 2: lane (r,hi) feeds k = hi*16 + 4t + e for both operands ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = *reinterpret_cast<const float4*>(&As[(wm0 + 32 * mi + r) * LDT + hi * 16 + 4 * t]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *reinterpret_cast<const float4*>(&Bs[(wn0 + 32 * ni + r) * LDT + hi * 16 + 4 * t]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = mfma32(a[mi].x, b[ni].x, acc[mi][ni]);
                    acc[mi][ni] = mfma32(a[mi].y, b[ni].y, acc[mi][ni]);
                    acc[mi][ni] = mfma32(a[mi].z, b[ni].z, acc[mi][ni]);
                    acc[mi][ni] = mfma32(a[mi].w, b[ni].w, acc[mi][ni]);
                }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds column n = ... + r, rows drow32(q, hi) ----
    float* __restrict__ C = g.C + g1 * g.sc1 + g2 * g.sc2;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn0 + 32 * ni + r;
        if (n >= g.N) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm0 + 32 * mi + drow32(q, hi);
                if (m < g.M) epilogue_store(g, g1, g2, m, n, acc[mi][ni][q], C);
            }
    }
}

// ---- skinny GEMM: one 32x32 output tile per workgroup, the 4 waves split K (in-workgroup split-K),
// operands go global -> registers directly (they are L2 resident: M is a few hundred rows).  For the
// single-track linears (M = b*N rows) the generic kernel above launches fewer workgroups than there
// are CUs and is latency bound on its K loop; this one launches (M/32)*(N/32) workgroups and cuts the
// dependent K chain by 4. ----------------------------------------------------------------------------
template <int NWK, bool LN>                 // waves per workgroup = K splits; LN: fused LayerNorm of the A rows (own
__global__ __launch_bounds__(NWK * 64) void gemm_skinny_kernel(PrdGemm g) {   // instantiation: it holds the K slice in registers)
    __shared__ float red[NWK][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int tiles_n = (g.N + 31) / 32;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int gb = blockIdx.y, g1 = gb / g.G2, g2 = gb - g1 * g.G2;
    const float* __restrict__ A = g.A + g1 * g.sa1 + g2 * g.sa2;
    const float* __restrict__ B = g.B + g1 * g.sb1 + g2 * g.sb2;
    // K range of this wave, in groups of 8
    const int groups = (g.K + 7) / 8;
    const int gper = (groups + NWK - 1) / NWK;
    const int kbeg = wave * gper * 8;
    int kend = kbeg + gper * 8;
    if (kend > g.K) kend = g.K;
    const bool mv = (m0 + r) < g.M, nv = (n0 + r) < g.N;
    const float* arow = A + (size_t)(mv ? m0 + r : 0) * g.lda;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const bool k4 = (g.K & 3) == 0;                      // every 16-byte group inside K is complete
    if (LN) {
        // Fused LayerNorm of the A rows over K (no affine): the lane's K slice (<= LNG 16-byte groups) is loaded ONCE and
        // stays in registers; per-wave (sum, M2 about the wave mean) go through LDS, are merged with the parallel-variance
        // formula (as accurate as the two-pass form) and the MFMA operand is (a - mean) * rstd.  One barrier, no second
        // read of A -- the separate LayerNorm launch costs ~6 us for a 320 x 512 activation, all of it launch latency.
        constexpr int LNG = 16, BT = 4;
        const float* brow = B + (size_t)(nv ? n0 + r : 0) * g.ldb;
        const float am = mv ? 1.f : 0.f, bm = nv ? 1.f : 0.f;
        const int k0 = kbeg + 4 * hi;
        const int nk = (kend > k0) ? (kend - k0 + 7) / 8 : 0;
        const int ksafe = (k0 < g.K) ? k0 : 0;
        float4 av[LNG], cb[BT], nb[BT];
#pragma unroll
        for (int t = 0; t < LNG; ++t) av[t] = *reinterpret_cast<const float4*>(arow + ((t < nk) ? k0 + 8 * t : ksafe));
#pragma unroll
        for (int t = 0; t < BT; ++t) cb[t] = *reinterpret_cast<const float4*>(brow + ((t < nk) ? k0 + 8 * t : ksafe));
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < LNG; ++t) s1 += (t < nk) ? (av[t].x + av[t].y) + (av[t].z + av[t].w) : 0.f;
        s1 += __shfl_xor(s1, 32);
        const float nw = (float)((kend > kbeg) ? kend - kbeg : 0);      // values of this wave's slice (uniform)
        const float mw = nw > 0.f ? s1 / nw : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int t = 0; t < LNG; ++t) {
            const float d0 = av[t].x - mw, d1 = av[t].y - mw, d2 = av[t].z - mw, d3 = av[t].w - mw;
            m2 += (t < nk) ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
        }
        m2 += __shfl_xor(m2, 32);
        red[wave][0][lane] = s1;
        red[wave][1][lane] = m2;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NWK; ++w) tot += red[w][0][lane];
        const float mean = tot / (float)g.K;
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < NWK; ++w) {
            int ke = (w + 1) * gper * 8;
            if (ke > g.K) ke = g.K;
            const float n = (float)((ke > w * gper * 8) ? ke - w * gper * 8 : 0);
            const float dm = (n > 0.f ? red[w][0][lane] / n : mean) - mean;
            var += red[w][1][lane] + n * dm * dm;
        }
        const float rz = am / sqrtf(var / (float)g.K + 1e-5f);
        __syncthreads();                                  // red[] is reused by the K-split reduction below
        if (g.ln_out && tile_n == 0 && mv) {              // the normalised rows themselves (first column tile only)
            float* lo = g.ln_out + (size_t)(m0 + r) * g.ldlo + k0;
#pragma unroll
            for (int t = 0; t < LNG; ++t)
                if (t < nk) *reinterpret_cast<float4*>(lo + 8 * t) = make_float4((av[t].x - mean) * rz, (av[t].y - mean) * rz,
                                                                                 (av[t].z - mean) * rz, (av[t].w - mean) * rz);
        }
#pragma unroll
        for (int bt = 0; bt < LNG / BT; ++bt) {
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int gi = (bt + 1) * BT + t;
                nb[t] = *reinterpret_cast<const float4*>(brow + ((gi < nk) ? k0 + 8 * gi : ksafe));
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const float z = (bt * BT + t < nk) ? rz : 0.f;
                const float4 a = av[bt * BT + t];
                acc = mfma32((a.x - mean) * z, cb[t].x * bm, acc);
                acc = mfma32((a.y - mean) * z, cb[t].y * bm, acc);
                acc = mfma32((a.z - mean) * z, cb[t].z * bm, acc);
                acc = mfma32((a.w - mean) * z, cb[t].w * bm, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < BT; ++t) cb[t] = nb[t];
        }
    } else if (!g.b_kn) {
        const float* brow = B + (size_t)(nv ? n0 + r : 0) * g.ldb;
        if (k4) {
            const float am = mv ? 1.f : 0.f, bm = nv ? 1.f : 0.f;     // rows past the edge contribute zeros
            // software pipeline over batches of BT k-groups: the 2*BT loads of batch t+1 are in flight while the 4*BT
            // MFMAs of batch t execute.  The prefetch is UNCONDITIONAL (clamped indices; groups past the end are
            // multiplied by zero): loads under an `if` would be waited for with vmcnt(0) in front of the MFMAs.
            constexpr int BT = 4;
            float4 ca[BT], cb[BT], na[BT], nb[BT];
            const int k0 = kbeg + 4 * hi;
            const int nk = (kend > k0) ? (kend - k0 + 7) / 8 : 0;      // k-groups of this lane's half (k4: all complete)
            const int nkw = (kend > kbeg) ? (kend - kbeg + 7) / 8 : 0; // k-groups of the wave (uniform trip count)
            const int nbatch = (nkw + BT - 1) / BT;
            const int ksafe = (k0 < g.K) ? k0 : 0;
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int k = (t < nk) ? k0 + 8 * t : ksafe;
                ca[t] = *reinterpret_cast<const float4*>(arow + k);
                cb[t] = *reinterpret_cast<const float4*>(brow + k);
            }
            for (int bt = 0; bt < nbatch; ++bt) {
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const int gi = (bt + 1) * BT + t;
                    const int k = (gi < nk) ? k0 + 8 * gi : ksafe;
                    na[t] = *reinterpret_cast<const float4*>(arow + k);
                    nb[t] = *reinterpret_cast<const float4*>(brow + k);
                }
#pragma unroll
                for (int t = 0; t < BT; ++t) {
                    const float z = (bt * BT + t < nk) ? am : 0.f;     // groups past this lane's range add nothing
                    acc = mfma32(ca[t].x * z, cb[t].x * bm, acc);
                    acc = mfma32(ca[t].y * z, cb[t].y * bm, acc);
                    acc = mfma32(ca[t].z * z, cb[t].z * bm, acc);
                    acc = mfma32(ca[t].w * z, cb[t].w * bm, acc);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < BT; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
            }
        } else {
            for (int k = kbeg + 4 * hi; k < kend; k += 8) {
                float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
                for (int e = 0; e < 4; ++e)
                    if (k + e < g.K) { if (mv) ta[e] = arow[k + e]; if (nv) tb[e] = brow[k + e]; }
                acc = mfma32(ta[0], tb[0], acc);
                acc = mfma32(ta[1], tb[1], acc);
                acc = mfma32(ta[2], tb[2], acc);
                acc = mfma32(ta[3], tb[3], acc);
            }
        }
    } else {
        const float* bcol = B + (nv ? n0 + r : 0);
        for (int k = kbeg + 4 * hi; k < kend; k += 8) {
            float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < g.K) { if (mv) ta[e] = arow[k + e]; if (nv) tb[e] = bcol[(size_t)(k + e) * g.ldb]; }
            acc = mfma32(ta[0], tb[0], acc);
            acc = mfma32(ta[1], tb[1], acc);
            acc = mfma32(ta[2], tb[2], acc);
            acc = mfma32(ta[3], tb[3], acc);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave][q][lane] = acc[q];
    __syncthreads();
    // wave w finishes registers q = (16/NWK)*w ... of the tile: fixed summation order over the K splits
    float* __restrict__ C = g.C + g1 * g.sc1 + g2 * g.sc2;
    const int n = n0 + r;
    constexpr int QPW = 16 / NWK;
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq) {
        const int q = QPW * wave + qq;
        float v = red[0][q][lane];
#pragma unroll
        for (int w = 1; w < NWK; ++w) v += red[w][q][lane];
        const int m = m0 + drow32(q, hi);
        if (m < g.M && n < g.N) epilogue_store(g, g1, g2, m, n, v, C);
    }
}

// ---- single-track GEMM on the fp16 matrix pipe (gemm mode 1) ---------------------------------------------------------------
// C = epilogue(A B^T) for the node-row linears (M = b N rows, a few hundred): both operands are split into fp16 hi + lo while
// they are staged into LDS (prd_common.h: split2h; weights x 16), three products per K step on v_mfma_f32_32x32x16_f16.
// What bounded the fp32 skinny kernel was not the MFMA but operand traffic and latency: 32x32 tiles re-load every A row N/32
// times (82 MB of L2 reads for the 512 -> 2048 linear) through row-per-lane loads (64 cache lines per instruction).  Here a
// workgroup owns a 64x64 tile (2x2 waves of 32x32), streams K in 32-wide chunks through double-buffered LDS with coalesced
// 128-byte row segments, and -- for long K -- KG groups of four waves take every KG-th chunk (split-K inside the workgroup,
// merged in LDS in a fixed order).  Optional fused LayerNorm of the A rows (statistics in a prologue pass).
template <int KG>
__global__ __launch_bounds__(256 * KG) void gemm_h2_kernel(PrdGemm g) {
    constexpr int PLANE = 64 * 64;                      // bytes of one operand plane of a 64-row, 32-wide chunk
    constexpr int BUF = 4 * PLANE;                      // A hi | A lo | B hi | B lo
    extern __shared__ __attribute__((aligned(16))) unsigned char gh[];      // [KG][2 buffers][BUF] + [64] mean + [64] rstd
    float* mean_l = reinterpret_cast<float*>(gh + KG * 2 * BUF);
    float* rstd_l = mean_l + 64;
    const int tid = threadIdx.x, kg = tid >> 8, t8 = tid & 255, lane = tid & 63, wave = (tid >> 6) & 3;
    const int r = lane & 31, hi = lane >> 5;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int tiles_n = (g.N + 63) / 64;
    int tile_m, tile_n;
    if (g.tile_hint == -8) {
        // (host: PrdGemm.tile_hint is re-used as the launch's block map; -8 = XCD-aware columns)  All row tiles of ONE column tile on
        // ONE XCD, next to each other in dispatch order: block b runs on XCD b % 8 (observed placement, speed only), so the weight tile
        // of a column crosses the fabric once and the other row tiles find it in that XCD's L2.  Row-major order (block = tile_m *
        // tiles_n + tile_n) put the five row tiles of a column of the trunk-head projection (M = 320, N = 8448, 17 MB of weights) on
        // different XCDs and 132 blocks apart: 54.6 MB of HBM-side traffic for 17 MB of weights (profiles/r04_roofline.txt).
        const int tiles_m = (g.M + 63) / 64, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile_m = j % tiles_m;
        tile_n = (j / tiles_m) * 8 + xcd;
        if (tile_n >= tiles_n) return;                  // (grid rounded up to a multiple of 8 column tiles; uniform per workgroup)
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
    }
    const int m0 = tile_m * 64, n0 = tile_n * 64;
    const int gb = blockIdx.y, g1 = gb / g.G2, g2 = gb - g1 * g.G2;
    const float* __restrict__ A = g.A + g1 * g.sa1 + g2 * g.sa2;
    const float* __restrict__ B = g.B + g1 * g.sb1 + g2 * g.sb2;
    if (g.a_ln) {                                       // per-row LayerNorm statistics of the 64 A rows (4 threads per row)
        if (tid < 256) {
            const int row = tid >> 2, part = tid & 3, m = m0 + row;
            const float* ar = A + (size_t)(m < g.M ? m : 0) * g.lda;
            // ONE pass, sixteen 16-byte loads in flight per thread: sums of (x - p) and (x - p)^2 about a pivot p = the row's first
            // element (a sample of the row: |mean - p| is of the order of the spread, so the shifted form loses nothing even for
            // rows with |mean| >> spread).  A load-at-a-time two-pass loop is a chain of 2 K / 16 L2 round trips: 8 us at K = 512.
            constexpr int UB = 16;
            const float pv = ar[0];
            float s1 = 0.f, s2 = 0.f;
            for (int k0 = 4 * part; k0 < g.K; k0 += 16 * UB) {
                float4 v[UB];
#pragma unroll
                for (int j = 0; j < UB; ++j) v[j] = *reinterpret_cast<const float4*>(ar + (k0 + 16 * j < g.K ? k0 + 16 * j : 0));
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const float d0 = v[j].x - pv, d1 = v[j].y - pv, d2 = v[j].z - pv, d3 = v[j].w - pv;
                    const bool in = k0 + 16 * j < g.K;
                    s1 += in ? (d0 + d1) + (d2 + d3) : 0.f;
                    s2 += in ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
                }
            }
            s1 += __shfl_xor(s1, 1);
            s1 += __shfl_xor(s1, 2);
            const float dm = s1 / (float)g.K;                   // mean - p
            const float mu = pv + dm;
            s2 += __shfl_xor(s2, 1);
            s2 += __shfl_xor(s2, 2);
            if (part == 0) { mean_l[row] = mu; rstd_l[row] = 1.0f / sqrtf(fmaxf(s2 / (float)g.K - dm * dm, 0.f) + 1e-5f); }
        }
        __syncthreads();
    }
    // staging of a K-group: 64 A rows + 64 B rows x 8 pieces of 4 floats per chunk = 1024 pieces, 4 per thread (2 A, 2 B)
    const prd_rsrc ra = make_rsrc(A + (size_t)m0 * g.lda), rb = make_rsrc(B + (size_t)n0 * g.ldb);
    unsigned off[4], dst[4];
    float amu[2], ars[2];
    float* lo_[2] = {nullptr, nullptr};                 // PrdGemm.ln_out: the normalised A pieces this thread stages (first column tile only)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = (t8 + 256 * i) & 511, row = p >> 3, f = p & 7;
        const bool isb = i >= 2;
        const bool ok = isb ? (n0 + row < g.N) : (m0 + row < g.M);
        if (!isb && ok && g.a_ln && g.ln_out && tile_n == 0) lo_[i] = g.ln_out + (size_t)(m0 + row) * g.ldlo + 4 * f;
        off[i] = ok ? ((unsigned)row * (isb ? g.ldb : g.lda) + 4 * f) * 4u : BUF_OOB;
        dst[i] = (isb ? 2 * PLANE : 0) + row * 64 + (((f >> 1) ^ ((row >> 2) & 3)) << 4) + (f & 1) * 8;
        if (!isb) { amu[i] = g.a_ln ? mean_l[row] : 0.f; ars[i] = g.a_ln ? (ok ? rstd_l[row] : 0.f) : 1.f; }
    }
    unsigned char* mybuf = gh + kg * 2 * BUF;
    const unsigned swz = (unsigned)((r >> 2) & 3);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int nchunk = g.K / 32;                        // K is a multiple of 32 on this path
    const int myn = (nchunk - kg + KG - 1) / KG;        // chunks kg, kg + KG, ... of this K-group
    u32x4 u[4], v[4];
#define PRD_GH_LOAD(R, CI)                                                                                          \
    {                                                                                                               \
        const int c_ = kg + KG * ((CI) < myn ? (CI) : (myn > 0 ? myn - 1 : 0));                                     \
        R[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off[0], c_ * 128, 0));           \
        R[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off[1], c_ * 128, 0));           \
        R[2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off[2], c_ * 128, 0));           \
        R[3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off[3], c_ * 128, 0));           \
    }
#define PRD_GH_STAGE(R, BUFI, SC_)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
        float x0 = __uint_as_float(R[i][0]), x1 = __uint_as_float(R[i][1]), x2 = __uint_as_float(R[i][2]), x3 = __uint_as_float(R[i][3]); \
        if (i < 2) { x0 = (x0 - amu[i]) * ars[i]; x1 = (x1 - amu[i]) * ars[i]; x2 = (x2 - amu[i]) * ars[i]; x3 = (x3 - amu[i]) * ars[i];   \
                     if (lo_[i]) *reinterpret_cast<float4*>(lo_[i] + (size_t)(SC_) * 32) = make_float4(x0, x1, x2, x3); }               \
        else { x0 *= H2_WSCALE; x1 *= H2_WSCALE; x2 *= H2_WSCALE; x3 *= H2_WSCALE; }                                \
        unsigned h0, l0, h1, l1;                                                                                    \
        split2h(x0, x1, h0, l0);                                                                                    \
        split2h(x2, x3, h1, l1);                                                                                    \
        unsigned char* d_ = mybuf + (BUFI) * BUF + dst[i];                                                          \
        *reinterpret_cast<u32x2*>(d_) = u32x2{h0, h1};                                                              \
        *reinterpret_cast<u32x2*>(d_ + PLANE) = u32x2{l0, l1};                                                      \
    }
#define PRD_GH_CHUNK(CI, CUR, R, NX)                                                                                \
    {                                                                                                               \
        PRD_GH_LOAD(NX, (CI) + 2)                                                                                   \
        if ((CI) < myn) {                                                                                           \
            const unsigned char* base = mybuf + (CUR) * BUF;                                                        \
            _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                      \
                const unsigned col = (((unsigned)(2 * st + hi)) ^ swz) << 4;                                        \
                const unsigned char* ap = base + (wm0 + r) * 64 + col;                                              \
                const unsigned char* bp = base + 2 * PLANE + (wn0 + r) * 64 + col;                                  \
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), al = *reinterpret_cast<const u32x4*>(ap + PLANE); \
                const u32x4 bh = *reinterpret_cast<const u32x4*>(bp), bl = *reinterpret_cast<const u32x4*>(bp + PLANE); \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), acc, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0); \
            }                                                                                                       \
        }                                                                                                           \
        if ((CI) + 1 < myn) { PRD_GH_STAGE(R, (CUR) ^ 1, kg + KG * ((CI) + 1)) }                                                        \
        __syncthreads();                                                                                            \
    }
    const int maxn = (nchunk + KG - 1) / KG;            // trip count of the longest K-group: barriers are workgroup-wide
    PRD_GH_LOAD(u, 0)
    if (myn > 0) { PRD_GH_STAGE(u, 0, kg) }
    PRD_GH_LOAD(u, 1)
    __syncthreads();
    for (int ci = 0; ci < maxn; ci += 2) {
        PRD_GH_CHUNK(ci, 0, u, v)
        if (ci + 1 < maxn) PRD_GH_CHUNK(ci + 1, 1, v, u)
    }
#undef PRD_GH_LOAD
#undef PRD_GH_STAGE
#undef PRD_GH_CHUNK
    if (KG > 1) {                                       // merge the K-groups in LDS (over the staging buffers), fixed order
        float* part = reinterpret_cast<float*>(gh);     // [KG - 1][256 threads][17]
        if (kg > 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) part[((kg - 1) * 256 + t8) * 17 + q] = acc[q];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int k2 = 0; k2 < KG - 1; ++k2)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += part[(k2 * 256 + t8) * 17 + q];
        }
    }
    if (kg == 0) {
        float* __restrict__ C = g.C + g1 * g.sc1 + g2 * g.sc2;
        const int n = n0 + wn0 + r;
        if (n < g.N) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm0 + drow32(q, hi);
                if (m < g.M) epilogue_store(g, g1, g2, m, n, acc[q] * H2_INV_WSCALE, C);
            }
        }
    }
}

// ---- node-row linears of the single track in gemm mode 1: 32x32 tiles, deep operand ring ---------------------------------
// What bounds the M = b N (a few hundred) linears is neither the matrix pipe nor the launch (1.7 us between dependent
// kernels in a graph), but operand delivery (tools/ubench/nodegemm_bench.hip, profiles/r02_nodegemm_ubench.txt):
//   * a lane-per-row load touches 32 cache lines per instruction and the CU's L1 tag pipe saturates (TCP_TOTAL_CACHE_ACCESSES
//     = one per cycle for the whole launch of gemm_skinny_kernel);
//   * a CU streams operands at ~50 GB/s however many loads it has in flight, and a chunked loop with two chunks in flight pays
//     the L2-miss latency once per chunk.
// Here eight lanes read one 128-byte line of a row (4x fewer tag accesses), a group of four waves keeps D chunks of 64 k in
// flight in registers (16 KB each), splits fp16 hi | lo while staging a chunk into LDS in the MFMA operand layout (16-byte
// columns XOR-swizzled by row >> 1: a row is 128 bytes, two rows share a 256-byte bank line, so the sixteen rows of a ds_read_b128
// group are (row & 1) x (slot ^ (row >> 1) & 7) = sixteen different 16-byte slots, and the extra ^ 4 (row & 1) puts the two rows of
// a 16-lane ds_write_b64 group on different 64-byte halves of the 128-byte write bank line; round 3 XORed with row & 7, which put
// rows r and r + 8 on one read slot: 45-50 % of the kernel's LDS cycles were conflicts), and -- for long K -- KG groups take every KG-th chunk with their own LDS stages.  The four
// waves of a group each take one 16-wide k-step of a chunk; partial tiles are merged in LDS in a fixed order.  LayerNorm
// statistics come from the ring itself (the whole row is in flight: K <= 64 D KG), so the first load is the only exposed one.
template <int D, int KG, bool LN, bool BKN = false>      // BKN: B is given as [K][N] (row pitch ldb) -- staged transposed, 2 bytes at a time
__global__ __launch_bounds__(256 * KG) void gemm_ring_kernel(PrdGemm g) {
    constexpr int KCH = 64, NJ = KCH / 32, PL = 32 * KCH * 2, STAGE = 4 * PL;      // plane = 32 rows x 64 fp16; A hi | A lo | B hi | B lo
    extern __shared__ __attribute__((aligned(16))) unsigned char gr[];             // [KG][2][STAGE] (the partial tiles alias it) + LN sums
    const int tid = threadIdx.x, kg = tid >> 8, t8 = tid & 255, lane = tid & 63, wave = (tid >> 6) & 3;
    const int r = lane & 31, hi = lane >> 5;
    const int tiles_n = (g.N + 31) / 32;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int gb = blockIdx.y, g1 = gb / g.G2, g2 = gb - g1 * g.G2;
    const int row = t8 >> 3, seg = t8 & 7;
    // rows past the edge read row 0 (their outputs are never stored)
    const float* ap = g.A + g1 * g.sa1 + g2 * g.sa2 + (size_t)(m0 + row < g.M ? m0 + row : 0) * g.lda + 4 * seg;
    // BKN: thread (row, seg) takes k = row and row + 32 of a chunk, columns n0 + 4 seg .. + 3 (clamped inside the matrix: the
    // columns past N are never stored)
    const int bn4 = n0 + 4 * seg + 3 < g.N ? n0 + 4 * seg : (g.N >= 4 ? g.N - 4 : 0);
    const float* bp = BKN ? g.B + g1 * g.sb1 + g2 * g.sb2 + (size_t)row * g.ldb + bn4
                          : g.B + g1 * g.sb1 + g2 * g.sb2 + (size_t)(n0 + row < g.N ? n0 + row : 0) * g.ldb + 4 * seg;
    const int nch = g.K / KCH;
    const int myn = (nch - kg + KG - 1) / KG;           // chunks kg, kg + KG, ... of this group (>= 1: the host picks KG <= nch)
    unsigned char* mysm = gr + kg * 2 * STAGE;
    // PrdGemm.a_scale: an exact power of two on the A operand while it is split (small operands -- probabilities -- would lose
    // their lo part to the fp16 subnormal range), taken back out of the accumulator
    const float asc = (g.a_scale > 0.f && !LN) ? g.a_scale : 1.0f, inv_asc = 1.0f / asc;
    float ascr = asc;                                   // per-row factor of the A operand (a_ln = 2: includes 1 / softmax denominator)
    float4 ra[D][NJ], rb[D][NJ];
#define PRD_GR_LOAD(SLOT, CI)                                                                            \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                     \
        const int c_ = kg + KG * ((CI) < myn ? (CI) : myn - 1);                                          \
        ra[SLOT][j] = *reinterpret_cast<const float4*>(ap + c_ * KCH + 32 * j);                          \
        rb[SLOT][j] = BKN ? *reinterpret_cast<const float4*>(bp + (size_t)(c_ * KCH + 32 * j) * g.ldb)   \
                          : *reinterpret_cast<const float4*>(bp + c_ * KCH + 32 * j);                    \
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { PRD_GR_LOAD(d, d) }
    float mean = 0.f, rz = 1.f;
    if (LN) {                                           // two-pass statistics over the row values held in the ring (myn <= D)
        float* st = reinterpret_cast<float*>(gr + KG * 2 * STAGE);      // [KG][32 rows][2]
        float s1 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) s1 += d < myn ? (ra[d][j].x + ra[d][j].y) + (ra[d][j].z + ra[d][j].w) : 0.f;
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
        if (KG > 1) {
            if (seg == 0) st[(kg * 32 + row) * 2] = s1;
            __syncthreads();
            s1 = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < KG; ++k2) s1 += st[(k2 * 32 + row) * 2];
        }
        mean = s1 / (float)g.K;
        float s2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float d0 = ra[d][j].x - mean, d1 = ra[d][j].y - mean, d2 = ra[d][j].z - mean, d3 = ra[d][j].w - mean;
                s2 += d < myn ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
            }
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
        if (KG > 1) {
            if (seg == 0) st[(kg * 32 + row) * 2 + 1] = s2;
            __syncthreads();
            s2 = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < KG; ++k2) s2 += st[(k2 * 32 + row) * 2 + 1];
        }
        rz = 1.0f / sqrtf(s2 / (float)g.K + 1e-5f);
    }
    if (BKN && g.a_ln == 2) {
        // PrdGemm.a_ln = 2: row SOFTMAX of the A rows over K (the whole row is in the ring: K <= 64 D) -- SPAttention's softmax
        // rides in its P V product (models/AF2_modules.py:613-628) instead of a launch of its own that rewrites the logits
        constexpr float L2E = 1.4426950408889634f;
        float m = -INFINITY;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (d < myn) m = fmaxf(m, fmaxf(fmaxf(ra[d][j].x, ra[d][j].y), fmaxf(ra[d][j].z, ra[d][j].w)));
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        float ssum = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ra[d][j].x = __builtin_amdgcn_exp2f((ra[d][j].x - m) * L2E); ra[d][j].y = __builtin_amdgcn_exp2f((ra[d][j].y - m) * L2E);
                ra[d][j].z = __builtin_amdgcn_exp2f((ra[d][j].z - m) * L2E); ra[d][j].w = __builtin_amdgcn_exp2f((ra[d][j].w - m) * L2E);
                ssum += d < myn ? (ra[d][j].x + ra[d][j].y) + (ra[d][j].z + ra[d][j].w) : 0.f;
            }
        ssum += __shfl_xor(ssum, 1); ssum += __shfl_xor(ssum, 2); ssum += __shfl_xor(ssum, 4);
        ascr = asc / ssum;
    }
#define PRD_GR_STAGE(SLOT, ST, CI)                                                                       \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                     \
        float4 a = ra[SLOT][j];                                                                          \
        const float4 b = rb[SLOT][j];                                                                    \
        if (!LN) { a.x *= ascr; a.y *= ascr; a.z *= ascr; a.w *= ascr; }                                 \
        if (LN) {                                                                                        \
            a.x = (a.x - mean) * rz; a.y = (a.y - mean) * rz; a.z = (a.z - mean) * rz; a.w = (a.w - mean) * rz; \
            if (lo) *reinterpret_cast<float4*>(lo + (kg + KG * (CI)) * KCH + 32 * j) = a;               \
        }                                                                                                \
        unsigned h0, l0, h1, l1;                                                                         \
        unsigned char* d_ = mysm + (ST) * STAGE + row * (KCH * 2) + (((4 * j + (seg >> 1)) ^ (((row >> 1) & 7) ^ ((row & 1) << 2))) << 4) + (seg & 1) * 8; \
        split2h(a.x, a.y, h0, l0); split2h(a.z, a.w, h1, l1);                                            \
        *reinterpret_cast<u32x2*>(d_) = u32x2{h0, h1};                                                   \
        *reinterpret_cast<u32x2*>(d_ + PL) = u32x2{l0, l1};                                              \
        split2h(H2_WSCALE * b.x, H2_WSCALE * b.y, h0, l0); split2h(H2_WSCALE * b.z, H2_WSCALE * b.w, h1, l1); \
        if (!BKN) {                                                                                      \
            *reinterpret_cast<u32x2*>(d_ + 2 * PL) = u32x2{h0, h1};                                      \
            *reinterpret_cast<u32x2*>(d_ + 3 * PL) = u32x2{l0, l1};                                      \
        } else {          /* element e of the piece belongs to B row (column of the matrix) bn4 - n0 + e, k = row + 32 j */ \
            const unsigned hh[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16}, ll[4] = {l0 & 0xffffu, l0 >> 16, l1 & 0xffffu, l1 >> 16}; \
            const int kl = row + 32 * j;                                                                 \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                              \
                const int nl = bn4 - n0 + e;                                                             \
                if (nl >= 0 && nl < 32) {                                                                \
                    unsigned char* t_ = mysm + (ST) * STAGE + 2 * PL + nl * (KCH * 2) + ((((kl >> 3)) ^ (((nl >> 1) & 7) ^ ((nl & 1) << 2))) << 4) + (kl & 7) * 2; \
                    *reinterpret_cast<unsigned short*>(t_) = (unsigned short)hh[e];                      \
                    *reinterpret_cast<unsigned short*>(t_ + PL) = (unsigned short)ll[e];                 \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }
    // the normalised rows themselves, written by the first column tile (PrdGemm.ln_out)
    float* lo = (LN && g.ln_out && tile_n == 0 && m0 + row < g.M) ? g.ln_out + (size_t)(m0 + row) * g.ldlo + 4 * seg : nullptr;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#define PRD_GR_MFMA(ST)                                                                                  \
    {                                                                                                    \
        const int col = 2 * wave + hi;                   /* wave w takes k-step w of the chunk */        \
        const unsigned char* a_ = mysm + (ST) * STAGE + r * (KCH * 2) + ((col ^ (((r >> 1) & 7) ^ ((r & 1) << 2))) << 4);         \
        const u32x4 ah = *reinterpret_cast<const u32x4*>(a_), al = *reinterpret_cast<const u32x4*>(a_ + PL); \
        const u32x4 bh = *reinterpret_cast<const u32x4*>(a_ + 2 * PL), bl = *reinterpret_cast<const u32x4*>(a_ + 3 * PL); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0); \
    }
    // chunk c: stage it (LDS stage c & 1 was last read by MFMA(c - 2), which every wave finished before the barrier of
    // chunk c - 1), refill its ring slot with chunk c + D, barrier, multiply.  One barrier per chunk.
    const int maxn = (nch + KG - 1) / KG;               // trip count of the longest group: the barriers are workgroup-wide
    for (int c0 = 0; c0 < maxn; c0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int c = c0 + d;
            if (c < maxn) {
                if (c < myn) { PRD_GR_STAGE(d, d & 1, c) }
                PRD_GR_LOAD(d, c + D)
                __syncthreads();
                if (c < myn) PRD_GR_MFMA(d & 1)
            }
        }
    }
#undef PRD_GR_LOAD
#undef PRD_GR_STAGE
#undef PRD_GR_MFMA
    __syncthreads();
    float* red = reinterpret_cast<float*>(gr);          // [KG * 4 waves][16][64]
#pragma unroll
    for (int q = 0; q < 16; ++q) red[((kg * 4 + wave) * 16 + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (kg == 0) {
        float* __restrict__ C = g.C + g1 * g.sc1 + g2 * g.sc2;
        const int n = n0 + r;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = 4 * wave + qq;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4 * KG; ++w) v += red[(w * 16 + q) * 64 + lane];
            const int m = m0 + drow32(q, hi);
            if (m < g.M && n < g.N) epilogue_store(g, g1, g2, m, n, v * (H2_INV_WSCALE * inv_asc), C);
        }
    }
}

template <int D, int KG>
static int launch_ring(const PrdGemm& g, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)KG * 2 * 4 * 32 * 64 * 2 + 1024;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<D, KG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<D, KG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (g.a_ln) hipLaunchKernelGGL((gemm_ring_kernel<D, KG, true>), grid, dim3(256 * KG), lds, stream, g);
    else hipLaunchKernelGGL((gemm_ring_kernel<D, KG, false>), grid, dim3(256 * KG), lds, stream, g);
    return (int)hipGetLastError();
}

static int launch_ring_bkn(const PrdGemm& g, dim3 grid, hipStream_t stream) {      // B as [K][N]: P V of SPAttention (K = keys)
    const size_t lds = (size_t)2 * 4 * 32 * 64 * 2 + 1024;
    hipLaunchKernelGGL((gemm_ring_kernel<8, 1, false, true>), grid, dim3(256), lds, stream, g);
    return (int)hipGetLastError();
}

// ---- node-row linears with LARGE weights (the single-track transition 512 -> 2048 -> 512): K split ACROSS workgroups ----------
// What bounds these GEMMs at M = b N = a few hundred rows is the operand stream per CU (~50 GB/s per CU whatever is in flight:
// tools/ubench/nodegemm_bench.hip; the MI355X guide's M = 256 projection GEMM finds the same 21 B/clk), so the lever is BYTES PER
// CU: 32 x 32 tiles move (1/32 + 1/32) x 4 K M N bytes in total (82 MB for 320 x 512 x 2048: 320 KB per CU, and the 2048 -> 512
// layer has only 160 tiles of 512 KB each).  Here a workgroup owns a 160 x 64 tile of ONE 128-wide K slab: 112 KB of operands,
// all requested up front, 256 workgroups = one per CU for both layers of the transition (64 tiles x 4 slabs / 16 tiles x 16
// slabs), 29 MB in total.  Operands are split into fp16 hi | lo while they are staged into LDS (rows of 256 B per plane, 16-byte
// slots XOR-swizzled by row: conflict-free fragment reads), ten waves take one 32 x 32 sub-tile each (24 MFMAs), and the fp32
// partial tiles go to a workspace [slab][M][N]; gemm_slab_reduce_kernel sums the slabs in a fixed order and applies the epilogue.
// (A reduction inside the launch -- last-arriver per tile -- costs a release + acquire fence pair and serialises 120 KB of reads
// on one CU: the guide's `splitk-seam` row prices it above the kernel boundary it would save.)
constexpr int SL_BM = 160, SL_BN = 64, SL_KS = 128, SL_NW = 10;
constexpr int SL_LDS = (SL_BM + SL_BN) * SL_KS * 4;                  // hi | lo planes of A and W: 112 KB
__global__ __launch_bounds__(SL_NW * 64) void gemm_slab_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                               float* __restrict__ ws, int M, int N, int K, int lda, int ldb,
                                                               int tiles_m, int tiles_n, int nslab, int spw, int xmap, float wscale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sl[];      // A hi [160][256 B] | A lo | W hi [64][256 B] | W lo
    constexpr int APL = SL_BM * 256, WOFF = 2 * APL, WPL = SL_BN * 256;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 1, ct = wave & 1;
    // Block -> (tile, slab group): every operand slab is shared -- the A slab (row block, slabs) by all column tiles, the W slab
    // (column tile, slabs) by all row blocks -- so the workgroups of ONE slab group go to ONE XCD (block b is observed on XCD
    // b % 8; speed only, any placement is correct): each slab then crosses the fabric once and is served from that XCD's L2.
    // With fewer than 8 slab groups the column tiles of a group are cut into 8 / SK parts instead.
    const int SK = (nslab + spw - 1) / spw;
    int tile_m, tile_n, sg;
    if (xmap == 1) {                    // SK % 8 == 0
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, tiles = tiles_m * tiles_n;
        sg = xcd + 8 * (j / tiles);
        const int t_ = j % tiles;
        tile_m = t_ % tiles_m; tile_n = t_ / tiles_m;
    } else if (xmap == 2) {             // 8 % SK == 0 and tiles_n % (8 / SK) == 0
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, tn_per = tiles_n / (8 / SK);
        sg = xcd % SK;
        tile_m = j % tiles_m; tile_n = (xcd / SK) * tn_per + j / tiles_m;
    } else {
        const int ncombo = tiles_m * SK, combo = blockIdx.x % ncombo;
        tile_n = blockIdx.x / ncombo; tile_m = combo % tiles_m; sg = combo / tiles_m;
    }
    const int m0 = tile_m * SL_BM, n0 = tile_n * SL_BN;
    const int prow = tid >> 5, piece = tid & 31;                            // 32 threads per row: 512 contiguous bytes
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int s_end = (sg + 1) * spw < nslab ? (sg + 1) * spw : nslab;
    for (int sl_i = sg * spw; sl_i < s_end; ++sl_i) {
        const prd_rsrc ra = make_rsrc(A + (size_t)m0 * lda + (size_t)sl_i * SL_KS), rb = make_rsrc(W + (size_t)n0 * ldb + (size_t)sl_i * SL_KS);
        u32x4 va[8], vb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = prow + 20 * i;
            va[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (m0 + row < M) ? ((unsigned)row * lda + 4 * piece) * 4u : BUF_OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = prow + 20 * i;
            vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (row < SL_BN && n0 + row < N) ? ((unsigned)row * ldb + 4 * piece) * 4u : BUF_OOB, 0, 0));
        }
        if (sl_i > sg * spw) __syncthreads();                               // the previous slab's fragments have been read
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = prow + 20 * i;
            unsigned h0, l0, h1, l1;
            split2h(__uint_as_float(va[i][0]), __uint_as_float(va[i][1]), h0, l0);
            split2h(__uint_as_float(va[i][2]), __uint_as_float(va[i][3]), h1, l1);
            unsigned char* d_ = sl + row * 256 + (((piece >> 1) ^ (row & 15)) << 4) + (piece & 1) * 8;
            *reinterpret_cast<u32x2*>(d_) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d_ + APL) = u32x2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = prow + 20 * i;
            if (row < SL_BN) {
                unsigned h0, l0, h1, l1;
                split2h(wscale * __uint_as_float(vb[i][0]), wscale * __uint_as_float(vb[i][1]), h0, l0);
                split2h(wscale * __uint_as_float(vb[i][2]), wscale * __uint_as_float(vb[i][3]), h1, l1);
                unsigned char* d_ = sl + WOFF + row * 256 + (((piece >> 1) ^ (row & 15)) << 4) + (piece & 1) * 8;
                *reinterpret_cast<u32x2*>(d_) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(d_ + WPL) = u32x2{l0, l1};
            }
        }
        __syncthreads();
        const unsigned char* ab = sl + (rt * 32 + r) * 256;
        const unsigned char* bb = sl + WOFF + (ct * 32 + r) * 256;
#pragma unroll
        for (int st = 0; st < SL_KS / 16; ++st) {
            const unsigned so = (unsigned)((2 * st + hi) ^ (r & 15)) << 4;   // (rt * 32 + r) & 15 == r & 15
            const u32x4 ah = *reinterpret_cast<const u32x4*>(ab + so), al = *reinterpret_cast<const u32x4*>(ab + APL + so);
            const u32x4 bh = *reinterpret_cast<const u32x4*>(bb + so), bl = *reinterpret_cast<const u32x4*>(bb + WPL + so);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), acc, 0, 0, 0);
        }
    }
    // partial tile: lane holds column n of 16 rows; 32 lanes = 128 contiguous bytes of a workspace row per store
    const int n = n0 + ct * 32 + r;
    if (n < N) {
        float* wp = ws + (size_t)sg * M * N + n;
        const float inv = 1.0f / wscale;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + rt * 32 + drow32(q, hi);
            if (m < M) wp[(size_t)m * N] = acc[q] * inv;
        }
    }
}

// Sum of the K slabs + epilogue of gemm_slab_kernel: one workgroup of 128 threads per (row, 512-column chunk), 16 bytes per
// thread and slab, all slab loads in flight together, summed in slab order.  With a_ln the GEMM ran on the RAW rows and the
// LayerNorm is applied here by linearity: LN(x) W^T = rstd (x W^T - mean colsum(W)) -- the workgroup computes the statistics
// of its row of A itself (two passes over 4 K bytes) and `wsum` = row sums of W comes from the caller.  With out_ln (N <= 512:
// the workgroup holds the whole output row) the LayerNorm of the OUTPUT row is written as well, for the next linear.
__global__ __launch_bounds__(128) void gemm_slab_reduce_kernel(PrdGemm g, const float* __restrict__ ws, int SK) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunks = (g.N + 511) / 512;
    const int m = blockIdx.x / chunks, n = (blockIdx.x - m * chunks) * 512 + 4 * tid;
    const bool live = n < g.N;                                              // N is a multiple of 4 on this path
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = ws + (size_t)m * g.N + (live ? n : 0);
    const size_t sstride = (size_t)g.M * g.N;
    for (int s0 = 0; s0 < SK; s0 += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(wp + (size_t)(s0 + j < SK ? s0 + j : 0) * sstride);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s0 + j < SK) { sum.x += v[j].x; sum.y += v[j].y; sum.z += v[j].z; sum.w += v[j].w; }
    }
    auto block_sum = [&](float x) {                                         // over the 128 threads, same value in every thread
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        return red[0] + red[1];
    };
    float vv[4] = {sum.x, sum.y, sum.z, sum.w};
    if (g.a_ln) {                                                           // LayerNorm of the A row by linearity
        const float* ar = g.A + (size_t)m * g.lda;
        float s1 = 0.f;
        for (int k = 4 * tid; k < g.K; k += 512) { const float4 a = *reinterpret_cast<const float4*>(ar + k); s1 += (a.x + a.y) + (a.z + a.w); }
        const float mu = block_sum(s1) / (float)g.K;
        float s2 = 0.f;
        for (int k = 4 * tid; k < g.K; k += 512) {
            const float4 a = *reinterpret_cast<const float4*>(ar + k);
            const float d0 = a.x - mu, d1 = a.y - mu, d2 = a.z - mu, d3 = a.w - mu;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float rz = 1.0f / sqrtf(block_sum(s2) / (float)g.K + 1e-5f);
        if (g.ln_out && blockIdx.x - m * chunks == 0)                        // the row's first chunk also writes the normalised row
            for (int k = 4 * tid; k < g.K; k += 512) {
                const float4 a = *reinterpret_cast<const float4*>(ar + k);
                *reinterpret_cast<float4*>(g.ln_out + (size_t)m * g.ldlo + k) = make_float4((a.x - mu) * rz, (a.y - mu) * rz, (a.z - mu) * rz, (a.w - mu) * rz);
            }
        if (live) {
            const float4 cs = *reinterpret_cast<const float4*>(g.wsum + n);
            vv[0] = (vv[0] - mu * cs.x) * rz; vv[1] = (vv[1] - mu * cs.y) * rz; vv[2] = (vv[2] - mu * cs.z) * rz; vv[3] = (vv[3] - mu * cs.w) * rz;
        }
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = vv[e] * g.alpha;
            if (g.colscale) v *= g.colscale[n + e];
            if (g.bias) v += g.bias[n + e];
            const int act = (n + e >= g.act_from) ? g.act : 0;
            if (act == 1) v = relu_nan(v);
            else if (act == 2) v = sigmoidf_(v);
            if (g.rowmask && (g.rowmask_cols <= 0 || n + e < g.rowmask_cols)) v *= g.rowmask[m];
            if (g.resid) { const float rv = g.resid[(size_t)m * g.ldr + n + e]; v += g.rscale ? rv * g.rscale[n + e] : rv; }
            vv[e] = v;
        }
        // (a second output block: n_split is a multiple of 4 on this path, a 16-byte piece never straddles it)
        if (g.C2 && n >= g.n_split) *reinterpret_cast<float4*>(g.C2 + (size_t)m * g.ldc2 + (n - g.n_split)) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        else *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + n) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    if (g.out_ln) {                                                         // N <= 512: this workgroup holds the whole output row
        const float mu = block_sum(live ? (vv[0] + vv[1]) + (vv[2] + vv[3]) : 0.f) / (float)g.N;
        const float d0 = vv[0] - mu, d1 = vv[1] - mu, d2 = vv[2] - mu, d3 = vv[3] - mu;
        const float rz = 1.0f / sqrtf(block_sum(live ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f) / (float)g.N + 1e-5f);
        if (live) *reinterpret_cast<float4*>(g.out_ln + (size_t)m * g.ldol + n) = make_float4(d0 * rz, d1 * rz, d2 * rz, d3 * rz);
    }
}

// ---- LayerNorm rows: one wave per row ------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int rows, int C, int ldx, int ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / C + 1e-5f);
    float* yr = y + (size_t)row * ldy;
    for (int c = lane; c < C; c += 64) {
        float t = (xr[c] - mean) * rstd;
        if (gamma) t = t * gamma[c] + beta[c];
        yr[c] = t;
    }
}

// rows of 64 channels (the pair track): 16 lanes x float4 per row, four rows per wave, reductions inside a DPP row
__global__ __launch_bounds__(256) void ln_rows64_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int ldx, int ldy) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), c4 = 4 * (threadIdx.x & 15);
    if (row >= rows) return;
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c4);
    const float mean = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64);
    const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
    const float rstd = 1.0f / sqrtf(row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 64) + 1e-5f);
    float4 t = make_float4(d0 * rstd, d1 * rstd, d2 * rstd, d3 * rstd);
    if (gamma) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c4), bt = *reinterpret_cast<const float4*>(beta + c4);
        t = make_float4(t.x * g.x + bt.x, t.y * g.y + bt.y, t.z * g.z + bt.z, t.w * g.w + bt.w);
    }
    *reinterpret_cast<float4*>(y + (size_t)row * ldy + c4) = t;
}

// ---- softmax rows (in place), one wave per row, zero-fills [n, ld) ---------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int rows, int n, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, xr[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += expf(xr[c] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    for (int c = lane; c < ld; c += 64) xr[c] = (c < n) ? expf(xr[c] - m) / s : 0.f;
}

// does the K-slab path serve this shape, and with which split?  (one predicate for prd_gemm and prd_gemm_slab_ok)
bool slab_plan(int M, int N, int K, int arith_full, int* tiles_m, int* tiles_n, int* nslab, int* spw) {
    if (arith_full < 0 || (arith_full & 0xff) != PRD_ARITH_SPLIT16 || ((arith_full >> 8) & (1 << 16))) return false;
    if (M < 96 || N <= 0 || K <= 0 || (K % SL_KS) || (N % 4)) return false;
    *tiles_m = prd_ceil_div(M, SL_BM);
    *tiles_n = prd_ceil_div(N, SL_BN);
    *nslab = K / SL_KS;
    const long tiles = (long)*tiles_m * *tiles_n;
    if (tiles * *nslab < 128 || tiles > 4096) return false;        // too little work for 256 CUs / too many tiles: other kernels
    int s = 1;                                                     // slabs per workgroup: at most ~2 workgroups per CU
    while (tiles * prd_ceil_div(*nslab, s) > 512 && s < *nslab) ++s;
    *spw = s;
    return true;
}
}  // namespace

extern "C" int prd_gemm_slab_ok(int M, int N, int K, int arith) {
    int a, b, c, d;
    return slab_plan(M, N, K, arith, &a, &b, &c, &d) ? 1 : 0;
}

extern "C" int prd_gemm(const PrdGemm* args, hipStream_t stream) {
    const PrdGemm& g = *args;
    if (!g.A || !g.B || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0 || g.G1 <= 0 || g.G2 <= 0) return PRD_ERR_ARG;
    if (g.arith < 0 || (g.arith & 0xff) > 1) return PRD_ERR_ARG;
    const int arith = g.arith & 0xff;                  // upper bits: PRD_TUNE_* switches (none applies to the GEMMs)
    if ((g.lda & 3) || (g.ldb & 3)) return PRD_ERR_ALIGN;
    const int batches = g.G1 * g.G2;
    if (g.ln_out && (!g.a_ln || batches != 1 || (g.ldlo & 3) || g.ldlo < g.K)) return PRD_ERR_ARG;
    if (g.C2 && (batches != 1 || g.n_split <= 0 || g.n_split >= g.N || g.ldc2 < g.N - g.n_split)) return PRD_ERR_ARG;
    const long tiles64 = (long)prd_ceil_div(g.M, 64) * prd_ceil_div(g.N, 64) * batches;
    // gemm mode 1, large weights on few rows (the transition layers): K split across workgroups + reduce / epilogue launch
    if (g.ws && g.tile_hint == 0 && !g.b_kn && batches == 1 && !g.addmat && !g.colmask && !g.mulmat &&
        (!g.C2 || ((g.n_split & 3) == 0 && (g.ldc2 & 3) == 0 && !g.out_ln)) && (!g.ln_out || (g.wsum && (g.K & 3) == 0)) &&
        (!g.a_ln || g.wsum) && (!g.out_ln || (g.N <= 512 && (g.ldol & 3) == 0)) && (g.ldc & 3) == 0 && (!g.resid || (g.ldr & 3) == 0)) {
        int tiles_m, tiles_n, nslab, spw;
        if (slab_plan(g.M, g.N, g.K, g.arith, &tiles_m, &tiles_n, &nslab, &spw)) {
            const int SK = prd_ceil_div(nslab, spw);
            if ((size_t)SK * g.M * g.N * sizeof(float) > g.ws_bytes) return PRD_ERR_WORKSPACE;
            static std::once_flag once;
            std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)gemm_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
            const int xmap = (SK % 8 == 0) ? 1 : ((SK < 8 && 8 % SK == 0 && tiles_n % (8 / SK) == 0) ? 2 : 0);
            hipLaunchKernelGGL(gemm_slab_kernel, dim3((unsigned)(tiles_m * tiles_n * SK)), dim3(SL_NW * 64), SL_LDS, stream, g.A, g.B, g.ws, g.M, g.N,
                               g.K, g.lda, g.ldb, tiles_m, tiles_n, nslab, spw, xmap, H2_WSCALE);
            hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)(g.M * prd_ceil_div(g.N, 512))), dim3(128), 0, stream, g, g.ws, SK);
            return (int)hipGetLastError();
        }
    }
    if (g.out_ln) return PRD_ERR_UNSUPPORTED;           // the LayerNorm of the output rows exists on the slab path only (prd_gemm_slab_ok)
    // gemm mode 1: THROUGHPUT-bound linears (>= 512 tiles of 64x64: SPAttention's 512 -> 4 x 2048 projection, the transition at
    // b = 8) go to the 64x64-tile fp16 x 2 kernel (41 -> 26 us); the latency-bound ones (20-160 tiles) to gemm_ring_kernel below.
    if (arith == PRD_ARITH_SPLIT16 && g.tile_hint == 0 && !g.b_kn && (g.K % 32) == 0 && tiles64 >= 512 && g.G1 * g.G2 == 1 &&
        (!g.a_ln || (g.K % 16) == 0)) {
        dim3 grid(prd_ceil_div(g.M, 64) * prd_ceil_div(g.N, 64), batches);
        PrdGemm gx = g;
        // more weight bytes than row bytes and at least 8 column tiles: the XCD-aware column map (see the kernel).  OPT-IN (PRD_TUNE bit
        // 18, PRD_GEMM_XCDCOLS=1): measured in round 5 on the trunk-head projection, same box, one replayed step each: 33.8 us with the
        // map, 33.4 us row-major -- the launch is bound by its per-chunk load -> split -> LDS -> barrier chain, not by HBM traffic
        if (g.N >= 512 && (long)g.N > 2L * g.M && ((g.arith >> 8) & (1 << 18))) {
            gx.tile_hint = -8;
            grid.x = 8u * prd_ceil_div(g.M, 64) * prd_ceil_div(prd_ceil_div(g.N, 64), 8);
        }
        static std::once_flag once1, once4;
        if (g.K >= 1024) {
            const size_t lds = (size_t)4 * 2 * 4 * 64 * 64 + 512;
            std::call_once(once4, [] { (void)hipFuncSetAttribute((const void*)gemm_h2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
            hipLaunchKernelGGL((gemm_h2_kernel<4>), grid, dim3(1024), lds, stream, gx);
        } else {
            const size_t lds = (size_t)2 * 4 * 64 * 64 + 512;
            std::call_once(once1, [] { (void)hipFuncSetAttribute((const void*)gemm_h2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
            hipLaunchKernelGGL((gemm_h2_kernel<1>), grid, dim3(256), lds, stream, gx);
        }
        return (int)hipGetLastError();
    }
    // gemm mode 1, latency-bound node-row linears (fewer 64x64 tiles than that): 32x32 tiles with a deep operand ring
    // (batched, e.g. the per-head logits / P V of SPAttention: same kernel, blockIdx.y = batch; without the fused LayerNorm)
    if (arith == PRD_ARITH_SPLIT16 && g.tile_hint == 0 && g.b_kn && (g.K % 64) == 0 && g.K <= 512 && tiles64 < 2048 && (g.a_ln == 0 || g.a_ln == 2) &&
        g.N >= 4 && (g.ldb & 3) == 0 && !((g.arith >> 8) & (1 << 17))) {
        dim3 grid(prd_ceil_div(g.M, 32) * prd_ceil_div(g.N, 32), batches);
        return launch_ring_bkn(g, grid, stream);
    }
    if (g.a_ln == 2) return PRD_ERR_UNSUPPORTED;        // the fused row softmax exists in that kernel only
    if (arith == PRD_ARITH_SPLIT16 && g.tile_hint == 0 && !g.b_kn && (g.K % 64) == 0 && tiles64 < 512 &&
        (batches == 1 || (!g.a_ln && !((g.arith >> 8) & (1 << 17))))) {
        const int nch = g.K / 64;
        dim3 grid(prd_ceil_div(g.M, 32) * prd_ceil_div(g.N, 32), batches);
        const bool many = (long)grid.x * grid.y > 1024;         // throughput regime: one group per workgroup, more workgroups per CU
        // few tiles (the 512 -> 256 / 64 projections: 80 / 20 tiles on 256 CUs): the chunk loop is a serial chain of
        // stage -> barrier -> MFMA steps per workgroup, so split it over two or four wave groups (PRD_TUNE_GEMM_NO_KG: A/B switch)
        const long ntile = (long)grid.x * grid.y;
        const bool kgsplit = !((g.arith >> 8) & (1 << 15));
        if (kgsplit && nch == 8 && ntile <= 64) return launch_ring<2, 4>(g, grid, stream);
        if (kgsplit && nch == 8 && ntile <= 192) return launch_ring<4, 2>(g, grid, stream);
        if (kgsplit && nch == 4 && ntile <= 128) return launch_ring<2, 2>(g, grid, stream);
        if (nch <= 1) return launch_ring<1, 1>(g, grid, stream);
        if (nch <= 2) return launch_ring<2, 1>(g, grid, stream);
        if (nch <= 4) return launch_ring<4, 1>(g, grid, stream);
        if (nch <= 8) return launch_ring<8, 1>(g, grid, stream);
        if (!g.a_ln) {
            if (many) return launch_ring<4, 1>(g, grid, stream);
            if (nch <= 16) return launch_ring<8, 2>(g, grid, stream);
            return launch_ring<4, 4>(g, grid, stream);
        }
        if (nch <= 16) return launch_ring<8, 2>(g, grid, stream);   // LayerNorm needs the whole row in the ring: K <= 1024 here
    }
    int tile = g.tile_hint;
    if (g.a_ln) {                                   // fused LayerNorm lives in the K-split kernel only; the K slice of a lane
        // (K / splits / 2 values) has to fit its 16 register groups: K <= 512 with 4 splits
        if (g.b_kn || (g.K & 3) || g.K > 512 || (tile != 0 && tile != 32)) return PRD_ERR_UNSUPPORTED;
        tile = 32;
    }
    if (tile == 0) tile = (tiles64 < 512) ? 32 : ((tiles64 <= 1024 || g.M <= 64 || g.N <= 64) ? 64 : 128);
    if (tile == 32) {          // fewer 64x64 tiles than two per CU: skinny kernel, 32x32 tiles + in-workgroup split-K
        dim3 grid(prd_ceil_div(g.M, 32) * prd_ceil_div(g.N, 32), batches);
        // long K and few tiles: 8 K-splits (twice the waves per CU to cover the L2 latency of the operand stream)
        if (g.a_ln) hipLaunchKernelGGL((gemm_skinny_kernel<4, true>), grid, dim3(256), 0, stream, g);
        else if (g.K >= 1024 && (long)grid.x * grid.y <= 512) hipLaunchKernelGGL((gemm_skinny_kernel<8, false>), grid, dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((gemm_skinny_kernel<4, false>), grid, dim3(256), 0, stream, g);
    } else if (tile == 64) {
        dim3 grid(prd_ceil_div(g.M, 64) * prd_ceil_div(g.N, 64), batches);
        hipLaunchKernelGGL((gemm_kernel<32, 32>), grid, dim3(256), 0, stream, g);
    } else {
        dim3 grid(prd_ceil_div(g.M, 128) * prd_ceil_div(g.N, 128), batches);
        hipLaunchKernelGGL((gemm_kernel<64, 64>), grid, dim3(256), 0, stream, g);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_ln_rows(const float* x, float* y, const float* gamma, const float* beta,
                           int rows, int C, int ldx, int ldy, hipStream_t stream) {
    if (!x || !y || rows <= 0 || C <= 0 || (gamma && !beta)) return PRD_ERR_ARG;
    const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                        reinterpret_cast<uintptr_t>(beta)) & 15) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0;
    if (C == 64 && al16 && rows >= 4096) {              // the pair-track calls of the training path
        hipLaunchKernelGGL(ln_rows64_kernel, dim3(prd_ceil_div(rows, 16)), dim3(256), 0, stream, x, y, gamma, beta, rows, ldx, ldy);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(ln_rows_kernel, dim3(prd_ceil_div(rows, 4)), dim3(256), 0, stream, x, y, gamma, beta, rows, C, ldx, ldy);
    return (int)hipGetLastError();
}

extern "C" int prd_softmax_rows(float* x, int rows, int n, int ld, hipStream_t stream) {
    if (!x || rows <= 0 || n <= 0 || ld < n) return PRD_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(prd_ceil_div(rows, 4)), dim3(256), 0, stream, x, rows, n, ld);
    return (int)hipGetLastError();
}

extern "C" size_t prd_gemm_slab_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % SL_KS)) return 0;
    return (size_t)(K / SL_KS) * M * N * sizeof(float);
}
