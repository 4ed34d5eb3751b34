// Triangle attention core, second generation (split-16 arithmetic, rows of up to 352 positions).
//
// Replaces the reference's TriangleAttention -> Attention.forward chain (modules.py:236-243 -> 185-225) up to the gated
// per-head output `og`; the output projection stays in tri_attn_out / pair_tail (prd_tri.hip, prd_pair.hip).
//
// One PERSISTENT workgroup (8 waves) per CU serves one head for a strided set of pair rows, like the first-generation
// kernel (tri_attn_core_split_kernel), but everything inside a row is laid out for the 32x32x16 fp16 MFMA:
//
//   phase 1  per 32-position block of the row, two half-items dealt to the waves:
//            [K|Q]: unswapped row GEMM (A = weights, B = the lane's LayerNorm-ed row, fp16 x 2): lane (pos, hi) ends up with
//                   the 8 K channels {4hi+e, 8+4hi+e} and the same 8 Q channels of ITS position -- exactly the 8 contraction
//                   values an A / B operand lane of the QK^T MFMA holds.  K and Q go to LDS as fp16 hi | lo planes with one
//                   16-byte store per plane; no transposition, no three-way split.
//            [V|G]: SWAPPED row GEMM (A = the rows, B = weights): lane (channel, hi) ends up with 16 positions of ONE
//                   channel, which is the key-major order the P V MFMA wants from its A operand: V hi | lo planes are
//                   stored with 16-byte stores (the first generation scattered 2-byte values), the gate (lanes 16-31)
//                   goes to a [channel][position] fp32 tile.
//   phase 2  S^T = K Q^T for 32 keys x 32 queries per MFMA triple (kh qh + kh ql + kl qh; the contraction is the head width
//            16 = the K of the instruction, nothing is padded), so a lane holds 16 logits of ONE query: softmax statistics
//            are lane-local plus one cross-half exchange.  The probabilities, split into fp16 hi | lo, are directly the B
//            operand of O^T = [V_hi; V_lo] P^T (M = 32 = both planes of the 16 channels: hi*hi, lo*hi, hi*lo, lo*lo in two
//            MFMAs per 16 keys), because V was stored in the key order of the S^T register layout.
//            The (query block, key tile) iterations of a row are cut into 8 CONTIGUOUS ranges, one per wave ("stream-K"):
//            every wave gets the same number of tiles +-1 whatever N is; a wave's range crosses at most two query-block
//            boundaries, every piece leaves a partial (reference, sum, O) in LDS and the partials of a query block are
//            merged after one barrier (flash-decoding merge), gated and stored.
//            The reference maximum of a piece is fixed by its first tile (later tiles: accumulator preloaded with
//            -reference, one v_exp_f32 per logit); should a probability leave the fp16 range the piece is redone with the
//            online update in every tile.  QK^T of tile t+1 is issued before the softmax arithmetic of tile t.
//
// Arithmetic: operands hi = RN_fp16(x), lo = RN_fp16(x - hi) (v_cvt_pk_f16_f32 + v_fma_mix): |x - hi - lo| <= 2^-24 |x|
// while lo is a normal fp16 number, an absolute 2^-25 below that; products accumulate in fp32.  Probabilities are kept
// x 2^4 relative to the reference maximum so that small probabilities keep a normal lo part.
#include "prd_common.h"
#include "../../include/prd_hip.h"
#include <mutex>

#ifdef PRD_TIMING     // diagnostic builds only (tools/ta2_timing.py): cycle stamps [workgroup][8 waves][8 rows][8 stamps]
__device__ unsigned long long prd_dbg2[256 * 8 * 8 * 8];
extern "C" int prd_debug_read2(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prd_dbg2), sizeof(prd_dbg2)); }
#define PRD2_STAMP(k) do { if (lane == 0 && it < 8) prd_dbg2[((blockIdx.x * 8 + wave) * 8 + it) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PRD2_STAMP(k)
#endif

namespace {

constexpr float LOG2E_2 = 1.4426950408889634f;
constexpr float P_SHIFT = 4.0f;                 // probabilities are 2^(s - max + P_SHIFT)
constexpr int V2_MAXN = 352;

PRD_DEV f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// hi = RN_fp16 of the pair (a, b), lo = RN_fp16(x - hi) -- one more bit than the RTZ form of prd_common.h (the residual is
// signed), same three instructions per pair
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
PRD_DEV void split2h_rn(float a, float b, unsigned& hi, unsigned& lo) {
    // the conversion is left to the compiler (it selects v_cvt_pk_f16_f32): as the FIRST reader of a value that may come
    // straight out of an MFMA or a transcendental it must be an instruction whose hazards hipcc pads (an asm statement's
    // reads are not padded)
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
// 8 consecutive registers of an MFMA fragment -> hi / lo operand registers (element jj in half-word jj)
PRD_DEV void split8_rn(const f32x16& v, int base, u32x4& h, u32x4& l) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned a, b;
        split2h_rn(v[base + 2 * w], v[base + 2 * w + 1], a, b);
        h[w] = a;
        l[w] = b;
    }
}
template <int NE>
PRD_DEV void split2h_rn_cll(const float (&x)[NE], u32x4 (&p)[2][NE / 8]) {
#pragma unroll
    for (int s = 0; s < NE / 8; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2h_rn(x[8 * s + 2 * q], x[8 * s + 2 * q + 1], h, l);
            p[0][s][q] = h;
            p[1][s][q] = l;
        }
}

// acc += X * W^T for 32 positions x the 32 image rows row0..row0+31 (SWAPPED operands: lane (n, hi) register j = output
// channel row0 + n of position drow32(j, hi)); image as staged by stage_weight_h2_rows
template <int K>
PRD_DEV void rowgemm_h2_swapped(const u32x4* Wh, int nout, int row0, const u32x4 (&p)[2][K / 16], f32x16& acc, int r, int hi) {
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
        const int o = row0 + r;
        const int slot = h2_slot<K>(o, 2 * s + hi);
        const u32x4 wh = Wh[(size_t)o * (K / 8) + slot], wl = Wh[(size_t)(nout + o) * (K / 8) + slot];
        acc = mfma_h(p[0][s], wh, acc);
        acc = mfma_h(p[1][s], wh, acc);
        acc = mfma_h(p[0][s], wl, acc);
    }
}

PRD_DEV float xhalf_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max2f(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
PRD_DEV float xhalf_add(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
PRD_DEV float max16(const f32x16& s) {
    float m = max3f(s[0], s[1], s[2]);
    m = max3f(m, s[3], s[4]);
    m = max3f(m, s[5], s[6]);
    m = max3f(m, s[7], s[8]);
    m = max3f(m, s[9], s[10]);
    m = max3f(m, s[11], s[12]);
    m = max3f(m, s[13], s[14]);
    return max2f(m, s[15]);
}

struct V2Lds {                  // byte offsets from the dynamic LDS base
    unsigned kh, kl, qh, ql, v, g, kadd, flag, bias, part;
    unsigned plane;             // bytes between the hi = 0 and hi = 1 halves of a K / Q plane
};

PRD_DEV V2Lds v2_layout(int P, int NP) {
    V2Lds L;
    unsigned off = 64u * P * 4u;                       // weight image: fp16 hi | lo planes of 64 rows [K | Q | V | G]
    L.plane = (unsigned)NP * 16u;
    L.kh = off; off += 2 * L.plane;
    L.kl = off; off += 2 * L.plane;
    L.qh = off; off += 2 * L.plane;
    L.ql = off; off += 2 * L.plane;
    L.v = off; off += (unsigned)NP * 64u;              // [tile][a][hi][m = plane * 16 + c][8 fp16]
    L.g = off; off += (unsigned)NP * 64u;              // [c][NP] fp32
    L.kadd = off; off += (unsigned)NP * 4u;
    L.flag = off; off += 64u;
    L.bias = off; off += 64u;
    L.part = off;                                      // [slot][10][64] fp32
    return L;
}

// S^T tile: 32 keys x 32 queries; C = cinit (all registers)
PRD_DEV f32x16 qk_tile(const unsigned char* lds, const V2Lds& L, int T, int r, int hi, u32x4 qh, u32x4 ql, const f32x16& cinit) {
    const unsigned ko = (unsigned)hi * L.plane + (unsigned)(32 * T + r) * 16u;
    const u32x4 kh = *reinterpret_cast<const u32x4*>(lds + L.kh + ko);
    const u32x4 kl = *reinterpret_cast<const u32x4*>(lds + L.kl + ko);
    f32x16 s = mfma_h(kh, qh, cinit);
    s = mfma_h(kh, ql, s);
    s = mfma_h(kl, qh, s);
    return s;
}

// logit override of masked / padded keys of tile T (absolute values in the exp2 domain; 0 = keep the logit)
PRD_DEV void mask_tile(const unsigned char* lds, const V2Lds& L, int T, int hi, float mref, f32x16& s) {
    const float* kadd = reinterpret_cast<const float*>(lds + L.kadd) + 32 * T + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 ka = *reinterpret_cast<const float4*>(kadd + 8 * g);
        s[4 * g + 0] = (ka.x == 0.f) ? s[4 * g + 0] : ka.x - mref;
        s[4 * g + 1] = (ka.y == 0.f) ? s[4 * g + 1] : ka.y - mref;
        s[4 * g + 2] = (ka.z == 0.f) ? s[4 * g + 2] : ka.z - mref;
        s[4 * g + 3] = (ka.w == 0.f) ? s[4 * g + 3] : ka.w - mref;
    }
}

// p = 2^s (s already relative to the reference), row-sum, split, O += [V_hi; V_lo] P
PRD_DEV void exp_pv_tile(const unsigned char* lds, const V2Lds& L, int T, int r, int hi, f32x16& s, float& lsum, bool& big,
                         f32x16& o0, f32x16& o1) {
    const unsigned vo = L.v + (unsigned)(T * 4 + hi) * 512u + (unsigned)r * 16u;
    const u32x4 va0 = *reinterpret_cast<const u32x4*>(lds + vo);
    const u32x4 va1 = *reinterpret_cast<const u32x4*>(lds + vo + 1024u);
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        s[j] = __builtin_amdgcn_exp2f(s[j]);
        s[j + 1] = __builtin_amdgcn_exp2f(s[j + 1]);
        t0 += s[j];
        t1 += s[j + 1];
    }
    const float ts = t0 + t1;
    big |= !(ts < 30000.0f);                           // a probability near the fp16 range (or inf / NaN)
    lsum += ts;
    u32x4 ph0, pl0, ph1, pl1;
    split8_rn(s, 0, ph0, pl0);
    split8_rn(s, 8, ph1, pl1);
    o0 = mfma_h(va0, ph0, o0);
    o1 = mfma_h(va1, ph1, o1);
    o0 = mfma_h(va0, pl0, o0);
    o1 = mfma_h(va1, pl1, o1);
}

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_v2_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const V2Lds L = v2_layout(P, NP);
    u32x4* Wb = reinterpret_cast<u32x4*>(lds);
    float* Gl = reinterpret_cast<float*>(lds + L.g);
    float* kadd = reinterpret_cast<float*>(lds + L.kadd);
    int* tflag = reinterpret_cast<int*>(lds + L.flag);
    float* biasl = reinterpret_cast<float*>(lds + L.bias);
    float* part = reinterpret_cast<float*>(lds + L.part);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int nqb = NP / 32;                            // query blocks = key tiles
    const int rstride = gridDim.x / H;
    int h, slot;
    if ((rstride & 7) == 0) {                           // the H heads of one row on one XCD
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        h = idx % H;
        slot = (idx / H) * 8 + xcd;
    } else {
        h = blockIdx.x % H;
        slot = blockIdx.x / H;
    }
    const float sc = 0.25f * LOG2E_2;
    stage_weight_h2_rows<P>(Wb, 64, 0, wk + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, C, wq + (long)h * C * P, C, P, tid, NT, sc * H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 2 * C, wv + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 3 * C, wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E * H2_WSCALE);
    if (tid < 16) biasl[tid] = H2_WSCALE * NEG_LOG2E * bg[h * C + tid];
    const long nrows = (long)b * N;
    auto row_pos = [&](long bu, int v) -> long {
        const long bb = bu / N;
        const long u = bu - bb * N;
        return ending ? ((bb * N + v) * N + u) : (bu * N + v);
    };
    // ---- static work split ----
    // phase 1: half-items i = 2 * block + kind dealt round-robin (kind 0 = [K|Q], 1 = [V|G])
    const int nitem = 2 * nqb;
    // phase 2: iterations (query block, key tile) cut into NW contiguous ranges
    const int niter = nqb * nqb;
    const int it_begin = (int)((long)niter * wave / NW), it_end = (int)((long)niter * (wave + 1) / NW);
    const float inv16 = H2_INV_WSCALE;

    float xnext[KH];                                    // the wave's first phase-1 block of the NEXT row
    {
        const int blk = wave >> 1;
        const int v = blk * 32 + r;
        const bool ok = slot < nrows && wave < nitem && v < N;
        load_row_cll<P>(pair + row_pos(ok ? slot : 0, ok ? v : 0) * P, hi, ok, xnext);
    }
    int it = 0;
    for (long bu = slot; bu < nrows; bu += rstride, ++it) {
        const int bb = (int)(bu / N);
        __syncthreads();                                // previous row's LDS consumed (and the weight image staged)
        PRD2_STAMP(0);
        const float mu = mask[bu];
        // ================= phase 1 =================
        {
            // one half-item: LayerNorm + split of the block's rows (in place), one row GEMM, stores
            auto do_item = [&](int item, float (&x)[KH]) {
                const int blk = item >> 1, kind = item & 1;
                const int v = blk * 32 + r;
                const bool valid = v < N;
                ln_cll<KH>(x);
                u32x4 xs[2][P / 16];
                split2h_rn_cll<KH>(x, xs);
                if (kind == 0) {
                    {
                        const bool keep = valid && (mu * mask[(long)bb * N + (valid ? v : 0)] >= 0.5f);
                        if (hi == 0) kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E_2 : -INFINITY);
                        const bool any_override = __any(!keep);
                        if (lane == 0) tflag[blk] = any_override ? 1 : 0;
                    }
                    f32x16 acc[1];
                    zero_acc(acc);
                    rowgemm_h2<P, 1>(Wb, 64, 0, xs, acc, r, hi);
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][e] *= inv16;
                    u32x4 kh4, kl4, qh4, ql4;
                    split8_rn(acc[0], 0, kh4, kl4);
                    split8_rn(acc[0], 8, qh4, ql4);
                    const unsigned po = (unsigned)hi * L.plane + (unsigned)v * 16u;
                    *reinterpret_cast<u32x4*>(lds + L.kh + po) = kh4;
                    *reinterpret_cast<u32x4*>(lds + L.kl + po) = kl4;
                    *reinterpret_cast<u32x4*>(lds + L.qh + po) = qh4;
                    *reinterpret_cast<u32x4*>(lds + L.ql + po) = ql4;
                } else {
                    f32x16 acc;
                    {
                        const float b0 = r >= 16 ? biasl[r - 16] : 0.f;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = b0;
                    }
                    rowgemm_h2_swapped<P>(Wb, 64, 32, xs, acc, r, hi);
                    if (r < 16) {                       // V channel r (x 16): hi | lo planes of positions drow32(j, hi)
                        u32x4 vh0, vl0, vh1, vl1;
                        split8_rn(acc, 0, vh0, vl0);
                        split8_rn(acc, 8, vh1, vl1);
                        const unsigned vo = L.v + (unsigned)(blk * 4 + hi) * 512u;
                        *reinterpret_cast<u32x4*>(lds + vo + (unsigned)r * 16u) = vh0;
                        *reinterpret_cast<u32x4*>(lds + vo + (unsigned)(16 + r) * 16u) = vl0;
                        *reinterpret_cast<u32x4*>(lds + vo + 1024u + (unsigned)r * 16u) = vh1;
                        *reinterpret_cast<u32x4*>(lds + vo + 1024u + (unsigned)(16 + r) * 16u) = vl1;
                    } else {                            // gate channel r - 16
                        float* gp = Gl + (r - 16) * NP + blk * 32 + 4 * hi;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<float4*>(gp + 8 * g) =
                                make_float4(gate_from_scaled(acc[4 * g] * inv16), gate_from_scaled(acc[4 * g + 1] * inv16),
                                            gate_from_scaled(acc[4 * g + 2] * inv16), gate_from_scaled(acc[4 * g + 3] * inv16));
                    }
                }
            };
            auto fetch = [&](int item, float (&x)[KH]) {
                const int v = (item >> 1) * 32 + r;
                const bool ok = item < nitem && v < N;
                load_row_cll<P>(pair + row_pos(bu, ok ? v : 0) * P, hi, ok, x);
            };
            // the rows of item k + 1 are requested before item k is computed (item 0 came in during the previous key loops)
            float xb[KH];
            fetch(wave + NW, xb);
            if (wave < nitem) do_item(wave, xnext);
            if (wave + NW < nitem) {
                fetch(wave + 2 * NW, xnext);
                do_item(wave + NW, xb);
                if (wave + 2 * NW < nitem) do_item(wave + 2 * NW, xnext);
            }
        }
        PRD2_STAMP(1);
        __syncthreads();
        PRD2_STAMP(2);
        {   // the wave's first block of the next row: in flight during the key loops
            const long bun = bu + rstride;
            const int v = (wave >> 1) * 32 + r;
            const bool ok = bun < nrows && wave < nitem && v < N;
            load_row_cll<P>(pair + row_pos(ok ? bun : 0, ok ? v : 0) * P, hi, ok, xnext);
        }
        // ================= phase 2 =================
        unsigned long long fmask;
        {
            const int f = lane < nqb ? tflag[lane] : 0;
            fmask = __ballot(f != 0);
        }
        for (int i0 = it_begin; i0 < it_end;) {
            const int qb = i0 / nqb;
            const int T0 = i0 - qb * nqb;
            const int T1 = (it_end - qb * nqb) < nqb ? (it_end - qb * nqb) : nqb;       // exclusive
            i0 += T1 - T0;
            const unsigned qo = (unsigned)hi * L.plane + (unsigned)(32 * qb + r) * 16u;
            const u32x4 qh = *reinterpret_cast<const u32x4*>(lds + L.qh + qo);
            const u32x4 ql = *reinterpret_cast<const u32x4*>(lds + L.ql + qo);
            f32x16 o0, o1, zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; zero[e] = 0.f; }
            float lsum = 0.f, mref;
            bool big = false;
            {
                // first tile: fixes the reference
                f32x16 sA = qk_tile(lds, L, T0, r, hi, qh, ql, zero);
                if ((fmask >> T0) & 1) mask_tile(lds, L, T0, hi, 0.f, sA);
                const float tmax = xhalf_max(max16(sA));
                mref = tmax - P_SHIFT;
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) negm[e] = -mref;
                f32x16 sB;
                int t = T0 + 1;
                if (t < T1) sB = qk_tile(lds, L, t, r, hi, qh, ql, negm);
#pragma unroll
                for (int e = 0; e < 16; ++e) sA[e] -= mref;
                exp_pv_tile(lds, L, T0, r, hi, sA, lsum, big, o0, o1);
                while (t < T1) {
                    if (t + 1 < T1) sA = qk_tile(lds, L, t + 1, r, hi, qh, ql, negm);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, mref, sB);
                    exp_pv_tile(lds, L, t, r, hi, sB, lsum, big, o0, o1);
                    ++t;
                    if (t >= T1) break;
                    if (t + 1 < T1) sB = qk_tile(lds, L, t + 1, r, hi, qh, ql, negm);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, mref, sA);
                    exp_pv_tile(lds, L, t, r, hi, sA, lsum, big, o0, o1);
                    ++t;
                }
            }
            if (__any(big || !(lsum < 3.0e38f))) {
                // rare: a later logit exceeded the reference by more than the fp16 range of the probabilities allows --
                // redo the piece with the online update in every tile (probabilities <= 2^P_SHIFT)
#pragma unroll
                for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
                lsum = 0.f;
                float m_run = -1e30f;
                for (int t = T0; t < T1; ++t) {
                    f32x16 s = qk_tile(lds, L, t, r, hi, qh, ql, zero);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, 0.f, s);
                    const float m_new = max2f(m_run, xhalf_max(max16(s)));
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    mref = m_new - P_SHIFT;
                    lsum *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; s[e] -= mref; }
                    bool dummy = false;
                    exp_pv_tile(lds, L, t, r, hi, s, lsum, dummy, o0, o1);
                }
            }
            // partial of this piece: slot wave + qb
            float* pp = part + (size_t)(wave + qb) * 640 + lane;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pp[jj * 64] = (o0[jj] + o0[jj + 8]) + (o1[jj] + o1[jj + 8]);
            pp[8 * 64] = lsum;
            pp[9 * 64] = mref;
        }
        PRD2_STAMP(3);
        __syncthreads();
        PRD2_STAMP(4);
        // ================= merge, gate, store =================
        for (int qb = wave; qb < nqb; qb += NW) {
            const int lo_it = qb * nqb, hi_it = lo_it + nqb - 1;
            // waves whose range meets [lo_it, hi_it]: w_first = wave holding lo_it, w_last = wave holding hi_it
            int wf = 0, wl = 0;
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const int st = (int)((long)niter * w / NW);
                if (st <= lo_it) wf = w;
                if (st <= hi_it) wl = w;
            }
            // (a wave between the two whose own range is empty left no partial)
            auto has_piece = [&](int w) { return (int)((long)niter * (w + 1) / NW) > (int)((long)niter * w / NW); };
            float M = -INFINITY;
            for (int w = wf; w <= wl; ++w)
                if (has_piece(w)) M = max2f(M, part[(size_t)(w + qb) * 640 + 9 * 64 + lane]);
            float o[8], l = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
            for (int w = wf; w <= wl; ++w) {
                if (!has_piece(w)) continue;
                const float* pp = part + (size_t)(w + qb) * 640 + lane;
                const float scl = __builtin_amdgcn_exp2f(pp[9 * 64] - M);
                l += scl * pp[8 * 64];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o[jj] += scl * pp[jj * 64];
            }
            const float ltot = xhalf_add(l);
            const int v = 32 * qb + r;
            if (v < N) {
                const float il = 1.0f / (VSCALE * ltot);
                float res[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int c = 4 * hi + (jj & 3) + 8 * (jj >> 2);
                    res[jj] = Gl[c * NP + v] * (o[jj] * il);
                }
                float* dst = og + row_pos(bu, v) * HC + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
                *reinterpret_cast<float4*>(dst + 8) = make_float4(res[4], res[5], res[6], res[7]);
            }
        }
        PRD2_STAMP(5);
    }
}

size_t v2_lds_bytes(int N, int P, int NW) {
    const int NP = prd_round_up(N, 32);
    const size_t base = (size_t)64 * P * 4 + (size_t)NP * (4 * 32 + 64 + 64 + 4) + 128;
    return base + (size_t)(NP / 32 + NW) * 2560;
}

}  // namespace

#define PRD2_SET_LDS(kernel)                                                                                    \
    do {                                                                                                        \
        static std::once_flag once;                                                                             \
        std::call_once(once, [] {                                                                               \
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                     \
    } while (0)

// 1 when the second-generation core serves rows of N positions (split-16 arithmetic only)
extern "C" int prd_tri_attn_v2_supported(int N, int P) {
    if (N <= 0 || (P != 32 && P != 64)) return 0;
    return (N <= V2_MAXN && v2_lds_bytes(N, P, 8) <= 160 * 1024) ? 1 : 0;
}

extern "C" int prd_tri_attn_core_v2(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                                    const float* wv, const float* wg, const float* bg, int ending,
                                    int b, int N, int P, int H, int c, hipStream_t stream) {
    if (!og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (!prd_tri_attn_v2_supported(N, P)) return PRD_ERR_UNSUPPORTED;
    const int NP = prd_round_up(N, 32);
    const size_t lds = v2_lds_bytes(N, P, 8);
    const long rows_total = (long)b * N;
    const long cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head < 1) per_head = 1;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    const int grid = (int)(per_head * H);
    if (P == 64) {
        PRD2_SET_LDS((tri_attn_core_v2_kernel<64, 8>));
        hipLaunchKernelGGL((tri_attn_core_v2_kernel<64, 8>), dim3(grid), dim3(512), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N, NP, H,
                           ending);
    } else {
        PRD2_SET_LDS((tri_attn_core_v2_kernel<32, 8>));
        hipLaunchKernelGGL((tri_attn_core_v2_kernel<32, 8>), dim3(grid), dim3(512), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N, NP, H,
                           ending);
    }
    return (int)hipGetLastError();
}
