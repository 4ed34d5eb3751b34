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

#ifdef PRD_TIMING     // diagnostic builds only (tools/ta2_timing.py): cycle stamps [workgroup][8 waves][8 rows][16 stamps]
__device__ unsigned long long prd_dbg2[256 * 8 * 8 * 16];
extern "C" int prd_debug_read2(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prd_dbg2), sizeof(prd_dbg2)); }
#define PRD2_STAMP(k) do { if (lane == 0 && it < 8) prd_dbg2[((blockIdx.x * 8 + wave) * 8 + it) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PRD2_STAMP(k)
#endif

namespace {

constexpr float LOG2E_2 = 1.4426950408889634f;
constexpr float P_SHIFT = 4.0f;                 // probabilities are 2^(s - max + P_SHIFT)
constexpr int V2_MAXN = 384;

PRD_DEV f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// hi = RN_fp16 of the pair (a, b), lo = RN_fp16(x - hi) -- one more bit than the RTZ form of prd_common.h (the residual is
// signed), same three instructions per pair
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
PRD_DEV void split2h_rn(float a, float b, unsigned& hi, unsigned& lo) {
    // Measured on gfx950 (tools/ubench/valu_rate_bench.hip, SIMD cycles per wave64 instruction at >= 2 waves per SIMD):
    // v_cvt_pk_f16_f32 and v_fma_mix_f32 4.6, v_fma_mixlo/hi_f16 8.4 (the rate of a transcendental).  So the residuals are
    // formed in fp32 (v_fma_mix_f32: fp16 source half * -1 + fp32 source, exact) and packed by a second conversion:
    // 4 x 4.6 cycles per pair instead of 4.6 + 2 x 8.4.
    // The first conversion is left to the compiler (it selects v_cvt_pk_f16_f32): as the FIRST reader of a value that may come
    // straight out of an MFMA or a transcendental it must be an instruction whose hazards hipcc pads (an asm statement's
    // reads are not padded)
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, h16x2));
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
    L.g = off; off += (unsigned)NP * 64u;              // [position][hi][8] fp32: the 8 gate channels {4hi+e, 8+4hi+e} of a lane
    L.kadd = off; off += (unsigned)NP * 4u;
    L.flag = off; off += 64u;
    L.bias = off; off += 64u;
    L.part = off;                                      // [slot][10][64] fp32
    return L;
}

// LayerNorm without affine over a CLL row (ln_cll of prd_common.h with the cross-half sums on v_permlane32_swap instead of
// ds_bpermute: no LDS round trip in the middle of the projection phase)
template <int KH>
PRD_DEV void ln_cll_p(float (&x)[KH]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < KH; k += 2) { s0 += x[k]; s1 += x[k + 1]; }
    const float mean = xhalf_add(s0 + s1) * (1.0f / (2 * KH));
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int k = 0; k < KH; k += 2) {
        x[k] -= mean;
        x[k + 1] -= mean;
        v0 = __builtin_fmaf(x[k], x[k], v0);
        v1 = __builtin_fmaf(x[k + 1], x[k + 1], v1);
    }
    const float rstd = 1.0f / sqrtf(xhalf_add(v0 + v1) * (1.0f / (2 * KH)) + 1e-5f);
#pragma unroll
    for (int k = 0; k < KH; ++k) x[k] *= rstd;
}

struct KOp { u32x4 h, l; };                            // K hi | lo operands of one 32-key tile
struct PBuf { u32x4 ph0, pl0, ph1, pl1, va0, va1; };   // probabilities of a tile (fp16 hi | lo, 2 x 16 keys) + its V operands

PRD_DEV KOp load_k(const unsigned char* lds, unsigned kaddr, unsigned kl_off) {
    KOp k;
    k.h = *reinterpret_cast<const u32x4*>(lds + kaddr);
    k.l = *reinterpret_cast<const u32x4*>(lds + kaddr + kl_off);
    return k;
}

// S^T tile: 32 keys x 32 queries; C = cinit (all registers)
PRD_DEV f32x16 qk_tile(const KOp& k, u32x4 qh, u32x4 ql, const f32x16& cinit) {
    f32x16 s = mfma_h(k.h, qh, cinit);
    s = mfma_h(k.h, ql, s);
    s = mfma_h(k.l, qh, s);
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

// p = 2^s (s already relative to the reference), row-sum, split into the B operands of P V
PRD_DEV void exp_split(f32x16& s, float& lsum, bool& big, PBuf& p) {
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
    split8_rn(s, 0, p.ph0, p.pl0);
    split8_rn(s, 8, p.ph1, p.pl1);
}
PRD_DEV void load_v(const unsigned char* lds, unsigned vaddr, PBuf& p) {
    p.va0 = *reinterpret_cast<const u32x4*>(lds + vaddr);
    p.va1 = *reinterpret_cast<const u32x4*>(lds + vaddr + 1024u);
}
// O += [V_hi; V_lo] P for both 16-key halves of a tile
PRD_DEV void pv_tile(const PBuf& p, f32x16& o0, f32x16& o1) {
    o0 = mfma_h(p.va0, p.ph0, o0);
    o1 = mfma_h(p.va1, p.ph1, o1);
    o0 = mfma_h(p.va0, p.pl0, o0);
    o1 = mfma_h(p.va1, p.pl1, o1);
}

#define PRD2_FENCE() __builtin_amdgcn_sched_barrier(0)

// Priority = fraction of the wave's own key-loop work (tiles) still to do.  The waves of a SIMD are arbitrated oldest first:
// without this the older wave of a SIMD runs its tiles at full speed and the younger one then finishes alone, at the VALU
// issue rate of a single wave (measured: 14.0k vs 20.3k cycles per row for equal work; the SIMD is done when the slower is).
PRD_DEV void v2_prio(int rem, int tot) {
    if (4 * rem > 3 * tot) __builtin_amdgcn_s_setprio(3);
    else if (2 * rem > tot) __builtin_amdgcn_s_setprio(2);
    else if (4 * rem > tot) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// One steady-state step of the key loop, written in ISSUE ORDER.  The seven MFMAs of a step -- P V of tile t-1 (operands p,
// produced by the previous step) and Q K^T of tile t+1 (operands k, into sn) -- do not depend on the softmax arithmetic of
// tile t (sc -> p), so the ~60 VALU instructions of that arithmetic are dealt into the seven 32-cycle gaps behind the MFMAs
// (a wave issues in order: an MFMA occupies the matrix pipe for 32 cycles while the instructions behind it issue).  The P V
// MFMAs come first: once they are issued, their operand registers take the probabilities of tile t and the V operands of tile
// t (loaded for the next step); likewise the K registers take tile t+2 after the Q K^T MFMAs.  The scheduling fences keep
// hipcc from regrouping the MFMAs into one cluster.
PRD_DEV void pipe_step(const unsigned char* lds, unsigned kaddr_next, unsigned kl_off, unsigned vaddr, f32x16& sc, f32x16& sn,
                       const f32x16& negm, u32x4 qh, u32x4 ql, KOp& k, PBuf& p, f32x16& o0, f32x16& o1, float& lsum, bool& big) {
    float t0, t1;
    o0 = mfma_h(p.va0, p.ph0, o0);
    PRD2_FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
    t0 = sc[0] + sc[2];
    t1 = sc[1] + sc[3];
    PRD2_FENCE();
    o1 = mfma_h(p.va1, p.ph1, o1);
    PRD2_FENCE();
#pragma unroll
    for (int j = 4; j < 8; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
    t0 += sc[4]; t1 += sc[5]; t0 += sc[6]; t1 += sc[7];
    PRD2_FENCE();
    o0 = mfma_h(p.va0, p.pl0, o0);
    PRD2_FENCE();
#pragma unroll
    for (int j = 8; j < 12; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
    t0 += sc[8]; t1 += sc[9]; t0 += sc[10]; t1 += sc[11];
    PRD2_FENCE();
    o1 = mfma_h(p.va1, p.pl1, o1);
    PRD2_FENCE();
    load_v(lds, vaddr, p);                              // V of tile t: operands of the NEXT step's P V
#pragma unroll
    for (int j = 12; j < 16; ++j) sc[j] = __builtin_amdgcn_exp2f(sc[j]);
    t0 += sc[12]; t1 += sc[13]; t0 += sc[14]; t1 += sc[15];
    PRD2_FENCE();
    sn = mfma_h(k.h, qh, negm);
    PRD2_FENCE();
    split8_rn(sc, 0, p.ph0, p.pl0);
    PRD2_FENCE();
    sn = mfma_h(k.h, ql, sn);
    PRD2_FENCE();
    split8_rn(sc, 8, p.ph1, p.pl1);
    PRD2_FENCE();
    sn = mfma_h(k.l, qh, sn);
    PRD2_FENCE();
    k = load_k(lds, kaddr_next, kl_off);                // K of tile t+2
    {
        const float ts = t0 + t1;
        big |= !(ts < 30000.0f);
        lsum += ts;
    }
    PRD2_FENCE();
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
    stage_weight_h2_rows<P>(Wb, 64, 2 * C, wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E * H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 3 * C, wv + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    if (tid < 16) {                                     // gate bias in the register order of a lane: [hi][8]
        const int hh = tid >> 3, e = tid & 7;
        biasl[tid] = H2_WSCALE * NEG_LOG2E * bg[h * C + 4 * hh + (e & 3) + 8 * (e >> 2)];
    }
    const int nrows = b * N;                            // (the host checks that b * N * N fits an int)
    struct RowIx { int bu, bb, u; };
    auto make_row = [&](int bu) { RowIx x; x.bu = bu; x.bb = bu / N; x.u = bu - x.bb * N; return x; };       // 32-bit, once per row
    auto row_pos = [&](const RowIx& x, int v) -> long { return ending ? (long)((x.bb * N + v) * N + x.u) : (long)(x.bu * N + v); };
    // ---- static work split ----
    // phase 1: block w (all four projections) by wave w; the blocks past the NW-th as half-items (kind 0 = [K|Q], 1 = [V|G])
    // dealt round-robin
    const int nfullblk = nqb < NW ? nqb : NW;
    const int nextra = 2 * (nqb - nfullblk);
    // phase 2: query blocks [0, wholeq) belong to one wave each (w, w + NW, ...); the (query block, key tile) iterations of the
    // remaining R blocks are cut into NW contiguous ranges
    const int wholeq = (nqb / NW) * NW;
    const int R = nqb - wholeq;
    const int rem_iter = R * nqb;
    const int it_begin = (int)((long)rem_iter * wave / NW), it_end = (int)((long)rem_iter * (wave + 1) / NW);
    const float inv16 = H2_INV_WSCALE;
    const unsigned kl_off = L.kl - L.kh;
    const unsigned kbase = L.kh + (unsigned)hi * L.plane + (unsigned)r * 16u;      // + 512 t
    const unsigned vbase = L.v + (unsigned)hi * 512u + (unsigned)r * 16u;          // + 2048 t

    float xnext[KH];                                    // the wave's own block of the NEXT row
    float mknext = 0.f, munext = 0.f;                   // ... and its mask values (own block's positions; the row itself)
    RowIx rnext = make_row(slot < nrows ? slot : 0);
    {
        const int v = wave * 32 + r;
        const bool ok = slot < nrows && wave < nfullblk && v < N;
        load_row_cll<P>(pair + row_pos(rnext, ok ? v : 0) * P, hi, ok, xnext);
        if (ok) mknext = mask[rnext.bb * N + v];
        if (slot < nrows) munext = mask[slot];
    }
    int it = 0;
    for (int bu = slot; bu < nrows; bu += rstride, ++it) {
        const RowIx row = rnext;
        const int bb = row.bb;
        __syncthreads();                                // previous row's LDS consumed (and the weight image staged)
        PRD2_STAMP(0);
        const float mu = munext;
        // ================= phase 1 =================
        {
            auto key_override = [&](int blk, float mk) {           // logit override + tile flag of block blk
                const int v = blk * 32 + r;
                const bool valid = v < N;
                const bool keep = valid && (mu * mk >= 0.5f);
                if (hi == 0) kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E_2 : -INFINITY);
                const bool any_override = __any(!keep);
                if (lane == 0) tflag[blk] = any_override ? 1 : 0;
            };
            // image rows: K 0-15 | Q 16-31 | G 32-47 | V 48-63.  Three row GEMMs per block:
            //   kq: unswapped (A = rows r of the image): lane (pos, hi) registers 0-7 = K, 8-15 = Q channels {4hi+e, 8+4hi+e}
            //   g : unswapped, A = G rows 32 + (r & 15) (lanes 16-31 repeat them): registers 0-7 = gate channels {4hi+e, 8+4hi+e}
            //   v : SWAPPED, B = V rows 48 + (r & 15): lane (n, hi) register j = V channel n & 15 of position drow32(j, hi);
            //       lanes 0-15 keep the fp16 hi part, lanes 16-31 the lo part = rows 0-15 / 16-31 of the P V A operand
            auto wop = [&](int row, int s_, u32x4& wh, u32x4& wl) {
                const int slot_ = h2_slot<P>(row, 2 * s_ + hi);
                wh = Wb[(size_t)row * (P / 8) + slot_];
                wl = Wb[(size_t)(64 + row) * (P / 8) + slot_];
            };
            auto gemm_kq = [&](const u32x4 (&xs)[2][P / 16], f32x16& acc) {
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(r, s_, wh, wl);
                    acc = mfma_h(wh, xs[0][s_], acc);
                    acc = mfma_h(wh, xs[1][s_], acc);
                    acc = mfma_h(wl, xs[0][s_], acc);
                }
            };
            auto gemm_gv = [&](const u32x4 (&xs)[2][P / 16], f32x16& ag, f32x16& av) {
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 gh, gl, vh, vl;
                    wop(32 + (r & 15), s_, gh, gl);
                    wop(48 + (r & 15), s_, vh, vl);
                    ag = mfma_h(gh, xs[0][s_], ag);
                    av = mfma_h(xs[0][s_], vh, av);
                    ag = mfma_h(gh, xs[1][s_], ag);
                    av = mfma_h(xs[1][s_], vh, av);
                    ag = mfma_h(gl, xs[0][s_], ag);
                    av = mfma_h(xs[0][s_], vl, av);
                }
            };
            auto store_kq = [&](int blk, f32x16& acc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] *= inv16;
                u32x4 kh4, kl4, qh4, ql4;
                split8_rn(acc, 0, kh4, kl4);
                split8_rn(acc, 8, qh4, ql4);
                const unsigned po = (unsigned)hi * L.plane + (unsigned)(blk * 32 + r) * 16u;
                *reinterpret_cast<u32x4*>(lds + L.kh + po) = kh4;
                *reinterpret_cast<u32x4*>(lds + L.kl + po) = kl4;
                *reinterpret_cast<u32x4*>(lds + L.qh + po) = qh4;
                *reinterpret_cast<u32x4*>(lds + L.ql + po) = ql4;
            };
            auto g_init = [&](f32x16& ag) {
                const float4 b0 = *reinterpret_cast<const float4*>(biasl + 8 * hi), b1 = *reinterpret_cast<const float4*>(biasl + 8 * hi + 4);
                ag[0] = b0.x; ag[1] = b0.y; ag[2] = b0.z; ag[3] = b0.w; ag[4] = b1.x; ag[5] = b1.y; ag[6] = b1.z; ag[7] = b1.w;
#pragma unroll
                for (int e = 8; e < 16; ++e) ag[e] = 0.f;
            };
            auto store_gv = [&](int blk, const f32x16& ag, const f32x16& av) {
                float* gp = Gl + (size_t)((blk * 32 + r) * 2 + hi) * 8;
                *reinterpret_cast<float4*>(gp) = make_float4(gate_from_scaled(ag[0] * inv16), gate_from_scaled(ag[1] * inv16),
                                                             gate_from_scaled(ag[2] * inv16), gate_from_scaled(ag[3] * inv16));
                *reinterpret_cast<float4*>(gp + 4) = make_float4(gate_from_scaled(ag[4] * inv16), gate_from_scaled(ag[5] * inv16),
                                                                 gate_from_scaled(ag[6] * inv16), gate_from_scaled(ag[7] * inv16));
                u32x4 vh0, vl0, vh1, vl1;               // V stays x 16
                split8_rn(av, 0, vh0, vl0);
                split8_rn(av, 8, vh1, vl1);
                const bool lo_lane = r >= 16;
                u32x4 s0, s1;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s0[w] = lo_lane ? vl0[w] : vh0[w]; s1[w] = lo_lane ? vl1[w] : vh1[w]; }
                const unsigned vo = L.v + (unsigned)(blk * 4 + hi) * 512u + (unsigned)r * 16u;
                *reinterpret_cast<u32x4*>(lds + vo) = s0;
                *reinterpret_cast<u32x4*>(lds + vo + 1024u) = s1;
            };
            // extra half-items of this wave: rows requested before the own block is computed
            float xe[KH];
            float mke = 0.f;
            const int e0 = wave;                        // first extra half-item (block nfullblk + e0 / 2, kind e0 & 1)
            {
                const int v = (nfullblk + (e0 >> 1)) * 32 + r;
                const bool ok = e0 < nextra && v < N;
                load_row_cll<P>(pair + row_pos(row, ok ? v : 0) * P, hi, ok, xe);
                if (ok) mke = mask[bb * N + v];
            }
            if (wave < nfullblk) {                      // the wave's own block: LayerNorm + split once, three row GEMMs
                key_override(wave, mknext);
                PRD2_STAMP(6);
                ln_cll_p<KH>(xnext);
                u32x4 xs[2][P / 16];
                split2h_rn_cll<KH>(xnext, xs);
                f32x16 akq, ag, av;
#pragma unroll
                for (int e = 0; e < 16; ++e) { akq[e] = 0.f; av[e] = 0.f; }
                g_init(ag);
                PRD2_STAMP(7);
                gemm_kq(xs, akq);
                gemm_gv(xs, ag, av);
                PRD2_STAMP(8);
                store_kq(wave, akq);
                store_gv(wave, ag, av);
                PRD2_STAMP(9);
            }
            for (int e = e0; e < nextra; e += NW) {
                const int blk = nfullblk + (e >> 1), kind = e & 1;
                if (e != e0) {
                    const int v = blk * 32 + r;
                    const bool ok = v < N;
                    load_row_cll<P>(pair + row_pos(row, ok ? v : 0) * P, hi, ok, xe);
                    mke = ok ? mask[bb * N + v] : 0.f;
                }
                ln_cll_p<KH>(xe);
                u32x4 xs[2][P / 16];
                split2h_rn_cll<KH>(xe, xs);
                if (kind == 0) {
                    key_override(blk, mke);
                    f32x16 akq;
#pragma unroll
                    for (int q_ = 0; q_ < 16; ++q_) akq[q_] = 0.f;
                    gemm_kq(xs, akq);
                    store_kq(blk, akq);
                } else {
                    f32x16 ag, av;
#pragma unroll
                    for (int q_ = 0; q_ < 16; ++q_) av[q_] = 0.f;
                    g_init(ag);
                    gemm_gv(xs, ag, av);
                    store_gv(blk, ag, av);
                }
            }
        }
        PRD2_STAMP(1);
        __syncthreads();
        PRD2_STAMP(2);
        {   // the wave's own block of the next row: in flight during the key loops
            const int bun = bu + rstride;
            rnext = make_row(bun < nrows ? bun : 0);
            const int v = wave * 32 + r;
            const bool ok = bun < nrows && wave < nfullblk && v < N;
            load_row_cll<P>(pair + row_pos(rnext, ok ? v : 0) * P, hi, ok, xnext);
            mknext = ok ? mask[rnext.bb * N + v] : 0.f;
            munext = bun < nrows ? mask[bun] : 0.f;
        }
        // ================= phase 2 =================
        unsigned fmask;
        {
            const int f = lane < nqb ? tflag[lane] : 0;
            fmask = (unsigned)__ballot(f != 0);
        }
        // this wave's key-loop work of the row, in tiles (priority = share still to do)
        const int work_tot = ((wholeq - wave + NW - 1) / NW) * nqb + (it_end - it_begin);
        int work_rem = work_tot;
        // key tiles [T0, T1) for query block qb: o8 = O (x 16, relative to mref), lsum = the lane's part of the row sum
        auto run_piece = [&](int qb, int T0, int T1, float (&o8)[8], float& lsum, float& mref) {
            v2_prio(work_rem, work_tot);
            const unsigned qo = (unsigned)hi * L.plane + (unsigned)(32 * qb + r) * 16u;
            const u32x4 qh = *reinterpret_cast<const u32x4*>(lds + L.qh + qo);
            const u32x4 ql = *reinterpret_cast<const u32x4*>(lds + L.ql + qo);
            f32x16 o0, o1, zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; zero[e] = 0.f; }
            lsum = 0.f;
            bool big = false;
            const unsigned range_bits = (T1 >= 32 ? 0xffffffffu : ((1u << T1) - 1u)) & ~((1u << T0) - 1u);
            if ((fmask & range_bits) == 0) {
                // ---- fast path: no masked / padded key in the piece; software-pipelined steps ----
                const int tl = T1 - 1;                  // tiles past the piece are clamped to its last one (results unused)
                KOp k = load_k(lds, kbase + 512u * T0, kl_off);
                f32x16 sA = qk_tile(k, qh, ql, zero), sB;
                k = load_k(lds, kbase + 512u * (T0 + 1 < tl ? T0 + 1 : tl), kl_off);
                const float tmax = xhalf_max(max16(sA));
                mref = tmax - P_SHIFT;
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) negm[e] = -mref;
                sB = qk_tile(k, qh, ql, negm);                                   // tile T0 + 1 (or a clamped repeat)
                k = load_k(lds, kbase + 512u * (T0 + 2 < tl ? T0 + 2 : tl), kl_off);
#pragma unroll
                for (int e = 0; e < 16; ++e) sA[e] -= mref;
                PBuf p;
                exp_split(sA, lsum, big, p);
                load_v(lds, vbase + 2048u * T0, p);
                // invariant at the top of a step for tile t: s? = logits of tile t, p = tile t - 1, k = K of tile t + 1
                int t = T0 + 1;
                while (t < T1) {
                    v2_prio(work_rem - (t - T0), work_tot);
                    pipe_step(lds, kbase + 512u * (t + 2 < tl ? t + 2 : tl), kl_off, vbase + 2048u * t, sB, sA, negm, qh, ql, k, p,
                              o0, o1, lsum, big);
                    ++t;
                    if (t >= T1) break;
                    pipe_step(lds, kbase + 512u * (t + 2 < tl ? t + 2 : tl), kl_off, vbase + 2048u * t, sA, sB, negm, qh, ql, k, p,
                              o0, o1, lsum, big);
                    ++t;
                }
                pv_tile(p, o0, o1);                     // P V of the last tile
            } else {
                // ---- pieces with masked / padded keys: one tile at a time ----
                KOp k0 = load_k(lds, kbase + 512u * T0, kl_off);
                f32x16 s0 = qk_tile(k0, qh, ql, zero);
                if ((fmask >> T0) & 1) mask_tile(lds, L, T0, hi, 0.f, s0);
                const float tmax = xhalf_max(max16(s0));
                mref = tmax - P_SHIFT;
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) { negm[e] = -mref; s0[e] -= mref; }
                PBuf p;
                exp_split(s0, lsum, big, p);
                load_v(lds, vbase + 2048u * T0, p);
                pv_tile(p, o0, o1);
                for (int t = T0 + 1; t < T1; ++t) {
                    const KOp k = load_k(lds, kbase + 512u * t, kl_off);
                    f32x16 s = qk_tile(k, qh, ql, negm);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, mref, s);
                    exp_split(s, lsum, big, p);
                    load_v(lds, vbase + 2048u * t, p);
                    pv_tile(p, o0, o1);
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
                    const KOp k = load_k(lds, kbase + 512u * t, kl_off);
                    f32x16 s = qk_tile(k, qh, ql, zero);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, 0.f, s);
                    const float m_new = max2f(m_run, xhalf_max(max16(s)));
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    mref = m_new - P_SHIFT;
                    lsum *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; s[e] -= mref; }
                    bool dummy = false;
                    PBuf p;
                    exp_split(s, lsum, dummy, p);
                    load_v(lds, vbase + 2048u * t, p);
                    pv_tile(p, o0, o1);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o8[jj] = (o0[jj] + o0[jj + 8]) + (o1[jj] + o1[jj + 8]);
            work_rem -= T1 - T0;
        };
        // gate, normalise, store the 32 queries of block qb (l = the lane's part of the row sum)
        auto finish = [&](int qb, const float (&o)[8], float l) {
            const float ltot = xhalf_add(l);
            const int v = 32 * qb + r;
            if (v < N) {
                const float il = 1.0f / (VSCALE * ltot);
                const float* gp = Gl + (size_t)(v * 2 + hi) * 8;
                const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
                const float res[8] = {g0.x * (o[0] * il), g0.y * (o[1] * il), g0.z * (o[2] * il), g0.w * (o[3] * il),
                                      g1.x * (o[4] * il), g1.y * (o[5] * il), g1.z * (o[6] * il), g1.w * (o[7] * il)};
                float* dst = og + row_pos(row, v) * HC + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
                *reinterpret_cast<float4*>(dst + 8) = make_float4(res[4], res[5], res[6], res[7]);
            }
        };
        for (int qb = wave; qb < wholeq; qb += NW) {    // whole query blocks: no partials, no merge
            float o8[8], lsum, mref;
            run_piece(qb, 0, nqb, o8, lsum, mref);
            PRD2_STAMP(10);
            finish(qb, o8, lsum);
            PRD2_STAMP(11);
        }
        for (int i0 = it_begin; i0 < it_end;) {         // this wave's range of the shared blocks
            const int rq = i0 / nqb;
            const int T0 = i0 - rq * nqb;
            const int T1 = (it_end - rq * nqb) < nqb ? (it_end - rq * nqb) : nqb;       // exclusive
            i0 += T1 - T0;
            float o8[8], lsum, mref;
            run_piece(wholeq + rq, T0, T1, o8, lsum, mref);
            float* pp = part + (size_t)(wave + rq) * 640 + lane;               // partial of this piece: slot wave + rq
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pp[jj * 64] = o8[jj];
            pp[8 * 64] = lsum;
            pp[9 * 64] = mref;
        }
        __builtin_amdgcn_s_setprio(0);
        PRD2_STAMP(3);
        if (R > 0) {
            __syncthreads();
            PRD2_STAMP(4);
            // ================= merge of the shared blocks =================
            for (int rq = wave; rq < R; rq += NW) {
                const int lo_it = rq * nqb, hi_it = lo_it + nqb - 1;
                // waves whose range meets [lo_it, hi_it]: w_first = wave holding lo_it, w_last = wave holding hi_it
                int wf = 0, wl = 0;
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const int st = (int)((long)rem_iter * w / NW);
                    if (st <= lo_it) wf = w;
                    if (st <= hi_it) wl = w;
                }
                // (a wave between the two whose own range is empty left no partial).  Straight-line over the NW possible
                // pieces so that the LDS reads of different pieces are in flight together
                float mm[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const bool has = w >= wf && w <= wl && (int)((long)rem_iter * (w + 1) / NW) > (int)((long)rem_iter * w / NW);
                    mm[w] = has ? part[(size_t)(w + rq) * 640 + 9 * 64 + lane] : -INFINITY;
                }
                float M = mm[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) M = max2f(M, mm[w]);
                float o[8], l = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (mm[w] == -INFINITY) continue;              // wave-uniform: a piece exists for all lanes or for none
                    const float* pp = part + (size_t)(w + rq) * 640 + lane;
                    const float scl = __builtin_amdgcn_exp2f(mm[w] - M);
                    l += scl * pp[8 * 64];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) o[jj] += scl * pp[jj * 64];
                }
                finish(wholeq + rq, o, l);
            }
        }
        PRD2_STAMP(5);
    }
}

size_t v2_lds_bytes(int N, int P, int NW) {
    const int NP = prd_round_up(N, 32), nqb = NP / 32;
    const size_t base = (size_t)64 * P * 4 + (size_t)NP * (4 * 32 + 64 + 64 + 4) + 128;
    return base + (size_t)(NW + nqb % NW) * 2560;      // partials: slot = wave + shared-block index
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
    if ((long)b * N * N > 0x7fffffffL / 2) return PRD_ERR_UNSUPPORTED;      // 32-bit position arithmetic in the kernel
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
