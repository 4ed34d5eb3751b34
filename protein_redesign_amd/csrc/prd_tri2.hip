// Triangle attention core, second generation (split-16 arithmetic): tri_attn_core_v2_kernel for rows of up to 384 positions
// (K, Q, V and the gate of a row resident in LDS), tri_attn_core_v2l_kernel for longer rows as far as K and V of a row fit
// the LDS as fp16 planes (N <= 1024; described above that kernel).
//
// Replaces the reference's TriangleAttention -> Attention.forward chain (modules.py:236-243 -> 185-225) up to the gated
// per-head output `og`; the output projection stays in tri_attn_out / pair_tail (prd_tri.hip, prd_pair.hip).
//
// One PERSISTENT workgroup of 12 waves (three per SIMD) per CU serves one head for a strided set of pair rows, like the
// first-generation kernel (tri_attn_core_split_kernel), but everything inside a row is laid out for the 32x32x16 fp16 MFMA:
//
//   phase 1  per 32-position block of the row, three row GEMMs on the LayerNorm-ed, fp16 hi|lo split rows:
//            [K|Q]: unswapped (A = weights, B = the lane's row): lane (pos, hi) ends up with the 8 K channels {4hi+e, 8+4hi+e}
//                   and the same 8 Q channels of ITS position -- exactly the 8 contraction values an A / B operand lane of
//                   the QK^T MFMA holds.  K and Q go to LDS as fp16 hi | lo planes with one 16-byte store per plane; no
//                   transposition, no three-way split.
//            [G]  : unswapped: the 8 gate channels of the lane's position, kept in fp32 for the epilogue.
//            [V]  : SWAPPED (A = the rows, B = weights): lane (channel, hi) ends up with 16 positions of ONE channel, which is
//                   the key-major order the P V MFMA wants from its A operand; lanes 0-15 keep the fp16 hi part, lanes 16-31
//                   (the same 16 channels again) the lo part: two 16-byte stores per lane (the first generation scattered
//                   2-byte values).
//   phase 2  S^T = K Q^T for 32 keys x 32 queries per MFMA triple (kh qh + kh ql + kl qh; the contraction is the head width
//            16 = the K of the instruction, nothing is padded), so a lane holds 16 logits of ONE query: softmax statistics
//            are lane-local plus one cross-half exchange.  The probabilities, split into fp16 hi | lo, are directly the B
//            operand of O^T = [V_hi; V_lo] P^T (M = 32 = both planes of the 16 channels: hi*hi, lo*hi, hi*lo, lo*lo in two
//            MFMAs per 16 keys), because V was stored in the key order of the S^T register layout.
//            Work split: query block q belongs to wave q (its keys in one sweep, result gated and stored directly).  When
//            the number of blocks is not a multiple of 4, the SIMDs that own one block more shed key tiles of that block to
//            the idle waves of the other SIMDs (every shed piece is a quarter of the row's keys); those blocks are merged
//            from partials (reference, sum, O) in LDS after one barrier (flash-decoding merge).
//            The reference maximum of a piece is fixed by its first tile (later tiles: accumulator preloaded with
//            -reference, one v_exp_f32 per logit); should a probability leave the fp16 range the piece is redone with the
//            online update in every tile.
//
// Why this shape (tools/ubench/{valu_rate,overlap,tile_step}_bench.hip, MI355X): the kernel is VALU-bound, not MFMA-bound.
// Per 32 x 32 logits a wave issues 7 MFMAs (224 matrix-pipe cycles) against 16 v_exp_f32 (8.4 SIMD cycles each), 32
// conversion-class instructions for the split (4.6 each) and 16 adds (2.9): ~330 VALU cycles, of which an MFMA only hides
// about 40 % of its own duration.  So the design minimises VALU instructions per logit (fp32 residuals + two packed
// conversions instead of v_fma_mix*_f16, which issues at the transcendental rate; no transposition work; LayerNorm and split
// of a block done once) and keeps three waves per SIMD resident to fill each other's stalls.
//
// Arithmetic: operands hi = RN_fp16(x), lo = RN_fp16(x - hi): |x - hi - lo| <= 2^-24 |x| while lo is a normal fp16 number, an
// absolute 2^-25 below that; products accumulate in fp32.  Probabilities are kept x 2^4 relative to the reference maximum
// so that small probabilities keep a normal lo part.
#include "prd_common.h"
#include "../../include/prd_hip.h"
#include <cstdlib>
#include <mutex>

#ifdef PRD_TIMING     // diagnostic builds only (tools/ta2_timing.py): cycle stamps [workgroup][12 waves][8 rows][16 stamps]
__device__ unsigned long long prd_dbg2[256 * 12 * 8 * 16];
extern "C" int prd_debug_read2(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prd_dbg2), sizeof(prd_dbg2)); }
#define PRD2_STAMP(k) do { if (lane == 0 && it < 8) prd_dbg2[((blockIdx.x * 12 + wave) * 8 + it) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PRD2_STAMP(k)
#endif

namespace {

constexpr float LOG2E_2 = 1.4426950408889634f;
constexpr float P_SHIFT = 4.0f;                 // probabilities are 2^(s - max + P_SHIFT)
constexpr int V2_MAXN = 384;
constexpr int PRD_V3_DEFAULT_KL = 0;        // key-loop form of tri_attn_core_v3_kernel when the caller gives no flags

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

// 16-byte slot of V channel ch (0-15; bit 4 = the lo plane) inside a (tile, 16-key half, khalf) block of the GV layout: bits 2 and 3
// of the channel swapped.  The transposed 4-byte stores of phase 1 are issued 32 lanes at a time (lanes of one hi half: channels
// {4 hi + j} on the even lanes, {8 + 4 hi + j} on the odd ones, 16 (a, khalf, w) combinations each); a bank is 4 (slot & 7) + w, so the
// two channel sets must differ in slot bit 2, not bit 3; the reader XORs 2 a + khalf on top (sixteen-lane read groups stay distinct
// mod 16 under both).
PRD_DEV int gv_slot(int ch) { return (ch & ~12) | ((ch & 4) << 1) | ((ch & 8) >> 1); }
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

PRD_DEV void mask_tile_at(const unsigned char* lds, unsigned kadd_off, int T, int hi, float mref, f32x16& s) {
    const float* kadd = reinterpret_cast<const float*>(lds + kadd_off) + 32 * T + 4 * hi;
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
// the two halves of exp_split apart (key-loop forms 1-3 of tri_attn_core_v3_kernel issue the next tile's Q K^T between them)
template <bool SUM>
PRD_DEV void exp_sum(f32x16& s, float& lsum, bool& big) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        s[j] = __builtin_amdgcn_exp2f(s[j]);
        s[j + 1] = __builtin_amdgcn_exp2f(s[j + 1]);
        if (SUM) { t0 += s[j]; t1 += s[j + 1]; }
    }
    if (SUM) {
        const float ts = t0 + t1;
        big |= !(ts < 30000.0f);                       // (the fp32 sum stays finite when an fp16 hi part overflows: test it here)
        lsum += ts;
    }
}
PRD_DEV void split_p(const f32x16& s, PBuf& p) {
    split8_rn(s, 0, p.ph0, p.pl0);
    split8_rn(s, 8, p.ph1, p.pl1);
}
// Row sum on the matrix pipe: v_mfma_f32_4x4x4_16B_f16 with A = ones adds a lane's OWN four fp16 values (its B operand) to each
// of its four accumulator registers -- 8 small MFMAs (8 cycles each) per tile for the hi and lo parts of the 16 probabilities
// instead of 18 v_add_f32.  An overflowed probability (+inf in fp16) makes the sum inf: the piece is then redone online.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
PRD_DEV void rowsum_mfma(const PBuf& p, f32x4& la0, f32x4& la1) {
    const h16x4 ones = {(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
#define PRD_SUM4(acc, v, w) acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(h16x4, u32x2_t{v[w], v[w + 1]}), acc, 0, 0, 0)
    PRD_SUM4(la0, p.ph0, 0); PRD_SUM4(la1, p.ph0, 2); PRD_SUM4(la0, p.ph1, 0); PRD_SUM4(la1, p.ph1, 2);
    PRD_SUM4(la0, p.pl0, 0); PRD_SUM4(la1, p.pl0, 2); PRD_SUM4(la0, p.pl1, 0); PRD_SUM4(la1, p.pl1, 2);
#undef PRD_SUM4
}
PRD_DEV void load_v(const unsigned char* lds, unsigned vaddr, PBuf& p) {
    p.va0 = *reinterpret_cast<const u32x4*>(lds + vaddr);
    p.va1 = *reinterpret_cast<const u32x4*>(lds + vaddr + 1024u);
}
// O += [V_hi; V_lo] P for both 16-key halves of a tile
PRD_DEV void pv_tile(const PBuf& p, f32x16& o0) {
    // one accumulator: an accumulate chain issues back to back, and the other waves of the SIMD fill the pipe anyway
    o0 = mfma_h(p.va0, p.ph0, o0);
    o0 = mfma_h(p.va1, p.ph1, o0);
    o0 = mfma_h(p.va0, p.pl0, o0);
    o0 = mfma_h(p.va1, p.pl1, o0);
}

// Priority = fraction of the wave's own key-loop work (tiles) still to do.  The waves of a SIMD are arbitrated oldest first:
// without this the older wave runs its tiles at full speed and the younger ones then finish alone, at the VALU issue rate
// of a single wave (measured with two waves: 14.0k vs 20.3k cycles per row for equal work; the SIMD is done when the slowest is).
PRD_DEV void v2_prio(int rem, int tot) {
    if (4 * rem > 3 * tot) __builtin_amdgcn_s_setprio(3);
    else if (2 * rem > tot) __builtin_amdgcn_s_setprio(2);
    else if (4 * rem > tot) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// maximum of the 16 registers of a fresh MFMA result: compiler-visible v_max (hipcc pads the MFMA -> VALU read hazard; it does
// not for the reads of an asm statement)
PRD_DEV float max16_mfma(const f32x16& s) {
    float m0 = __builtin_fmaxf(s[0], s[1]), m1 = __builtin_fmaxf(s[2], s[3]);
#pragma unroll
    for (int j = 4; j < 16; j += 2) { m0 = __builtin_fmaxf(m0, s[j]); m1 = __builtin_fmaxf(m1, s[j + 1]); }
    return __builtin_fmaxf(m0, m1);
}

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_v2_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending, int flags,
    float* __restrict__ lse_out) {
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
    const int nqb = NP / 32;                            // query blocks = key tiles (<= NW)
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
    // image rows: K 0-15 | Q 16-31 | G 32-47 | V 48-63 (fp16 hi | lo planes, x 16 so that small weights keep a normal lo part)
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

    // ---- static work split (wave w sits on SIMD w & 3: only the balance depends on it) ----
    // m = nqb mod 4 "group" blocks G_i = nqb - m + i are owned by waves on SIMDs 0 .. m-1, which therefore carry one block more
    // than the others.  Helper wave j = nqb + j (j < 4 - m, on SIMD m + j) takes a quarter of the keys of EVERY group block;
    // the owner keeps the first m quarters.  Phase 1: helper j < m also computes the [G, V] projections of group block j.
    const int m4 = nqb & 3, gbase = nqb - m4, nhelp = m4 ? 4 - m4 : 0;
    const bool owner = wave < nqb, helper = wave >= nqb && wave < nqb + nhelp;
    const int hj = wave - nqb;                          // helper index
    const bool group_owner = owner && wave >= gbase;    // owns a block that is shared with the helpers
    const int gi = wave - gbase;                        // its index in the group
    // phase 1 item: block + kinds (bit 0 = [K|Q], bit 1 = [G, V])
    int p1_blk = -1, p1_kinds = 0;
    if (owner) { p1_blk = wave; p1_kinds = (group_owner && gi < nhelp) ? 1 : 3; }
    else if (helper && hj < m4) { p1_blk = gbase + hj; p1_kinds = 2; }
    // quarter boundaries of the key tiles
    auto qtile = [&](int k) { return (nqb * k) >> 2; };
    const int own_t1 = group_owner ? qtile(m4) : nqb;   // the owner's key tiles [0, own_t1)
    const int work_tot = owner ? own_t1 : (helper ? m4 * (qtile(m4 + hj + 1) - qtile(m4 + hj)) : 0);
    const float inv16 = H2_INV_WSCALE;
    const unsigned kl_off = L.kl - L.kh;
    const unsigned kbase = L.kh + (unsigned)hi * L.plane + (unsigned)r * 16u;      // + 512 t
    // (V slots in the GV layout of tri_attn_core_v3_kernel: channel bits 2 / 3 swapped, XORed with 2 a + khalf)
    const unsigned vb0 = L.v + (unsigned)hi * 512u + (unsigned)(gv_slot(r) ^ hi) * 16u;              // + 2048 t
    const unsigned vb1 = L.v + 1024u + (unsigned)hi * 512u + (unsigned)(gv_slot(r) ^ (2 + hi)) * 16u;
    auto ldv = [&](int t, PBuf& p) {
        p.va0 = *reinterpret_cast<const u32x4*>(lds + vb0 + 2048u * t);
        p.va1 = *reinterpret_cast<const u32x4*>(lds + vb1 + 2048u * t);
    };

    float xnext[KH];                                    // the rows of the wave's phase-1 block of the NEXT row
    float mknext = 0.f, munext = 0.f;                   // ... and its mask values (the block's positions; the row itself)
    RowIx rnext = make_row(slot < nrows ? slot : 0);
    {
        const int v = p1_blk * 32 + r;
        const bool ok = slot < nrows && p1_blk >= 0 && v < N;
        load_row_cll<P>(pair + row_pos(rnext, ok ? v : 0) * P, hi, ok, xnext);
        if (ok) mknext = mask[rnext.bb * N + v];
        if (slot < nrows) munext = mask[slot];
    }
    int it = 0;
    for (int bu = slot; bu < nrows; bu += rstride, ++it) {
        const RowIx row = rnext;
        __syncthreads();                                // previous row's LDS consumed (and the weight image staged)
        PRD2_STAMP(0);
        const float mu = munext;
        // ================= phase 1 =================
        if (p1_blk >= 0) {
            const int blk = p1_blk;
            // lane coordinates made opaque per row: otherwise hipcc hoists the ~30 lane-invariant LDS addresses of this phase (weight
            // operands of 12 k-steps, K / Q / V / gate stores) out of the row loop, where they sit in registers through the key
            // loops and spill (measured: 9.4 MB of scratch writes per launch)
            int r1 = r, hi1 = hi;
            asm volatile("" : "+v"(r1), "+v"(hi1));
            auto wop = [&](int wrow, int s_, u32x4& wh, u32x4& wl) {
                const int slot_ = h2_slot<P>(wrow, 2 * s_ + hi1);
                wh = Wb[(size_t)wrow * (P / 8) + slot_];
                wl = Wb[(size_t)(64 + wrow) * (P / 8) + slot_];
            };
            ln_cll_p<KH>(xnext);
            u32x4 xs[2][P / 16];
            split2h_rn_cll<KH>(xnext, xs);
            PRD2_STAMP(6);
            if (p1_kinds & 1) {
                {   // logit override of masked / padded keys + tile flag
                    const int v = blk * 32 + r1;
                    const bool valid = v < N;
                    const bool keep = valid && (mu * mknext >= 0.5f);
                    if (hi1 == 0) kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E_2 : -INFINITY);
                    const bool any_override = __any(!keep);
                    if (lane == 0) tflag[blk] = any_override ? 1 : 0;
                }
                // [K|Q]: lane (pos, hi1) registers 0-7 = K, 8-15 = Q channels {4hi+e, 8+4hi+e}
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(r1, s_, wh, wl);
                    acc = mfma_h(wh, xs[0][s_], acc);
                    acc = mfma_h(wh, xs[1][s_], acc);
                    acc = mfma_h(wl, xs[0][s_], acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] *= inv16;
                u32x4 kh4, kl4, qh4, ql4;
                split8_rn(acc, 0, kh4, kl4);
                split8_rn(acc, 8, qh4, ql4);
                const unsigned po = (unsigned)hi1 * L.plane + (unsigned)(blk * 32 + r1) * 16u;
                *reinterpret_cast<u32x4*>(lds + L.kh + po) = kh4;
                *reinterpret_cast<u32x4*>(lds + L.kl + po) = kl4;
                *reinterpret_cast<u32x4*>(lds + L.qh + po) = qh4;
                *reinterpret_cast<u32x4*>(lds + L.ql + po) = ql4;
            }
            if (p1_kinds & 2) {
                // [G|V] as ONE unswapped row GEMM (image rows 32 + r1) + transposed V store: see tri_attn_core_v3_kernel
                f32x16 agv;
                {
                    const float4 b0 = *reinterpret_cast<const float4*>(biasl + 8 * hi1), b1 = *reinterpret_cast<const float4*>(biasl + 8 * hi1 + 4);
                    agv[0] = b0.x; agv[1] = b0.y; agv[2] = b0.z; agv[3] = b0.w; agv[4] = b1.x; agv[5] = b1.y; agv[6] = b1.z; agv[7] = b1.w;
#pragma unroll
                    for (int e = 8; e < 16; ++e) agv[e] = 0.f;
                }
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 gh, gl;
                    wop(32 + r1, s_, gh, gl);
                    agv = mfma_h(gh, xs[0][s_], agv);
                    agv = mfma_h(gh, xs[1][s_], agv);
                    agv = mfma_h(gl, xs[0][s_], agv);
                }
                float* gp = Gl + (size_t)((blk * 32 + r1) * 2 + hi1) * 8;
                *reinterpret_cast<float4*>(gp) = make_float4(gate_from_scaled(agv[0] * inv16), gate_from_scaled(agv[1] * inv16),
                                                             gate_from_scaled(agv[2] * inv16), gate_from_scaled(agv[3] * inv16));
                *reinterpret_cast<float4*>(gp + 4) = make_float4(gate_from_scaled(agv[4] * inv16), gate_from_scaled(agv[5] * inv16),
                                                                 gate_from_scaled(agv[6] * inv16), gate_from_scaled(agv[7] * inv16));
                const bool odd = (r1 & 1) != 0;
                const int kp0 = r1 & 30;
                const int a_ = kp0 >> 4, kk = kp0 & 15, kh_ = (kk >> 2) & 1, w_ = (kk >> 3) * 2 + ((kk & 3) >> 1);
                const int ch0 = 4 * hi1 + (odd ? 8 : 0);
                const unsigned vo = L.v + (unsigned)blk * 2048u + (unsigned)a_ * 1024u + (unsigned)kh_ * 512u + (unsigned)w_ * 4u;
                const int sx = 2 * a_ + kh_;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mine_lo = agv[8 + j], mine_hi = agv[12 + j];
                    const float give = odd ? mine_lo : mine_hi;
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, false));
                    const float ka = odd ? got : mine_lo, kb = odd ? mine_hi : got;
                    unsigned hh, ll;
                    split2h_rn(ka, kb, hh, ll);
                    const unsigned so = (unsigned)(gv_slot(ch0 + j) ^ sx) * 16u;
                    *reinterpret_cast<unsigned*>(lds + vo + so) = hh;
                    *reinterpret_cast<unsigned*>(lds + vo + 256u + so) = ll;
                }
            }
        }
        PRD2_STAMP(1);
        __syncthreads();
        PRD2_STAMP(2);
        {   // the wave's phase-1 block of the next row: in flight during the key loops
            const int bun = bu + rstride;
            rnext = make_row(bun < nrows ? bun : 0);
            const int v = p1_blk * 32 + r;
            const bool ok = bun < nrows && p1_blk >= 0 && v < N;
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
        int work_rem = work_tot;
        if (flags & 6) {                                // stagger the waves of a SIMD (w, w + 4, w + 8) so that their MFMA and VALU
            const int j = wave >> 2;                    // segments do not coincide
            if (j == 1) { if ((flags & 6) == 2) __builtin_amdgcn_s_sleep(3); else if ((flags & 6) == 4) __builtin_amdgcn_s_sleep(5); else __builtin_amdgcn_s_sleep(8); }
            else if (j == 2) { if ((flags & 6) == 2) __builtin_amdgcn_s_sleep(6); else if ((flags & 6) == 4) __builtin_amdgcn_s_sleep(10); else __builtin_amdgcn_s_sleep(16); }
        }
        // key tiles [T0, T1) for query block qb: o8 = O (x 16, relative to mref), lsum = the lane's part of the row sum
        auto run_piece = [&](int qb, int T0, int T1, float (&o8)[8], float& lsum, float& mref) {
            const unsigned qo = (unsigned)hi * L.plane + (unsigned)(32 * qb + r) * 16u;
            const u32x4 qh = *reinterpret_cast<const u32x4*>(lds + L.qh + qo);
            const u32x4 ql = *reinterpret_cast<const u32x4*>(lds + L.ql + qo);
            f32x16 o0, zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; zero[e] = 0.f; }
            lsum = 0.f;
            bool big = false;
            if (flags & 1) v2_prio(work_rem, work_tot);
            {
                KOp k = load_k(lds, kbase + 512u * T0, kl_off);
                f32x16 s0 = qk_tile(k, qh, ql, zero);
                if (T0 + 1 < T1) k = load_k(lds, kbase + 512u * (T0 + 1), kl_off);
                if ((fmask >> T0) & 1) mask_tile(lds, L, T0, hi, 0.f, s0);
                const float tmax = xhalf_max(max16_mfma(s0));
                mref = tmax - P_SHIFT;
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) { negm[e] = -mref; s0[e] -= mref; }
                PBuf p;
                ldv(T0, p);
                exp_split(s0, lsum, big, p);
                pv_tile(p, o0);
                for (int t = T0 + 1; t < T1; ++t) {
                    if (flags & 1) v2_prio(work_rem - (t - T0), work_tot);
                    f32x16 s = qk_tile(k, qh, ql, negm);
                    if (t + 1 < T1) k = load_k(lds, kbase + 512u * (t + 1), kl_off);
                    ldv(t, p);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, mref, s);
                    exp_split(s, lsum, big, p);
                    pv_tile(p, o0);
                }
            }
            if (__any(big || !(lsum < 3.0e38f))) {
                // rare: a later logit exceeded the reference by more than the fp16 range of the probabilities allows --
                // redo the piece with the online update in every tile (probabilities <= 2^P_SHIFT)
#pragma unroll
                for (int e = 0; e < 16; ++e) o0[e] = 0.f;
                lsum = 0.f;
                float m_run = -1e30f;
                for (int t = T0; t < T1; ++t) {
                    const KOp k = load_k(lds, kbase + 512u * t, kl_off);
                    f32x16 s = qk_tile(k, qh, ql, zero);
                    if ((fmask >> t) & 1) mask_tile(lds, L, t, hi, 0.f, s);
                    const float m_new = max2f(m_run, xhalf_max(max16_mfma(s)));
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    mref = m_new - P_SHIFT;
                    lsum *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; s[e] -= mref; }
                    bool dummy = false;
                    PBuf p;
                    ldv(t, p);
                    exp_split(s, lsum, dummy, p);
                    pv_tile(p, o0);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o8[jj] = o0[jj] + o0[jj + 8];
            work_rem -= T1 - T0;
        };
        // gate, normalise, store the 32 queries of block qb (l = the lane's part of the row sum)
        auto finish = [&](int qb, const float (&o)[8], float l, float mref) {
            const float ltot = xhalf_add(l);
            const int v = 32 * qb + r;
            if (v < N) {
                // log2-domain log-sum-exp of the query's logits, for the backward core (prd_tri_attn_bwd_core_v2)
                // (two terms: a fully masked row has m = the fill value -32768 log2(e), whose sum with log2(l) is not exact)
                if (lse_out && hi == 0)
                    *reinterpret_cast<float2*>(lse_out + (((long)bu * H + h) * N + v) * 2) = make_float2(mref, __builtin_amdgcn_logf(ltot));
                const float il = 1.0f / (VSCALE * ltot);
                const float* gp = Gl + (size_t)(v * 2 + hi) * 8;
                const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
                float* dst = og + row_pos(row, v) * HC + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(g0.x * (o[0] * il), g0.y * (o[1] * il), g0.z * (o[2] * il), g0.w * (o[3] * il));
                *reinterpret_cast<float4*>(dst + 8) = make_float4(g1.x * (o[4] * il), g1.y * (o[5] * il), g1.z * (o[6] * il), g1.w * (o[7] * il));
            }
        };
        auto put_partial = [&](int pslot, const float (&o8)[8], float lsum, float mref) {
            float* pp = part + (size_t)pslot * 640 + lane;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pp[jj * 64] = o8[jj];
            pp[8 * 64] = lsum;
            pp[9 * 64] = mref;
        };
        // partial slots of group block i: (nhelp + 1) * i + {0: owner, 1 + j: helper j}
        if (owner) {
            float o8[8], lsum, mref;
            if (own_t1 > 0) run_piece(wave, 0, own_t1, o8, lsum, mref);
            if (!group_owner) finish(wave, o8, lsum, mref);
            else if (own_t1 > 0) put_partial((nhelp + 1) * gi, o8, lsum, mref);
        } else if (helper) {
            const int T0 = qtile(m4 + hj), T1 = qtile(m4 + hj + 1);
            for (int i = 0; i < m4; ++i) {
                if (T1 <= T0) break;
                float o8[8], lsum, mref;
                run_piece(gbase + i, T0, T1, o8, lsum, mref);
                put_partial((nhelp + 1) * i + 1 + hj, o8, lsum, mref);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PRD2_STAMP(3);
        if (m4 > 0) {
            __syncthreads();
            PRD2_STAMP(4);
            // ================= merge of the shared blocks (by their owners) =================
            if (group_owner) {
                // partials of the owner (k = 0) and of the helpers (k = 1 ..): a missing piece reads a valid slot with weight 0,
                // so that all LDS reads are in flight together
                const int s0 = (nhelp + 1) * gi;
                float mm[5];
                bool has[5];
                // (a missing piece reads the slot of a piece that EXISTS, with weight 0: with fewer than four key tiles -- rows of up to
                // 96 positions -- some helpers have no tile and never write their slot; reading one of those fed whatever the LDS held,
                // possibly a NaN, into 0 * x: found in round 5 by poisoning the LDS, tools/ubench/lds_poison.hip)
                int kv = 0;
#pragma unroll
                for (int k = 4; k >= 0; --k) {
                    has[k] = k == 0 ? own_t1 > 0 : (k - 1 < nhelp && qtile(m4 + k) > qtile(m4 + k - 1));
                    if (has[k]) kv = k;
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float v_ = part[(size_t)(s0 + (has[k] ? k : kv)) * 640 + 9 * 64 + lane];
                    mm[k] = has[k] ? v_ : -INFINITY;
                }
                float M = mm[0];
#pragma unroll
                for (int k = 1; k < 5; ++k) M = max2f(M, mm[k]);
                float o[8], l = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float* pp = part + (size_t)(s0 + (has[k] ? k : kv)) * 640 + lane;
                    const float scl = has[k] ? __builtin_amdgcn_exp2f(mm[k] - M) : 0.f;
                    l += scl * pp[8 * 64];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) o[jj] += scl * pp[jj * 64];
                }
                finish(wave, o, l, M);
            }
        }
        PRD2_STAMP(5);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Third form for short rows (N <= 384): the phases of consecutive rows OVERLAP.  tri_attn_core_v2_kernel runs
// barrier | phase 1 | barrier | phase 2 | barrier | merge per row, and 39 % of a row's cycles are not key loops: phase 1 is
// latency (row load), LayerNorm / split VALU and a short MFMA burst, executed by all waves at the same time, then everybody
// waits.  Here K / V of a row live in one of TWO LDS buffers and a wave runs, per iteration,
//     barrier | merge of the previous row's shared blocks | phase 2 of row r (buffer r & 1) | phase 1 of row r + 1 (other buffer)
// so that one wave's projection of the next row fills the SIMD while its neighbours are still in their key loops, and a row
// costs ONE barrier.  What makes the second buffer fit: Q and the gate never go to LDS -- wave q projects block q and sweeps
// block q, and the lane that computes the 8 Q channels (gate channels) of a position is the lane that needs them as the B
// operand of Q K^T (that gates that position's output): they stay in 16 registers from phase 1 to phase 2.  Only the shared
// ("group") blocks publish Q (2 KB each, for their helpers) and their gate (for the merge).
// Buffer reuse is safe by construction: a buffer / Q-share slot / partial slot written in iteration i was last read in
// iteration i - 1, one barrier earlier; the gate share is read at the top of iteration i + 2 and therefore rotates over three.
struct V3Lds { unsigned buf0, bufsize, plane, bias, qs, gs, part; int nslot; };

PRD_DEV V3Lds v3_layout(int P, int NP) {
    V3Lds L;
    const int nqb = NP / 32, m4 = nqb & 3, nhelp = m4 ? 4 - m4 : 0;
    unsigned off = 64u * P * 4u;
    L.plane = (unsigned)NP * 16u;
    L.bufsize = 8u * L.plane + (unsigned)NP * 4u + 64u;      // kh (2 planes) | kl (2) | v (4) | kadd | tile flags
    L.buf0 = off; off += 2u * L.bufsize;
    L.bias = off; off += 64u;
    L.qs = off; off += 2u * (unsigned)m4 * 2048u;            // [2][group block][qh | ql][hi][32][16 B]
    L.gs = off; off += 3u * (unsigned)m4 * 2048u;            // [3][group block][position][hi][8] fp32
    L.nslot = (nhelp + 1) * m4;
    L.part = off;                                            // [2][slot][10][64] fp32
    return L;
}

// KL = key-loop form (A/B: PRD_TA2_FLAGS bits 1-2; measured in DESIGN.md 4.3, round 5): 0 the round-3 order (Q K^T, exp + row
// sum, split, P V); bit 0: the NEXT tile's Q K^T is issued between the exponentials and the split of this one (one more logit
// tile in registers); bit 1: the row sum on the matrix pipe (rowsum_mfma) instead of 18 v_add_f32
template <int P, int NW, int KL, bool GV>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_v3_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending, int flags,
    float* __restrict__ lse_out) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#include "prd_tri2_v3_body.inc"
}

// The same statements as a device function, for tri_attn_pair_kernel (below), which runs them twice inside one persistent launch.
// (__restrict__ on og / pair says what it always said -- no two of these pointers overlap; what OTHER workgroups wrote before a grid
// barrier is the barrier's business: agent-scope release / acquire; the rows are per-lane vector loads, not the scalar cache.)
template <int P, int NW, int KL, bool GV>
PRD_DEV void tri_attn_core_v3_body(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending, int flags,
    float* __restrict__ lse_out, unsigned char* lds) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;
#include "prd_tri2_v3_body.inc"
}

// ---------------------------------------------------------------------------------------------------------------------------
// SURVEY 8(f)#4 "persistent per-block kernels", built for one seam of the folding block (modules.py:338-339): the starting triangle
// attention's core, its output projection + residual, and the ending attention's core as ONE persistent launch --
//     core(starting): og <- attention(pair rows)  |  grid barrier  |  pair <- pair + W_o og + b_o  |  grid barrier  |  core(ending)
// instead of three launches (prd_tri_attn_core_v2 + prd_tri_attn_out + prd_tri_attn_core_v2).  Same stage bodies, same task order:
// bit-identical results.  Two barrier forms of the MI355X guide, both placement-independent: the XCD-hierarchical one (default) and the
// plain counter (every workgroup releases at agent scope and polls ONE word; A/B).  Every spin is BOUNDED: a grid that is not fully
// resident (a CU less than workgroups) sets the timeout word and runs on unsynchronised instead of hanging the box; the host entry
// refuses grids above the CU count.  Opt-in (PRD_PERSISTENT_ATTN=1): measured in DESIGN.md 4.3.
// bar (uint32 words, zeroed by the host entry before EVERY launch):
//   [0] arrivals of the plain counter form   [1] timeout flag   [2] registrations   [3] top counter (one arrival per XCD and barrier)
//   [8 + x] members of XCD x   [16 + x] arrivals on XCD x   [24 + x] generation published to XCD x     (x = hardware XCC id, 0-7)
constexpr unsigned PRD_SPIN_LIMIT = 1u << 22;          // ~ a second of polling: the grid is not resident

PRD_DEV bool spin_until_ge(unsigned* word, unsigned target, unsigned* tmo) {
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > PRD_SPIN_LIMIT) {
            __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

PRD_DEV void grid_barrier_counter(unsigned* bar, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the compiler may drop the wait behind buffer_wbl2: restated)
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until_ge(bar, epoch * gridDim.x, bar + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// XCD-hierarchical form ("barrier-xcd" of the guide): a release fence writes back the dirty lines of the issuing workgroup's WHOLE L2, so one
// per XCD and barrier is enough -- 256 of them (the plain form above) cost 40 us per folding block.  Which workgroups share an L2 is read
// from the hardware (HW_REG_XCC_ID), never assumed from blockIdx: every workgroup registers with its XCD at kernel entry; at a barrier the
// LAST arriver of an XCD (all its peers' stores have been acknowledged by that L2: s_waitcnt vmcnt(0) before the arrival) releases, arrives
// on the top counter, waits for the other XCDs there and publishes the generation to its peers; everybody acquires.
PRD_DEV unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

PRD_DEV void grid_barrier_register(unsigned* bar, unsigned xcc) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar + 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the membership is counted before the registration is
        __hip_atomic_fetch_add(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

PRD_DEV void grid_barrier_xcd(unsigned* bar, unsigned epoch, unsigned xcc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every wave: its stores are in this XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* tmo = bar + 1;
        bool ok = spin_until_ge(bar + 2, gridDim.x, tmo);                   // everybody has registered (true long before the first barrier)
        const unsigned mine = __hip_atomic_load(bar + 8 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(bar + 8 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        const unsigned arrived = __hip_atomic_fetch_add(bar + 16 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (ok && arrived == mine * epoch) {                                // the last of this XCD: release for all of it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until_ge(bar + 3, nx * epoch, tmo);
            __hip_atomic_store(bar + 24 + xcc, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until_ge(bar + 24 + xcc, epoch, tmo);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// output projection + residual of the triangle attention (tri_attn_out_kernel of prd_tri.hip, split-16 form), in place on `pair`
template <int P, int NW>
PRD_DEV void tri_attn_out_body(float* pair, const float* og, const float* __restrict__ wo, const float* __restrict__ bo, long rows,
                               unsigned char* lds) {
    constexpr int KH = P / 2, NB = P / 32, HC = 64;
    float* Wl = reinterpret_cast<float*>(lds);                           // fp16 hi | lo planes of W_o (P x 64)
    float* bl = Wl + P * HC;
    stage_weight_h2<HC>(reinterpret_cast<u32x4*>(Wl), wo, P, HC, threadIdx.x, NW * 64, H2_WSCALE);
    stage_vec_cll(bl, bo, P, threadIdx.x, NW * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long ntask = (rows + 31) / 32;
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float x[HC / 2];
        load_row_cll<HC>(og + pos * HC, hi, valid, x);
        f32x16 acc[NB];
        zero_acc(acc);
        u32x4 xs[2][HC / 16];
        split2h_cll<HC / 2>(x, xs);
        rowgemm_h2<HC, NB>(reinterpret_cast<const u32x4*>(Wl), P, 0, xs, acc, r, hi);
        float pr[KH];
        load_row_cll<P>(pair + pos * P, hi, valid, pr);
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) pr[s_] = pr[s_] + (acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + bl[hi * KH + s_]);
        store_row_cll<P>(pair + pos * P, hi, valid, pr);
    }
}

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_pair_kernel(
    float* og, float* pair, const float* __restrict__ mask,
    const float* __restrict__ wq1, const float* __restrict__ wk1, const float* __restrict__ wv1, const float* __restrict__ wg1,
    const float* __restrict__ bg1, const float* __restrict__ wo1, const float* __restrict__ bo1,
    const float* __restrict__ wq2, const float* __restrict__ wk2, const float* __restrict__ wv2, const float* __restrict__ wg2,
    const float* __restrict__ bg2, int b, int N, int NP, int H, unsigned* bar, int plain_barrier) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned xcc = xcc_id();
    if (!plain_barrier) grid_barrier_register(bar, xcc);
    tri_attn_core_v3_body<P, NW, 0, true>(og, pair, mask, wq1, wk1, wv1, wg1, bg1, b, N, NP, H, 0, 0, nullptr, lds);
    if (plain_barrier) grid_barrier_counter(bar, 1u); else grid_barrier_xcd(bar, 1u, xcc);
    tri_attn_out_body<P, NW>(pair, og, wo1, bo1, (long)b * N * N, lds);
    if (plain_barrier) grid_barrier_counter(bar, 2u); else grid_barrier_xcd(bar, 2u, xcc);
    tri_attn_core_v3_body<P, NW, 0, true>(og, pair, mask, wq2, wk2, wv2, wg2, bg2, b, N, NP, H, 1, 0, nullptr, lds);
}

size_t v3_lds_bytes(int N, int P) {
    const int NP = prd_round_up(N, 32), nqb = NP / 32, m4 = nqb & 3, nhelp = m4 ? 4 - m4 : 0;
    return (size_t)64 * P * 4 + 2 * ((size_t)NP * 132 + 64) + 64 + (size_t)5 * m4 * 2048 + (size_t)2 * (nhelp + 1) * m4 * 2560;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Long rows (more query blocks than waves: 385 <= N <= 1024: K and V of a row as fp16 hi | lo planes, 128 B per position).
// Same tile arithmetic as tri_attn_core_v2_kernel; what changes is what stays resident:
//   phase 1  K (fp16 hi | lo planes) and V of ALL blocks of the row, blocks dealt round-robin to the 12 waves;
//   phase 2  a wave re-loads, normalises and projects [Q|G] of a query block itself right before that block's key sweep: the
//            lane that computes the 8 Q channels of a position is the lane that needs them as the B operand of Q K^T, and the
//            lane that computes its gate channels is the one that gates the output -- Q and the gate never touch the LDS.
//            Blocks go to the waves in rounds of 12; the last round of rem = nqb mod 12 blocks is shared when rem <= 6: the
//            owner publishes Q of its block (2 KB), G = 12 / rem waves take a G-th of the keys each, partials (reference,
//            sum, O) meet in LDS and the owner merges (nqb = 25 at N = 769: twelve waves share the 25th block).
struct V2LLds { unsigned kh, kl, v, kadd, flag, bias, tail, qs, part, plane; };
constexpr int V2L_TAIL_MAX = 4;                         // a ragged last key tile of up to this many keys is swept as rank-1 updates

PRD_DEV V2LLds v2l_layout(int P, int NP, int nshare) {
    V2LLds L;
    unsigned off = 64u * P * 4u;
    L.plane = (unsigned)NP * 16u;
    L.kh = off; off += 2 * L.plane;
    L.kl = off; off += 2 * L.plane;
    L.v = off; off += (unsigned)NP * 64u;
    L.kadd = off; off += (unsigned)NP * 4u;
    L.flag = off; off += 128u;
    L.bias = off; off += 64u;
    L.tail = off; off += (unsigned)V2L_TAIL_MAX * 128u; // [key][hi][K 8 | V 8] fp32: the keys of a ragged last tile (rank-1 sweep)
    L.qs = off; off += (unsigned)nshare * 2048u;       // [block][qh | ql][hi][32 positions][16 B]
    L.part = off;                                      // [12 pieces][10][64] fp32
    return L;
}

template <int P, int NW, bool PREFETCH, bool GV>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_v2l_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending, int flags) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int nqb = NP / 32;                            // query blocks = key tiles (13 .. 32)
    const int nfull = nqb / NW, rem = nqb - nfull * NW;
    const int G = rem ? NW / rem : 0;                   // waves per block of the last round
    const bool share = G >= 2 && (flags & 16) == 0;     // (flag 16: the host found no room for the partials -- rows near 1024)
    const V2LLds L = v2l_layout(P, NP, share ? rem : 0);
    u32x4* Wb = reinterpret_cast<u32x4*>(lds);
    float* kadd = reinterpret_cast<float*>(lds + L.kadd);
    int* tflag = reinterpret_cast<int*>(lds + L.flag);
    float* biasl = reinterpret_cast<float*>(lds + L.bias);
    float* part = reinterpret_cast<float*>(lds + L.part);
    // Ragged last key tile (flag 64, set by the host when it holds 1 .. V2L_TAIL_MAX keys and GV): its keys are NOT swept as a 32-key
    // tile step (7 MFMAs, 16 exponentials and splits per lane for 1-4 useful columns: N = 769 = 24 x 32 + 1, BASELINE configs[4], paid
    // a 25th tile step in every sweep) but as rank-1 updates in fp32 from K / V rows that phase 1 leaves un-split in `tail`:
    // s = q . k_j, p = 2^(s - ref), l += p, o += p v_j.  The tile's regular K / V planes are still written: the online redo of an
    // overflowed piece sweeps all tiles the usual way.
    const int ntail = (GV && (flags & 64)) ? N - 32 * (nqb - 1) : 0;
    const float* tailkv = reinterpret_cast<const float*>(lds + L.tail);
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
    if (tid < 16) {
        const int hh = tid >> 3, e = tid & 7;
        biasl[tid] = H2_WSCALE * NEG_LOG2E * bg[h * C + 4 * hh + (e & 3) + 8 * (e >> 2)];
    }
    const int nrows = b * N;
    struct RowIx { int bu, bb, u; };
    auto make_row = [&](int bu) { RowIx x; x.bu = bu; x.bb = bu / N; x.u = bu - x.bb * N; return x; };
    auto row_pos = [&](const RowIx& x, int v) -> long { return ending ? (long)((x.bb * N + v) * N + x.u) : (long)(x.bu * N + v); };
    const float inv16 = H2_INV_WSCALE;
    const unsigned kl_off = L.kl - L.kh;
    const unsigned kbase = L.kh + (unsigned)hi * L.plane + (unsigned)r * 16u;      // + 512 t
    // (GV: V slots XORed with 2 a + khalf, see tri_attn_core_v3_kernel)
    const unsigned vb0 = L.v + (unsigned)hi * 512u + (unsigned)(GV ? (gv_slot(r) ^ hi) : r) * 16u;          // + 2048 t
    const unsigned vb1 = L.v + 1024u + (unsigned)hi * 512u + (unsigned)(GV ? (gv_slot(r) ^ (2 + hi)) : r) * 16u;
    auto ldv = [&](int t, PBuf& p) {
        p.va0 = *reinterpret_cast<const u32x4*>(lds + vb0 + 2048u * t);
        p.va1 = *reinterpret_cast<const u32x4*>(lds + vb1 + 2048u * t);
    };
    // Work items: `nfr` whole rounds of rstride rows (one row per workgroup of this head), then the Lr < rstride rows that are left.
    // A last round of few rows would leave most of the chip idle for a whole item (N = 769: 769 = 12 x 64 + 1 rows, i.e. a 13th
    // round for ONE row: 7.7 % of the launch): its rows are SPLIT by query blocks over tparts = min(nqb, rstride / Lr) workgroups
    // each -- every part projects K / V of the whole row (phase 1) and sweeps its own range [q0, q1) of query blocks, whose
    // outputs are independent.  (flag 32 = PRD_TUNE_TA2_NO_TAIL_SPLIT: A/B switch.)
    const int nfr = nrows / rstride, Lr = nrows - nfr * rstride;
    int tparts = (Lr > 0 && !(flags & 32)) ? rstride / Lr : 1;
    tparts = tparts > nqb ? nqb : tparts;
    const int nshare_room = share ? rem : 0;            // Q slots of the shared round the host made room for
    auto item = [&](int it, int& bu_, int& q0_, int& q1_) -> bool {
        q0_ = 0; q1_ = nqb; bu_ = 0;
        if (it < nfr) { bu_ = slot + it * rstride; return true; }
        if (it > nfr || Lr == 0) return false;
        if (tparts <= 1) { bu_ = nfr * rstride + slot; return slot < Lr; }
        const int part_ = slot / Lr;
        if (part_ >= tparts) return false;
        bu_ = nfr * rstride + (slot - part_ * Lr);
        q0_ = (nqb * part_) / tparts; q1_ = (nqb * (part_ + 1)) / tparts;
        return true;
    };

    float xnext[PREFETCH ? KH : 1];                     // the wave's FIRST phase-1 block of the next row
    float mknext = 0.f, munext = 0.f;
    int bu, q0, q1;
    bool have = item(0, bu, q0, q1);
    RowIx rnext = make_row(have ? bu : 0);
    if constexpr (PREFETCH) {
        const int v = wave * 32 + r;
        const bool ok = have && v < N;
        load_row_cll<P>(pair + row_pos(rnext, ok ? v : 0) * P, hi, ok, xnext);
        if (ok) mknext = mask[rnext.bb * N + v];
    }
    if (have) munext = mask[bu];
    // Shared last round (flag 128 = its round-5 form, A/B): the [Q|G] projection of a shared block is done INSIDE phase 1 by one of the
    // waves that have a block less to project there (N = 769: 25 blocks on 12 waves -- wave 0 projects three, the others two), the
    // key pieces are swept FIRST in phase 2 (Q is in LDS behind phase 1's closing barrier) and the merge waits behind the barrier
    // that opens the NEXT row anyway: no projection with eleven waves idle, no two extra barriers per row.
    const bool early_share = (flags & 128) == 0;
    bool last_shared = false;                           // (workgroup-uniform) the item just finished had a shared round in the early form
    bool m_pending = false;                             // a merge of the previous item is due (wave-uniform; the gate is in m_gate)
    int m_blk = 0, m_G = 0, m_j = 0;
    RowIx m_row = make_row(0);
    float m_gate[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m_gate[e] = 0.f;
    auto store_out = [&](const RowIx& rw, int qb, const float (&o)[8], float l, const float (&gate)[8]) {
        const float ltot = xhalf_add(l);
        const int v = 32 * qb + r;
        if (v < N) {
            const float il = 1.0f / (VSCALE * ltot);
            float* dst = og + row_pos(rw, v) * HC + h * C + 4 * hi;
            *reinterpret_cast<float4*>(dst) = make_float4(gate[0] * (o[0] * il), gate[1] * (o[1] * il), gate[2] * (o[2] * il), gate[3] * (o[3] * il));
            *reinterpret_cast<float4*>(dst + 8) = make_float4(gate[4] * (o[4] * il), gate[5] * (o[5] * il), gate[6] * (o[6] * il), gate[7] * (o[7] * il));
        }
    };
    // merge of the m_G key pieces of shared block m_j (flash-decoding merge), gated and stored
    auto merge_shared = [&](const RowIx& rw, int j, int G_, int qb, const float (&gate)[8]) {
        float M = -INFINITY;
        for (int k = 0; k < G_; ++k) {
            const bool has = (nqb * (k + 1)) / G_ > (nqb * k) / G_;
            if (has) M = max2f(M, part[(size_t)(j * G_ + k) * 640 + 9 * 64 + lane]);
        }
        float o[8], l = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
        for (int k = 0; k < G_; ++k) {
            const bool has = (nqb * (k + 1)) / G_ > (nqb * k) / G_;
            if (!has) continue;
            const float* pp = part + (size_t)(j * G_ + k) * 640 + lane;
            const float scl = __builtin_amdgcn_exp2f(pp[9 * 64] - M);
            l += scl * pp[8 * 64];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o[jj] += scl * pp[jj * 64];
        }
        store_out(rw, qb, o, l, gate);
    };
    for (int it = 0; have; ++it) {
        const RowIx row = rnext;
        // the query blocks [q0, q1) of this item in rounds of NW; a last round of rem_ blocks shared by G_ waves each
        const int nb_ = q1 - q0, nfull_ = nb_ / NW, rem_ = nb_ - nfull_ * NW;
        const int G_ = rem_ ? NW / rem_ : 0;
        const bool share_ = G_ >= 2 && rem_ <= nshare_room;
        // key tiles this wave sweeps in the shared round
        const int pj = share_ ? wave / G_ : 0, pp_ = share_ ? wave - pj * G_ : 0;
        const int pT0 = share_ ? (nqb * pp_) / G_ : 0, pT1 = share_ ? (nqb * (pp_ + 1)) / G_ : 0;
        const int work_tot = nfull_ * nqb + (share_ ? (wave < rem_ * G_ ? pT1 - pT0 : 0) : (wave < rem_ ? nqb : 0));   // (priorities only)
        __syncthreads();                                // previous row's LDS consumed (and the weight image staged)
        if (m_pending) {                                // (wave-uniform) the previous item's shared block: its pieces are all in LDS now
            merge_shared(m_row, m_j, m_G, m_blk, m_gate);
            m_pending = false;
        }
        const float mu = munext;
        int r1 = r, hi1 = hi;                           // opaque per row (see tri_attn_core_v2_kernel)
        asm volatile("" : "+v"(r1), "+v"(hi1));
        auto wop = [&](int wrow, int s_, u32x4& wh, u32x4& wl) {
            const int slot_ = h2_slot<P>(wrow, 2 * s_ + hi1);
            wh = Wb[(size_t)wrow * (P / 8) + slot_];
            wl = Wb[(size_t)(64 + wrow) * (P / 8) + slot_];
        };
        // [Q|G] of query block qb: Q as the B operands of Q K^T (fp16 hi | lo), the lane's 8 gate channels
        auto project_qg = [&](int qb, u32x4& qh4, u32x4& ql4, float (&gate)[8]) {
            float x[KH];
            const int v = qb * 32 + r1;
            const bool ok = v < N;
            load_row_cll<P>(pair + row_pos(row, ok ? v : 0) * P, hi1, ok, x);
            ln_cll_p<KH>(x);
            u32x4 xs[2][P / 16];
            split2h_rn_cll<KH>(x, xs);
            f32x16 acc;
            {
                const float4 b0 = *reinterpret_cast<const float4*>(biasl + 8 * hi1), b1 = *reinterpret_cast<const float4*>(biasl + 8 * hi1 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
                acc[8] = b0.x; acc[9] = b0.y; acc[10] = b0.z; acc[11] = b0.w; acc[12] = b1.x; acc[13] = b1.y; acc[14] = b1.z; acc[15] = b1.w;
            }
#pragma unroll
            for (int s_ = 0; s_ < P / 16; ++s_) {
                u32x4 wh, wl;
                wop(16 + r1, s_, wh, wl);               // image rows 16-47 = Q | G
                acc = mfma_h(wh, xs[0][s_], acc);
                acc = mfma_h(wh, xs[1][s_], acc);
                acc = mfma_h(wl, xs[0][s_], acc);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] *= inv16;
            split8_rn(acc, 0, qh4, ql4);
#pragma unroll
            for (int e = 0; e < 8; ++e) gate[e] = gate_from_scaled(acc[8 + e]);
        };
        // ================= phase 1: K and V of every block =================
        for (int blk = wave; blk < nqb; blk += NW) {
            float x[KH];
            float mk = 0.f;
            bool from_prefetch = false;
            if constexpr (PREFETCH) {
                if (blk == wave) {
#pragma unroll
                    for (int k = 0; k < KH; ++k) x[k] = xnext[k];
                    mk = mknext;
                    from_prefetch = true;
                }
            }
            if (!from_prefetch) {
                const int v = blk * 32 + r1;
                const bool ok = v < N;
                load_row_cll<P>(pair + row_pos(row, ok ? v : 0) * P, hi1, ok, x);
                mk = ok ? mask[row.bb * N + v] : 0.f;
            }
            ln_cll_p<KH>(x);
            u32x4 xs[2][P / 16];
            split2h_rn_cll<KH>(x, xs);
            {   // logit override of masked / padded keys + tile flag
                const int v = blk * 32 + r1;
                const bool valid = v < N;
                const bool keep = valid && (mu * mk >= 0.5f);
                if (hi1 == 0) kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E_2 : -INFINITY);
                const bool any_override = __any(!keep);
                if (lane == 0) tflag[blk] = any_override ? 1 : 0;
            }
            if constexpr (GV) {
                // [K|V] as ONE unswapped row GEMM (image rows 0-15 | 48-63): registers 0-7 = the lane's 8 K channels (as before),
                // 8-15 = 8 V channels of its position, stored transposed (see tri_attn_core_v3_kernel): 12 MFMAs instead of 24
                f32x16 akv;
#pragma unroll
                for (int e = 0; e < 16; ++e) akv[e] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(r1 < 16 ? r1 : 32 + r1, s_, wh, wl);
                    akv = mfma_h(wh, xs[0][s_], akv);
                    akv = mfma_h(wh, xs[1][s_], akv);
                    akv = mfma_h(wl, xs[0][s_], akv);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) akv[e] *= inv16;
                u32x4 kh4, kl4;
                split8_rn(akv, 0, kh4, kl4);
                const unsigned po = (unsigned)hi1 * L.plane + (unsigned)(blk * 32 + r1) * 16u;
                *reinterpret_cast<u32x4*>(lds + L.kh + po) = kh4;
                *reinterpret_cast<u32x4*>(lds + L.kl + po) = kl4;
                if (ntail > 0 && blk == nqb - 1 && r1 < ntail) {        // the lane's 8 K channels and 8 V channels (x 16) of a tail key, fp32
                    float* tp = reinterpret_cast<float*>(lds + L.tail) + (r1 * 2 + hi1) * 16;
                    *reinterpret_cast<float4*>(tp) = make_float4(akv[0], akv[1], akv[2], akv[3]);
                    *reinterpret_cast<float4*>(tp + 4) = make_float4(akv[4], akv[5], akv[6], akv[7]);
                    *reinterpret_cast<float4*>(tp + 8) = make_float4(akv[8], akv[9], akv[10], akv[11]);
                    *reinterpret_cast<float4*>(tp + 12) = make_float4(akv[12], akv[13], akv[14], akv[15]);
                }
                const bool odd = (r1 & 1) != 0;
                const int kp0 = r1 & 30;
                const int a_ = kp0 >> 4, kk = kp0 & 15, kh_ = (kk >> 2) & 1, w_ = (kk >> 3) * 2 + ((kk & 3) >> 1);
                const int ch0 = 4 * hi1 + (odd ? 8 : 0);
                const unsigned vo = L.v + (unsigned)blk * 2048u + (unsigned)a_ * 1024u + (unsigned)kh_ * 512u + (unsigned)w_ * 4u;
                const int sx = 2 * a_ + kh_;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mine_lo = akv[8 + j], mine_hi = akv[12 + j];
                    const float give = odd ? mine_lo : mine_hi;
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, false));
                    const float ka = odd ? got : mine_lo, kb = odd ? mine_hi : got;
                    unsigned hh, ll;
                    split2h_rn(ka, kb, hh, ll);
                    const unsigned so = (unsigned)(gv_slot(ch0 + j) ^ sx) * 16u;
                    *reinterpret_cast<unsigned*>(lds + vo + so) = hh;
                    *reinterpret_cast<unsigned*>(lds + vo + 256u + so) = ll;
                }
            } else {
                f32x16 acc, av;
    #pragma unroll
                for (int e = 0; e < 16; ++e) { acc[e] = 0.f; av[e] = 0.f; }
    #pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl, vh, vl;
                    wop(r1, s_, wh, wl);                    // rows 0-31 = K | Q (Q discarded here)
                    wop(48 + (r1 & 15), s_, vh, vl);
                    acc = mfma_h(wh, xs[0][s_], acc);
                    av = mfma_h(xs[0][s_], vh, av);
                    acc = mfma_h(wh, xs[1][s_], acc);
                    av = mfma_h(xs[1][s_], vh, av);
                    acc = mfma_h(wl, xs[0][s_], acc);
                    av = mfma_h(xs[0][s_], vl, av);
                }
    #pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] *= inv16;
                u32x4 kh4, kl4;
                split8_rn(acc, 0, kh4, kl4);
                const unsigned po = (unsigned)hi1 * L.plane + (unsigned)(blk * 32 + r1) * 16u;
                *reinterpret_cast<u32x4*>(lds + L.kh + po) = kh4;
                *reinterpret_cast<u32x4*>(lds + L.kl + po) = kl4;
                u32x4 vh0, vl0, vh1, vl1;                   // V stays x 16
                split8_rn(av, 0, vh0, vl0);
                split8_rn(av, 8, vh1, vl1);
                const bool lo_lane = r1 >= 16;
                u32x4 s0, s1;
    #pragma unroll
                for (int w = 0; w < 4; ++w) { s0[w] = lo_lane ? vl0[w] : vh0[w]; s1[w] = lo_lane ? vl1[w] : vh1[w]; }
                const unsigned vo = L.v + (unsigned)(blk * 4 + hi1) * 512u + (unsigned)r1 * 16u;
                *reinterpret_cast<u32x4*>(lds + vo) = s0;
                *reinterpret_cast<u32x4*>(lds + vo + 1024u) = s1;
                    }
        }
        // shared block j of this item: projected here by wave NW - 1 - j (those waves have the fewest blocks above), Q published for the
        // waves that sweep its key pieces, the gate kept by the projecting wave, which merges after the next barrier but one
        const bool sh_early = share_ && early_share;
        const int sh_j = NW - 1 - wave;                 // the shared block this wave owns (if < rem_)
        if (sh_early && sh_j < rem_) {
            u32x4 qh4, ql4;
            project_qg(q0 + nfull_ * NW + sh_j, qh4, ql4, m_gate);
            const unsigned qo = L.qs + (unsigned)sh_j * 2048u + (unsigned)hi * 512u + (unsigned)r * 16u;
            *reinterpret_cast<u32x4*>(lds + qo) = qh4;
            *reinterpret_cast<u32x4*>(lds + qo + 1024u) = ql4;
        }
        __syncthreads();
        int bun, q0n, q1n;
        const bool haven = item(it + 1, bun, q0n, q1n);
        {   // the wave's first phase-1 block of the next row: in flight during the key loops
            rnext = make_row(haven ? bun : 0);
            if constexpr (PREFETCH) {
                const int v = wave * 32 + r;
                const bool ok = haven && v < N;
                load_row_cll<P>(pair + row_pos(rnext, ok ? v : 0) * P, hi, ok, xnext);
                mknext = ok ? mask[rnext.bb * N + v] : 0.f;
            }
            munext = haven ? mask[bun] : 0.f;
        }
        // ================= phase 2 =================
        unsigned fmask;
        {
            const int f = lane < nqb ? tflag[lane] : 0;
            fmask = (unsigned)__ballot(f != 0);
        }
        int work_rem = work_tot;
        // key tiles [T0, T1) for the queries (qh, ql): o8 = O (x 16, relative to mref), lsum = the lane's part of the row sum
        auto run_piece = [&](const u32x4& qh, const u32x4& ql, int T0, int T1, float (&o8)[8], float& lsum, float& mref) {
            f32x16 o0, zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] = 0.f; zero[e] = 0.f; }
            lsum = 0.f;
            bool big = false;
            if (flags & 1) v2_prio(work_rem, work_tot);
            // the ragged last tile as rank-1 updates (a piece that consists of that tile alone has no reference yet: swept as a tile)
            const bool tail1 = ntail > 0 && T1 == nqb && T1 - 1 > T0;
            const int T1m = tail1 ? T1 - 1 : T1;
            {
                KOp k = load_k(lds, kbase + 512u * T0, kl_off);
                f32x16 s0 = qk_tile(k, qh, ql, zero);
                if (T0 + 1 < T1m) k = load_k(lds, kbase + 512u * (T0 + 1), kl_off);
                if ((fmask >> T0) & 1) mask_tile_at(lds, L.kadd, T0, hi, 0.f, s0);
                const float tmax = xhalf_max(max16_mfma(s0));
                mref = tmax - P_SHIFT;
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) { negm[e] = -mref; s0[e] -= mref; }
                PBuf p;
                ldv(T0, p);
                exp_split(s0, lsum, big, p);
                pv_tile(p, o0);
                for (int t = T0 + 1; t < T1m; ++t) {
                    if ((flags & 1) && ((t - T0) & 3) == 0) v2_prio(work_rem - (t - T0), work_tot);
                    f32x16 s = qk_tile(k, qh, ql, negm);
                    if (t + 1 < T1m) k = load_k(lds, kbase + 512u * (t + 1), kl_off);
                    ldv(t, p);
                    if ((fmask >> t) & 1) mask_tile_at(lds, L.kadd, t, hi, mref, s);
                    exp_split(s, lsum, big, p);
                    pv_tile(p, o0);
                }
            }
            if (tail1) {
                float qf[8];                            // the lane's 8 query channels in fp32 (hi + lo parts of the B operands)
                {   // (explicit half-word extraction: indexing the operand vector inside an unrolled loop and bit-casting the word to a
                    // 2 x fp16 vector compiled to four copies of word 0 with hipcc 7.2 -- found by the parity tests)
                    auto lo16 = [](unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); };
                    auto hi16 = [](unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); };
                    const unsigned h0 = qh[0], h1 = qh[1], h2 = qh[2], h3 = qh[3], l0 = ql[0], l1 = ql[1], l2 = ql[2], l3 = ql[3];
                    qf[0] = lo16(h0) + lo16(l0); qf[1] = hi16(h0) + hi16(l0);
                    qf[2] = lo16(h1) + lo16(l1); qf[3] = hi16(h1) + hi16(l1);
                    qf[4] = lo16(h2) + lo16(l2); qf[5] = hi16(h2) + hi16(l2);
                    qf[6] = lo16(h3) + lo16(l3); qf[7] = hi16(h3) + hi16(l3);
                }
                for (int j = 0; j < ntail; ++j) {
                    const float* tk = tailkv + (j * 2 + hi) * 16;
                    const float4 k0 = *reinterpret_cast<const float4*>(tk), k1 = *reinterpret_cast<const float4*>(tk + 4);
                    float sh = qf[0] * k0.x;
                    sh = __builtin_fmaf(qf[1], k0.y, sh); sh = __builtin_fmaf(qf[2], k0.z, sh); sh = __builtin_fmaf(qf[3], k0.w, sh);
                    sh = __builtin_fmaf(qf[4], k1.x, sh); sh = __builtin_fmaf(qf[5], k1.y, sh); sh = __builtin_fmaf(qf[6], k1.z, sh);
                    sh = __builtin_fmaf(qf[7], k1.w, sh);
                    const float sd = xhalf_add(sh);     // (both halves hold the whole dot product now)
                    const float ka = kadd[32 * (nqb - 1) + j];
                    const float sj = ((ka == 0.f) ? sd : ka) - mref;
                    const float pj = __builtin_amdgcn_exp2f(sj);
                    if (hi == 0) lsum += pj;            // (the halves of a query add their row sums in finish / the merge)
                    const float4 v0 = *reinterpret_cast<const float4*>(tk + 8), v1 = *reinterpret_cast<const float4*>(tk + 12);
                    o0[0] = __builtin_fmaf(pj, v0.x, o0[0]); o0[1] = __builtin_fmaf(pj, v0.y, o0[1]);
                    o0[2] = __builtin_fmaf(pj, v0.z, o0[2]); o0[3] = __builtin_fmaf(pj, v0.w, o0[3]);
                    o0[4] = __builtin_fmaf(pj, v1.x, o0[4]); o0[5] = __builtin_fmaf(pj, v1.y, o0[5]);
                    o0[6] = __builtin_fmaf(pj, v1.z, o0[6]); o0[7] = __builtin_fmaf(pj, v1.w, o0[7]);
                }
            }
            if (__any(big || !(lsum < 3.0e38f))) {      // rare: redo the piece with the online update in every tile
#pragma unroll
                for (int e = 0; e < 16; ++e) o0[e] = 0.f;
                lsum = 0.f;
                float m_run = -1e30f;
                for (int t = T0; t < T1; ++t) {
                    const KOp k = load_k(lds, kbase + 512u * t, kl_off);
                    f32x16 s = qk_tile(k, qh, ql, zero);
                    if ((fmask >> t) & 1) mask_tile_at(lds, L.kadd, t, hi, 0.f, s);
                    const float m_new = max2f(m_run, xhalf_max(max16_mfma(s)));
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    mref = m_new - P_SHIFT;
                    lsum *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; s[e] -= mref; }
                    bool dummy = false;
                    PBuf p;
                    ldv(t, p);
                    exp_split(s, lsum, dummy, p);
                    pv_tile(p, o0);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o8[jj] = o0[jj] + o0[jj + 8];
            work_rem -= T1 - T0;
        };
        auto finish = [&](int qb, const float (&o)[8], float l, const float (&gate)[8]) {
            const float ltot = xhalf_add(l);
            const int v = 32 * qb + r;
            if (v < N) {
                const float il = 1.0f / (VSCALE * ltot);
                float* dst = og + row_pos(row, v) * HC + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(gate[0] * (o[0] * il), gate[1] * (o[1] * il), gate[2] * (o[2] * il), gate[3] * (o[3] * il));
                *reinterpret_cast<float4*>(dst + 8) = make_float4(gate[4] * (o[4] * il), gate[5] * (o[5] * il), gate[6] * (o[6] * il), gate[7] * (o[7] * il));
            }
        };
        // ---- shared last round, early form: the key pieces first; the merge follows the barrier that opens the next item ----
        if (sh_early) {
            if (wave < rem_ * G_ && pT1 > pT0) {
                const unsigned qo = L.qs + (unsigned)pj * 2048u + (unsigned)hi * 512u + (unsigned)r * 16u;
                const u32x4 qh4 = *reinterpret_cast<const u32x4*>(lds + qo), ql4 = *reinterpret_cast<const u32x4*>(lds + qo + 1024u);
                float o8[8], lsum, mref;
                run_piece(qh4, ql4, pT0, pT1, o8, lsum, mref);
                float* pp = part + (size_t)wave * 640 + lane;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pp[jj * 64] = o8[jj];
                pp[8 * 64] = lsum;
                pp[9 * 64] = mref;
            }
            if (sh_j < rem_) { m_pending = true; m_row = row; m_j = sh_j; m_G = G_; m_blk = q0 + nfull_ * NW + sh_j; }
        }
        // ---- whole rounds (and an unshared last round): one query block per wave ----
        const int nrounds = nfull_ + ((rem_ && !share_) ? 1 : 0);
        for (int rd = 0; rd < nrounds; ++rd) {
            const int qb = q0 + rd * NW + wave;
            if (qb < q1) {
                u32x4 qh4, ql4;
                float gate[8], o8[8], lsum, mref;
                project_qg(qb, qh4, ql4, gate);
                run_piece(qh4, ql4, 0, nqb, o8, lsum, mref);
                finish(qb, o8, lsum, gate);
            }
        }
        // ---- shared last round, round-5 form (flag 128): projection, barrier, pieces, barrier, merge ----
        if (share_ && !sh_early) {
            float gate[8];
            if (wave < rem_) {
                u32x4 qh4, ql4;
                project_qg(q0 + nfull_ * NW + wave, qh4, ql4, gate);
                const unsigned qo = L.qs + (unsigned)wave * 2048u + (unsigned)hi * 512u + (unsigned)r * 16u;
                *reinterpret_cast<u32x4*>(lds + qo) = qh4;
                *reinterpret_cast<u32x4*>(lds + qo + 1024u) = ql4;
            }
            __syncthreads();
            if (wave < rem_ * G_ && pT1 > pT0) {
                const unsigned qo = L.qs + (unsigned)pj * 2048u + (unsigned)hi * 512u + (unsigned)r * 16u;
                const u32x4 qh4 = *reinterpret_cast<const u32x4*>(lds + qo), ql4 = *reinterpret_cast<const u32x4*>(lds + qo + 1024u);
                float o8[8], lsum, mref;
                run_piece(qh4, ql4, pT0, pT1, o8, lsum, mref);
                float* pp = part + (size_t)wave * 640 + lane;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pp[jj * 64] = o8[jj];
                pp[8 * 64] = lsum;
                pp[9 * 64] = mref;
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (wave < rem_) {                          // merge the G_ pieces of the wave's block (flash-decoding merge)
                float M = -INFINITY;
                for (int k = 0; k < G_; ++k) {
                    const bool has = (nqb * (k + 1)) / G_ > (nqb * k) / G_;
                    if (has) M = max2f(M, part[(size_t)(wave * G_ + k) * 640 + 9 * 64 + lane]);
                }
                float o[8], l = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
                for (int k = 0; k < G_; ++k) {
                    const bool has = (nqb * (k + 1)) / G_ > (nqb * k) / G_;
                    if (!has) continue;
                    const float* pp = part + (size_t)(wave * G_ + k) * 640 + lane;
                    const float scl = __builtin_amdgcn_exp2f(pp[9 * 64] - M);
                    l += scl * pp[8 * 64];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) o[jj] += scl * pp[jj * 64];
                }
                finish(q0 + nfull_ * NW + wave, o, l, gate);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        last_shared = sh_early;
        have = haven; bu = bun; q0 = q0n; q1 = q1n;
    }
    if (last_shared) {                                  // (workgroup-uniform) the last item's shared block
        __syncthreads();
        if (m_pending) merge_shared(m_row, m_j, m_G, m_blk, m_gate);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Triangle attention BACKWARD core in split-16 arithmetic (the fp32-MFMA form is tri_attn_bwd_core_kernel, prd_bwd.hip; the
// reference reaches this through autograd of modules.py:236-243 -> 185-225).  One persistent workgroup of 12 waves per CU; a
// work item is one (pair row, head); wave w owns the 32-position block w of the row: it projects it, sweeps it as QUERIES
// (pass A: d q) and as KEYS (pass B: d k, d v).  Given  dog = d(gated head output)  and the saved  og = gate * o  of the forward:
//
//   phase 1   LayerNorm + split of the block's rows, then four row GEMMs on the 32x32x16 fp16 MFMA:
//               [K|Q] unswapped : lane (position, hi) gets its 8 K and 8 Q channels = the B operands of the logit products and,
//                                 stored as fp16 hi | lo planes, their A operands (K rows now, Q rows after pass A)
//               [V|G] unswapped : V rows likewise (A operand of dP = do V^T), and the gate of the position:
//                                 do = dog * gate,  delta = sum_c dog_c og_c (= do . o),  d gate_pre = dog og (1 - gate)
//               [K|Q] swapped   : lane (channel, hi) gets 16 positions of one channel = the position-major A operands
//                                 [X_hi; X_lo] of the products that contract over positions (dq = dS^T K, dk = dS Q)
//               [G|G] swapped   : the gate by channel, for do^T = (dog gate)^T, the A operand of dv = P^T do
//   sweep 0   softmax statistics of the block's queries (logits only): lse_q
//   pass A    per 32-key tile: S^T = K Q^T - lse, dP^T = V do^T - delta (both constants ride in the accumulator), p = 2^S,
//             dS = p dP, split, dq^T += [K^T_hi; K^T_lo] dS  (two MFMAs per 16 keys give all four hi/lo products)
//   pass B    Q rows / do rows replace K rows / V rows in LDS (from the registers they were left in); per 32-query tile:
//             S, dP again with the lane's key as the B operand, p and dS split, dk^T += [Q^T; ..] dS, dv^T += [do^T; ..] p
//
// Scaling (all by powers of two, taken out at the stores): probabilities x 2^5; do of position q x 2^e_q with e_q from the
// largest |do_q| (gradients are tiny and fp16 has 5 exponent bits); pass B folds 2^(E - e_q), E = min_q e_q, into the
// probability exponent so that every dS, p of a key column carries the common factor 2^(5 + E).
struct B2Lds { unsigned krow, vrow, kt, qt, dt, lse, mq, cq, delta, esc, kadd, flag, bias, red, total, plane; };

__host__ __device__ inline B2Lds b2_layout(int P, int NP) {
    B2Lds L;
    unsigned off = 64u * (unsigned)P * 4u;              // weight image: rows K | Q | V | G, fp16 hi | lo planes
    L.plane = (unsigned)NP * 16u;
    L.krow = off; off += 4u * L.plane;                  // hi planes (lane half 0 | 1), lo planes; Q rows in pass B
    L.vrow = off; off += 4u * L.plane;                  // V rows; do rows in pass B
    L.kt = off; off += (unsigned)NP * 64u;              // [tile][16-key half][hi][m = plane * 16 + channel][8 fp16]
    L.qt = off; off += (unsigned)NP * 64u;
    L.dt = off; off += (unsigned)NP * 64u;
    L.lse = off; off += (unsigned)NP * 4u;              // m + c of the query (accumulator preload), and the two terms apart:
    L.mq = off; off += (unsigned)NP * 4u;               // a replaced logit is (fill - m) - c, exact where fill = m (fully masked row)
    L.cq = off; off += (unsigned)NP * 4u;
    L.delta = off; off += (unsigned)NP * 4u;
    L.esc = off; off += (unsigned)NP * 4u;
    L.kadd = off; off += (unsigned)NP * 4u;
    L.flag = off; off += 64u;
    L.bias = off; off += 64u;
    L.red = off; off += 64u;
    L.total = off;
    return L;
}

constexpr float B2_PSHIFT = 5.0f;

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_bwd_core_v2_kernel(
    float* __restrict__ dqkvg, const float* __restrict__ dog, const float* __restrict__ ogs, const float* __restrict__ pair,
    const float* __restrict__ mask, const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int NP, int H, int ending,
    const float* __restrict__ lse_in, float* __restrict__ x_out) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const B2Lds L = b2_layout(P, NP);
    u32x4* Wb = reinterpret_cast<u32x4*>(lds);
    float* biasl = reinterpret_cast<float*>(lds + L.bias);
    int* redl = reinterpret_cast<int*>(lds + L.red);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int nqb = NP / 32;
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
    const int nrows = b * N;
    const float inv16 = H2_INV_WSCALE;
    const unsigned kl_rel = 2u * L.plane;
    const bool active = wave < nqb;
    const int blk = wave;
    const int r_w = r, hi_w = hi;
    f32x16 zero;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero[e] = 0.f;

    for (int bu = slot; bu < nrows; bu += rstride) {
        // lane coordinates opaque per item: the lane-dependent LDS offsets are three VALU instructions per item; hoisted out of the
        // item loop, two of the kernel's lane constants went to scratch at the 168-register limit (12 B / lane)
        int r = r_w, hi = hi_w;
        asm volatile("" : "+v"(r), "+v"(hi));
        const unsigned krow_lane = L.krow + (unsigned)hi * L.plane + (unsigned)r * 16u;      // + 512 t
        const unsigned vrow_lane = L.vrow + (unsigned)hi * L.plane + (unsigned)r * 16u;
        const unsigned tlane = (unsigned)hi * 512u + (unsigned)r * 16u;                      // + region + 2048 t (+ 1024: second half)
        const int bb = bu / N, u = bu - bb * N;
        const int pos0 = ending ? bb * N * N + u : bu * N, pstride = ending ? N : 1;       // positions of the row: pos0 + v * pstride
        auto row_pos = [&](int v) -> long { return (long)(pos0 + v * pstride); };
        __syncthreads();                                // the previous item is done with the LDS (first pass: the weight image)
        u32x4 kh4 = {0u, 0u, 0u, 0u}, kl4 = kh4, qh4 = kh4, ql4 = kh4, vh4 = kh4, vl4 = kh4, dh4 = kh4, dl4 = kh4;
        float delta_s = 0.f;
        int e_q = 127;
        const int v = blk * 32 + r;
        const bool valid = active && v < N;
        // ================= phase 1 =================
        if (active) {
            int r1 = r, hi1 = hi;                       // opaque (see tri_attn_core_v2_kernel)
            asm volatile("" : "+v"(r1), "+v"(hi1));
            auto wop = [&](int wrow, int s_, u32x4& wh, u32x4& wl) {
                const int slot_ = h2_slot<P>(wrow, 2 * s_ + hi1);
                wh = Wb[(size_t)wrow * (P / 8) + slot_];
                wl = Wb[(size_t)(64 + wrow) * (P / 8) + slot_];
            };
            float x[KH];
            load_row_cll<P>(pair + row_pos(valid ? v : 0) * P, hi1, valid, x);
            const float mu = mask[bu];
            const float mk = valid ? mask[bb * N + v] : 0.f;
            // the lane's 8 channels {4hi+e, 8+4hi+e} of dog and og (issued early: consumed after the second GEMM)
            float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, o0 = g0, o1 = g0;
            if (valid) {
                const float* dp_ = dog + row_pos(v) * HC + h * C + 4 * hi1;
                const float* op_ = ogs + row_pos(v) * HC + h * C + 4 * hi1;
                g0 = *reinterpret_cast<const float4*>(dp_);
                g1 = *reinterpret_cast<const float4*>(dp_ + 8);
                o0 = *reinterpret_cast<const float4*>(op_);
                o1 = *reinterpret_cast<const float4*>(op_ + 8);
            }
            ln_cll_p<KH>(x);
            if (x_out && h == 0) store_row_cll<P>(x_out + row_pos(valid ? v : 0) * P, hi1, valid, x);    // LN(pair) for the weight gradients
            u32x4 xs[2][P / 16];
            split2h_rn_cll<KH>(x, xs);
            {   // logit override of masked / padded keys + tile flag
                const bool keep = valid && (mu * mk >= 0.5f);
                if (hi1 == 0) reinterpret_cast<float*>(lds + L.kadd)[v] = keep ? 0.f : (v < N ? -32768.0f * LOG2E_2 : -INFINITY);
                const bool any_override = __any(!keep);
                if (lane == 0) reinterpret_cast<int*>(lds + L.flag)[blk] = any_override ? 1 : 0;
            }
            {   // [K|Q] unswapped
                f32x16 acc = zero;
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(r1, s_, wh, wl);
                    acc = mfma_h(wh, xs[0][s_], acc);
                    acc = mfma_h(wh, xs[1][s_], acc);
                    acc = mfma_h(wl, xs[0][s_], acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] *= inv16;
                split8_rn(acc, 0, kh4, kl4);
                split8_rn(acc, 8, qh4, ql4);
                const unsigned po = L.krow + (unsigned)hi1 * L.plane + (unsigned)(blk * 32 + r1) * 16u;
                *reinterpret_cast<u32x4*>(lds + po) = kh4;
                *reinterpret_cast<u32x4*>(lds + po + kl_rel) = kl4;
            }
            float gate8[8];
            {   // [V|G] unswapped
                f32x16 acc = zero;
                {
                    const float4 b0 = *reinterpret_cast<const float4*>(biasl + 4 * hi1), b1 = *reinterpret_cast<const float4*>(biasl + 8 + 4 * hi1);
                    acc[8] = b0.x; acc[9] = b0.y; acc[10] = b0.z; acc[11] = b0.w;
                    acc[12] = b1.x; acc[13] = b1.y; acc[14] = b1.z; acc[15] = b1.w;
                }
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(32 + r1, s_, wh, wl);
                    acc = mfma_h(wh, xs[0][s_], acc);
                    acc = mfma_h(wh, xs[1][s_], acc);
                    acc = mfma_h(wl, xs[0][s_], acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] *= inv16;
                split8_rn(acc, 0, vh4, vl4);
                const unsigned po = L.vrow + (unsigned)hi1 * L.plane + (unsigned)(blk * 32 + r1) * 16u;
                *reinterpret_cast<u32x4*>(lds + po) = vh4;
                *reinterpret_cast<u32x4*>(lds + po + kl_rel) = vl4;
#pragma unroll
                for (int e = 0; e < 8; ++e) gate8[e] = gate_from_scaled(acc[8 + e]);
            }
            {   // do, delta, d(gate pre-activation), the position's exponent
                const float dg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                float dov[8], dgp[8], dsum = 0.f, mx = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = dg[e] * ov[e];
                    dsum += t;
                    dgp[e] = t * (1.0f - gate8[e]);
                    dov[e] = dg[e] * gate8[e];
                    mx = __builtin_fmaxf(mx, __builtin_fabsf(dov[e]));
                }
                dsum = xhalf_add(dsum);
                mx = xhalf_max(mx);
                e_q = (mx > 0.f && mx < 3.0e38f) ? -__builtin_amdgcn_frexp_expf(mx) : 127;
                if (e_q > 126) e_q = (mx > 0.f && mx < 3.0e38f) ? 126 : 127;
                if (valid) {
                    float* gp = dqkvg + row_pos(v) * (4 * HC) + 3 * HC + h * C + 4 * hi1;
                    *reinterpret_cast<float4*>(gp) = make_float4(dgp[0], dgp[1], dgp[2], dgp[3]);
                    *reinterpret_cast<float4*>(gp + 8) = make_float4(dgp[4], dgp[5], dgp[6], dgp[7]);
                }
                const int es = e_q == 127 ? 0 : e_q;
                delta_s = __builtin_ldexpf(dsum, es);
                f32x16 d16 = zero;
#pragma unroll
                for (int e = 0; e < 8; ++e) d16[e] = __builtin_ldexpf(dov[e], es);
                split8_rn(d16, 0, dh4, dl4);
                if (hi1 == 0) reinterpret_cast<int*>(lds + L.esc)[v] = es;
            }
            {   // [K|Q] swapped: lane n < 16 = K channel n, n >= 16 = Q channel n - 16, registers = positions
                f32x16 acc = zero;
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(r1, s_, wh, wl);
                    acc = mfma_h(xs[0][s_], wh, acc);
                    acc = mfma_h(xs[1][s_], wh, acc);
                    acc = mfma_h(xs[0][s_], wl, acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] *= inv16;
                u32x4 h0, l0, h1, l1;
                split8_rn(acc, 0, h0, l0);
                split8_rn(acc, 8, h1, l1);
                const unsigned base = (r1 < 16 ? L.kt : L.qt) + (unsigned)(blk * 4 + hi1) * 512u + (unsigned)(r1 & 15) * 16u;
                *reinterpret_cast<u32x4*>(lds + base) = h0;
                *reinterpret_cast<u32x4*>(lds + base + 1024u) = h1;
                *reinterpret_cast<u32x4*>(lds + base + 256u) = l0;
                *reinterpret_cast<u32x4*>(lds + base + 256u + 1024u) = l1;
            }
            {   // [G|G] swapped -> do^T: lanes 0-15 keep the hi parts, lanes 16-31 the lo parts of the same 16 channels
                const int ch = r1 & 15;
                float dgt[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int pv = blk * 32 + 8 * (j >> 2) + 4 * hi1 + (j & 3);
                    const float t = dog[row_pos(pv < N ? pv : N - 1) * HC + h * C + ch];       // unconditional loads, all in flight
                    dgt[j] = pv < N ? t : 0.f;
                }
                f32x16 acc;
                {
                    const float bch = biasl[ch];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = bch;
                }
#pragma unroll
                for (int s_ = 0; s_ < P / 16; ++s_) {
                    u32x4 wh, wl;
                    wop(48 + ch, s_, wh, wl);
                    acc = mfma_h(xs[0][s_], wh, acc);
                    acc = mfma_h(xs[1][s_], wh, acc);
                    acc = mfma_h(xs[0][s_], wl, acc);
                }
                __builtin_amdgcn_wave_barrier();        // esc[] of this block was written by this wave (LDS is in order per wave)
                asm volatile("" ::: "memory");
                const int* escl = reinterpret_cast<const int*>(lds + L.esc) + blk * 32 + 4 * hi1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int4 e4 = *reinterpret_cast<const int4*>(escl + 8 * g);
                    const int ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[4 * g + e] = __builtin_ldexpf(dgt[4 * g + e] * gate_from_scaled(acc[4 * g + e] * inv16), ev[e]);
                }
                u32x4 h0, l0, h1, l1;
                split8_rn(acc, 0, h0, l0);
                split8_rn(acc, 8, h1, l1);
                const bool lo_lane = r1 >= 16;
                u32x4 s0, s1;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s0[w] = lo_lane ? l0[w] : h0[w]; s1[w] = lo_lane ? l1[w] : h1[w]; }
                const unsigned base = L.dt + (unsigned)(blk * 4 + hi1) * 512u + (unsigned)r1 * 16u;
                *reinterpret_cast<u32x4*>(lds + base) = s0;
                *reinterpret_cast<u32x4*>(lds + base + 1024u) = s1;
            }
        }
        {   // E = min over the positions of the row
            int em = e_q;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int o_ = __shfl_xor(em, off);
                em = o_ < em ? o_ : em;
            }
            if (lane == 0) redl[wave] = active ? em : 127;
        }
        __syncthreads();
        int E = 127;
        for (int w = 0; w < nqb; ++w) { const int t = redl[w]; E = t < E ? t : E; }
        if (E == 127) E = 0;                            // no gradient anywhere in the row: every product below is zero
        const int es_q = e_q == 127 ? 0 : e_q;
        unsigned fmask;
        {
            const int f = lane < nqb ? reinterpret_cast<const int*>(lds + L.flag)[lane] : 0;
            fmask = (unsigned)__ballot(f != 0);
        }
        if (active) {
            // ================= sweep 0: lse of the block's queries (unless the forward kept it) =================
            float m_run = -1e30f, l_run = 0.f, cA;
            if (lse_in) {
                const float2 ml = valid ? *reinterpret_cast<const float2*>(lse_in + (((long)bu * H + h) * N + v) * 2) : make_float2(0.f, 0.f);
                m_run = ml.x;
                cA = ml.y - B2_PSHIFT;
            } else {
            for (int t = 0; t < nqb; ++t) {
                const KOp k = load_k(lds, krow_lane + 512u * t, kl_rel);
                f32x16 s = qk_tile(k, qh4, ql4, zero);
                if ((fmask >> t) & 1) mask_tile_at(lds, L.kadd, t, hi, 0.f, s);
                const float m_new = max2f(m_run, xhalf_max(max16_mfma(s)));
                l_run *= __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    t0 += __builtin_amdgcn_exp2f(s[j] - m_new);
                    t1 += __builtin_amdgcn_exp2f(s[j + 1] - m_new);
                }
                l_run += t0 + t1;
            }
            cA = __builtin_amdgcn_logf(xhalf_add(l_run)) - B2_PSHIFT;
            }
            // ================= pass A: dq of the block's queries =================
            const float mrefA = m_run + cA;
            f32x16 nl, nd, o;
#pragma unroll
            for (int e = 0; e < 16; ++e) { nl[e] = -mrefA; nd[e] = -delta_s; o[e] = 0.f; }
            for (int t = 0; t < nqb; ++t) {
                v2_prio(nqb - t, nqb);                  // the waves of a SIMD should reach the barrier together (see v2_prio)
                const KOp k = load_k(lds, krow_lane + 512u * t, kl_rel);
                const KOp vv = load_k(lds, vrow_lane + 512u * t, kl_rel);
                f32x16 s = qk_tile(k, qh4, ql4, nl);
                const f32x16 dp = qk_tile(vv, dh4, dl4, nd);
                const u32x4 va0 = *reinterpret_cast<const u32x4*>(lds + L.kt + tlane + 2048u * t);
                const u32x4 va1 = *reinterpret_cast<const u32x4*>(lds + L.kt + tlane + 2048u * t + 1024u);
                const bool fl = (fmask >> t) & 1;
                if (fl) {                               // replaced logits: (fill - m) - c, exact where fill = m
                    asm volatile("" ::: "memory");      // (a real branch: flagged tiles are rare, the selects are not free)
                    const float* kadd = reinterpret_cast<const float*>(lds + L.kadd) + 32 * t + 4 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 ka = *reinterpret_cast<const float4*>(kadd + 8 * g);
                        s[4 * g + 0] = ka.x == 0.f ? s[4 * g + 0] : (ka.x - m_run) - cA;
                        s[4 * g + 1] = ka.y == 0.f ? s[4 * g + 1] : (ka.y - m_run) - cA;
                        s[4 * g + 2] = ka.z == 0.f ? s[4 * g + 2] : (ka.z - m_run) - cA;
                        s[4 * g + 3] = ka.w == 0.f ? s[4 * g + 3] : (ka.w - m_run) - cA;
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) s[j] = __builtin_amdgcn_exp2f(s[j]) * dp[j];
                if (fl) {                               // no gradient through a replaced logit (masked or padded key)
                    asm volatile("" ::: "memory");
                    const float* kadd = reinterpret_cast<const float*>(lds + L.kadd) + 32 * t + 4 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 ka = *reinterpret_cast<const float4*>(kadd + 8 * g);
                        s[4 * g + 0] = ka.x == 0.f ? s[4 * g + 0] : 0.f;
                        s[4 * g + 1] = ka.y == 0.f ? s[4 * g + 1] : 0.f;
                        s[4 * g + 2] = ka.z == 0.f ? s[4 * g + 2] : 0.f;
                        s[4 * g + 3] = ka.w == 0.f ? s[4 * g + 3] : 0.f;
                    }
                }
                u32x4 h0, l0, h1, l1;
                split8_rn(s, 0, h0, l0);
                split8_rn(s, 8, h1, l1);
                o = mfma_h(va0, h0, o);
                o = mfma_h(va1, h1, o);
                o = mfma_h(va0, l0, o);
                o = mfma_h(va1, l1, o);
            }
            __builtin_amdgcn_s_setprio(0);
            if (valid) {
                const float f = __builtin_ldexpf(0.25f, -es_q - (int)B2_PSHIFT);
                float* dst = dqkvg + row_pos(v) * (4 * HC) + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(f * (o[0] + o[8]), f * (o[1] + o[9]), f * (o[2] + o[10]), f * (o[3] + o[11]));
                *reinterpret_cast<float4*>(dst + 8) = make_float4(f * (o[4] + o[12]), f * (o[5] + o[13]), f * (o[6] + o[14]), f * (o[7] + o[15]));
            }
            if (hi == 0) {
                const float cB = cA - (e_q == 127 ? 0.f : (float)(E - es_q));     // (no gradient at q: do' = 0 whatever the factor)
                reinterpret_cast<float*>(lds + L.lse)[v] = -(m_run + cB);         // negated: they start the accumulators of pass B
                reinterpret_cast<float*>(lds + L.mq)[v] = m_run;
                reinterpret_cast<float*>(lds + L.cq)[v] = cB;
                reinterpret_cast<float*>(lds + L.delta)[v] = -delta_s;
            }
        }
        __syncthreads();                                // K rows, V rows are free
        if (active) {
            const unsigned po = L.krow + (unsigned)hi * L.plane + (unsigned)(blk * 32 + r) * 16u;
            *reinterpret_cast<u32x4*>(lds + po) = qh4;
            *reinterpret_cast<u32x4*>(lds + po + kl_rel) = ql4;
            const unsigned pv = L.vrow + (unsigned)hi * L.plane + (unsigned)(blk * 32 + r) * 16u;
            *reinterpret_cast<u32x4*>(lds + pv) = dh4;
            *reinterpret_cast<u32x4*>(lds + pv + kl_rel) = dl4;
        }
        __syncthreads();
        if (active) {
            // ================= pass B: dk, dv of the block's keys =================
            const float ka = reinterpret_cast<const float*>(lds + L.kadd)[v];
            const bool kover = ka != 0.f;
            f32x16 ok, ov;
#pragma unroll
            for (int e = 0; e < 16; ++e) { ok[e] = 0.f; ov[e] = 0.f; }
            for (int t = 0; t < nqb; ++t) {
                v2_prio(nqb - t, nqb);
                const KOp qa = load_k(lds, krow_lane + 512u * t, kl_rel);
                const KOp da = load_k(lds, vrow_lane + 512u * t, kl_rel);
                f32x16 s, dp;                           // accumulators start at -lse_q, -delta_q of the register's query
                const float* lp = reinterpret_cast<const float*>(lds + L.lse) + 32 * t + 4 * hi;
                {
                    const float* dp_ = reinterpret_cast<const float*>(lds + L.delta) + 32 * t + 4 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 a = *reinterpret_cast<const float4*>(lp + 8 * g), d = *reinterpret_cast<const float4*>(dp_ + 8 * g);
                        s[4 * g] = a.x; s[4 * g + 1] = a.y; s[4 * g + 2] = a.z; s[4 * g + 3] = a.w;
                        dp[4 * g] = d.x; dp[4 * g + 1] = d.y; dp[4 * g + 2] = d.z; dp[4 * g + 3] = d.w;
                    }
                }
                s = qk_tile(qa, kh4, kl4, s);
                dp = qk_tile(da, vh4, vl4, dp);
                if (kover) {                            // the lane's key is masked / padded: its logit is a constant
                    asm volatile("" ::: "memory");
                    const float* mp = reinterpret_cast<const float*>(lds + L.mq) + 32 * t + 4 * hi;
                    const float* cp = reinterpret_cast<const float*>(lds + L.cq) + 32 * t + 4 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 a = *reinterpret_cast<const float4*>(mp + 8 * g), c4 = *reinterpret_cast<const float4*>(cp + 8 * g);
                        s[4 * g] = (ka - a.x) - c4.x; s[4 * g + 1] = (ka - a.y) - c4.y;
                        s[4 * g + 2] = (ka - a.z) - c4.z; s[4 * g + 3] = (ka - a.w) - c4.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s[j] = __builtin_amdgcn_exp2f(s[j]);
                    dp[j] = s[j] * dp[j];
                }
                {
                    const u32x4 va0 = *reinterpret_cast<const u32x4*>(lds + L.dt + tlane + 2048u * t);
                    const u32x4 va1 = *reinterpret_cast<const u32x4*>(lds + L.dt + tlane + 2048u * t + 1024u);
                    u32x4 h0, l0, h1, l1;
                    split8_rn(s, 0, h0, l0);
                    split8_rn(s, 8, h1, l1);
                    ov = mfma_h(va0, h0, ov);
                    ov = mfma_h(va1, h1, ov);
                    ov = mfma_h(va0, l0, ov);
                    ov = mfma_h(va1, l1, ov);
                }
                {
                    const u32x4 va0 = *reinterpret_cast<const u32x4*>(lds + L.qt + tlane + 2048u * t);
                    const u32x4 va1 = *reinterpret_cast<const u32x4*>(lds + L.qt + tlane + 2048u * t + 1024u);
                    u32x4 h0, l0, h1, l1;
                    split8_rn(dp, 0, h0, l0);
                    split8_rn(dp, 8, h1, l1);
                    ok = mfma_h(va0, h0, ok);
                    ok = mfma_h(va1, h1, ok);
                    ok = mfma_h(va0, l0, ok);
                    ok = mfma_h(va1, l1, ok);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (valid) {
                const float fv = __builtin_ldexpf(1.0f, -E - (int)B2_PSHIFT);
                const float fk = kover ? 0.f : fv * 0.6931471805599453f;      // Q rows carry log2(e); no gradient through a replaced logit
                float* dst = dqkvg + row_pos(v) * (4 * HC) + HC + h * C + 4 * hi;
                *reinterpret_cast<float4*>(dst) = make_float4(fk * (ok[0] + ok[8]), fk * (ok[1] + ok[9]), fk * (ok[2] + ok[10]), fk * (ok[3] + ok[11]));
                *reinterpret_cast<float4*>(dst + 8) = make_float4(fk * (ok[4] + ok[12]), fk * (ok[5] + ok[13]), fk * (ok[6] + ok[14]), fk * (ok[7] + ok[15]));
                *reinterpret_cast<float4*>(dst + HC) = make_float4(fv * (ov[0] + ov[8]), fv * (ov[1] + ov[9]), fv * (ov[2] + ov[10]), fv * (ov[3] + ov[11]));
                *reinterpret_cast<float4*>(dst + HC + 8) = make_float4(fv * (ov[4] + ov[12]), fv * (ov[5] + ov[13]), fv * (ov[6] + ov[14]), fv * (ov[7] + ov[15]));
            }
        }
    }
}

size_t v2l_lds_bytes(int N, int P, bool* share_out = nullptr) {
    const int NP = prd_round_up(N, 32), nqb = NP / 32, rem = nqb % 12, G = rem ? 12 / rem : 0;
    const size_t base = (size_t)64 * P * 4 + (size_t)NP * (2 * 32 + 64 + 4) + 128 + 64 + (size_t)V2L_TAIL_MAX * 128;
    const size_t extra = G >= 2 ? (size_t)rem * 2048 + 12 * 2560 : 0;
    const bool share = G >= 2 && base + extra <= 160 * 1024;       // else the last round runs unshared (idle waves, same result)
    if (share_out) *share_out = share;
    return base + (share ? extra : 0);
}

size_t v2_lds_bytes(int N, int P) {
    const int NP = prd_round_up(N, 32), nqb = NP / 32, m4 = nqb & 3;
    const size_t base = (size_t)64 * P * 4 + (size_t)NP * (4 * 32 + 64 + 64 + 4) + 128;
    return base + (size_t)(m4 ? m4 * (5 - m4) : 0) * 2560;      // partials: (owner + helpers) of every shared block
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
extern "C" int prd_tri_attn_v2_supported(int N, int P, int tune) {
    if (N <= 0 || (P != 32 && P != 64)) return 0;
    if (N <= V2_MAXN) return v2_lds_bytes(N, P) <= 160 * 1024 ? 1 : 0;
    if (PRD_TGET_TA2_NO_LONG(tune)) return 0;           // A/B switch: long rows stay on the first generation
    return (prd_round_up(N, 32) <= 1024 && v2l_lds_bytes(N, P) <= 160 * 1024) ? 1 : 0;
}

// which kernel prd_tri_attn_core_v2 launches for rows of N positions: 0 = none (unsupported), 1 = tri_attn_core_v2_kernel,
// 2 = tri_attn_core_v3_kernel (overlapped phases), 3 = tri_attn_core_v2l_kernel (long rows)
extern "C" int prd_tri_attn_v2_form(int N, int P, int tune) {
    if (!prd_tri_attn_v2_supported(N, P, tune)) return 0;
    if (N > V2_MAXN) return 3;
    return (!PRD_TGET_TA2_NO_V3(tune) && v3_lds_bytes(N, P) <= 160 * 1024) ? 2 : 1;
}

// lse (may be null; short rows only): [b * N rows][H][N][2] = (m, log2 l) of every query: reference and log2 of the sum of
// 2^(logit - m) in the log2 domain, kept by the training forward for prd_tri_attn_bwd_core_v2
extern "C" int prd_tri_attn_core_v2_lse(float* og, float* lse, const float* pair, const float* mask, const float* wq, const float* wk,
                                        const float* wv, const float* wg, const float* bg, int ending,
                                        int b, int N, int P, int H, int c, int tune, hipStream_t stream) {
    if (!og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0 || tune < 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (!prd_tri_attn_v2_supported(N, P, tune)) return PRD_ERR_UNSUPPORTED;
    if (lse && N > V2_MAXN) return PRD_ERR_UNSUPPORTED;
    if ((long)b * N * N > 0x7fffffffL / 2) return PRD_ERR_UNSUPPORTED;      // 32-bit position arithmetic in the kernel
    const int NP = prd_round_up(N, 32);
    const bool long_rows = N > V2_MAXN;
    bool share = false;
    const size_t lds = long_rows ? v2l_lds_bytes(N, P, &share) : v2_lds_bytes(N, P);
    const long rows_total = (long)b * N;
    const long cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head < 1) per_head = 1;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    // a multiple of 8 rows in flight per head (never more rounds: per_head <= cap = 64 either way): the kernels then keep the four
    // head-workgroups of a row on ONE XCD, whose L2 serves three of the four reads of the row.  (N = 769: 60 -> 64 rows in flight.)
    if (per_head >= 8 && !((tune >> 20) & 1)) per_head = (per_head + 7) / 8 * 8;      // (PRD_TUNE_TA2_NO_XCD8: A/B switch)
    if (per_head > cap) per_head = cap;
    const int grid = (int)(per_head * H);
    constexpr int NWV = 12;                             // nqb <= 12 query blocks, one wave each
    const int flags_env = PRD_TGET_TA2_FLAGS(tune);     // A/B switch: kernel flags given by the caller (-1: per-kernel default)
    const int use_v3 = !PRD_TGET_TA2_NO_V3(tune);       // A/B switch: 0 = the barrier-per-phase form
    const bool v3 = !long_rows && use_v3 && v3_lds_bytes(N, P) <= 160 * 1024;
    // bit 0 = key-loop priorities by remaining work: needed where the waves of a SIMD must end together (v2, v2l); with
    // overlapped phases (v3) an early finisher starts the next row's projection instead: 67.5 -> 65.9 us without them
    const int flags0 = flags_env >= 0 ? flags_env : (v3 ? 0 : 1);
    const int nqb_ = NP / 32, rem_ = nqb_ % 12;
    // long rows: a ragged last key tile of 1 .. V2L_TAIL_MAX keys is swept as rank-1 updates (flag 64; needs the [K|V] phase 1;
    // A/B: PRD_TA2_FLAGS with bit 1 set keeps it a regular tile)
    const int ntail_ = N - 32 * (nqb_ - 1);
    const bool tail1 = long_rows && nqb_ >= 2 && ntail_ >= 1 && ntail_ <= V2L_TAIL_MAX && !PRD_TGET_TA2_NO_GV(tune) && !(flags0 & 2);
    const int flags = flags0 | ((long_rows && rem_ && 12 / rem_ >= 2 && !share) ? 16 : 0) | (PRD_TGET_TA2_NO_TAIL_SPLIT(tune) ? 32 : 0)
                      | (tail1 ? 64 : 0) | ((long_rows && (flags0 & 4)) ? 128 : 0);    // (PRD_TA2_FLAGS bit 2: the shared last round in its round-5 form)
    if (long_rows) {
#define PRD_V2L_LAUNCH(PP, PF, GVF)                                                                                               \
        do {                                                                                                                      \
            PRD2_SET_LDS((tri_attn_core_v2l_kernel<PP, NWV, PF, GVF>));                                                           \
            hipLaunchKernelGGL((tri_attn_core_v2l_kernel<PP, NWV, PF, GVF>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, wq, wk, \
                               wv, wg, bg, b, N, NP, H, ending, flags);                                                           \
        } while (0)
        const bool pf = (flags & 8) != 0;               // next-row prefetch of the wave's first block (costs 32 registers)
        const bool gvl = !PRD_TGET_TA2_NO_GV(tune);     // phase 1 with [K|V] as one row GEMM + transposed V store
#ifdef PRD_AB       // (libprd_hip_ab.so) the measured-and-superseded forms: next-row prefetch (spills), the round-3 phase 1
        if (P == 64) {
            if (gvl) { if (pf) PRD_V2L_LAUNCH(64, true, true); else PRD_V2L_LAUNCH(64, false, true); }
            else { if (pf) PRD_V2L_LAUNCH(64, true, false); else PRD_V2L_LAUNCH(64, false, false); }
        } else {
            if (gvl) { if (pf) PRD_V2L_LAUNCH(32, true, true); else PRD_V2L_LAUNCH(32, false, true); }
            else { if (pf) PRD_V2L_LAUNCH(32, true, false); else PRD_V2L_LAUNCH(32, false, false); }
        }
#else               // the shipped library carries the default form only; the switches that select another one are ignored
        (void)pf; (void)gvl;
        if (P == 64) PRD_V2L_LAUNCH(64, false, true); else PRD_V2L_LAUNCH(32, false, true);
#endif
#undef PRD_V2L_LAUNCH
        return (int)hipGetLastError();
    }
    if (v3) {
        const size_t lds3 = v3_lds_bytes(N, P);
#define PRD_V3_LAUNCH(PP, KLF, GVF)                                                                                                 \
        do {                                                                                                                      \
            PRD2_SET_LDS((tri_attn_core_v3_kernel<PP, NWV, KLF, GVF>));                                                           \
            hipLaunchKernelGGL((tri_attn_core_v3_kernel<PP, NWV, KLF, GVF>), dim3(grid), dim3(NWV * 64), lds3, stream, og, pair, mask, wq, wk, wv, wg, \
                               bg, b, N, NP, H, ending, flags & ~6, lse);                                                         \
        } while (0)
        const int kl = flags_env >= 0 ? (flags >> 1) & 3 : PRD_V3_DEFAULT_KL;     // key-loop form (bits 1-2 of PRD_TA2_FLAGS; v3 has no stagger)
        // phase 1 with [G|V] as one row GEMM + transposed V store (default); PRD_TUNE_TA2_NO_GV: the G GEMM + swapped V GEMM of round 3.
        // The A/B key-loop forms 1-3 exist with the round-3 phase 1 only.
        const bool gvf = !PRD_TGET_TA2_NO_GV(tune) && kl == 0;
#ifdef PRD_AB       // (libprd_hip_ab.so) the key-loop forms 1-3 and the round-3 phase 1: measured in round 5, none faster
        if (P == 64) {
            if (gvf) PRD_V3_LAUNCH(64, 0, true);
            else if (kl == 0) PRD_V3_LAUNCH(64, 0, false); else if (kl == 1) PRD_V3_LAUNCH(64, 1, false); else if (kl == 2) PRD_V3_LAUNCH(64, 2, false); else PRD_V3_LAUNCH(64, 3, false);
        } else {
            if (gvf) PRD_V3_LAUNCH(32, 0, true);
            else if (kl == 0) PRD_V3_LAUNCH(32, 0, false); else if (kl == 1) PRD_V3_LAUNCH(32, 1, false); else if (kl == 2) PRD_V3_LAUNCH(32, 2, false); else PRD_V3_LAUNCH(32, 3, false);
        }
#else               // the shipped library carries the default form only (the scheduling switches -- bits 0, 3, 4 -- still reach it)
        (void)gvf;
        if (P == 64) PRD_V3_LAUNCH(64, 0, true); else PRD_V3_LAUNCH(32, 0, true);
#endif
#undef PRD_V3_LAUNCH
        return (int)hipGetLastError();
    }
    if (P == 64) {
        PRD2_SET_LDS((tri_attn_core_v2_kernel<64, NWV>));
        hipLaunchKernelGGL((tri_attn_core_v2_kernel<64, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N,
                           NP, H, ending, flags, lse);
    } else {
        PRD2_SET_LDS((tri_attn_core_v2_kernel<32, NWV>));
        hipLaunchKernelGGL((tri_attn_core_v2_kernel<32, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N,
                           NP, H, ending, flags, lse);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn_core_v2(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                                    const float* wv, const float* wg, const float* bg, int ending,
                                    int b, int N, int P, int H, int c, int tune, hipStream_t stream) {
    return prd_tri_attn_core_v2_lse(og, nullptr, pair, mask, wq, wk, wv, wg, bg, ending, b, N, P, H, c, tune, stream);
}

// ---- the triangle-attention pair of a folding block as ONE persistent launch (tri_attn_pair_kernel) ----
static int prd_cu_count() {
    static int cus = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = n;
    });
    return cus;
}

// 1 when prd_tri_attn_pair serves (N, P): split-16 arithmetic, rows on the overlapped-phase short-row core (the form the default
// dispatch takes for N <= 320-odd), default kernel switches
extern "C" int prd_tri_attn_pair_supported(int N, int P, int arith) {
    PRD_SPLIT_ARITH(arith);
    if (N <= 0 || (P != 32 && P != 64) || arith != PRD_ARITH_SPLIT16) return 0;
    if (N > V2_MAXN || PRD_TGET_TA2_NO_V3(tune) || PRD_TGET_TA2_NO_GV(tune) || PRD_TGET_TA2_FLAGS(tune) >= 0) return 0;
    return v3_lds_bytes(N, P) <= 160 * 1024 ? 1 : 0;
}

extern "C" int prd_tri_attn_pair(float* og, float* pair, const float* mask, const float* const* w_start, const float* const* w_end,
                                 int b, int N, int P, int H, int c, unsigned* bar, int arith, hipStream_t stream) {
    if (!og || !pair || !mask || !w_start || !w_end || !bar || b <= 0 || N <= 0 || arith < 0) return PRD_ERR_ARG;
    const int tune = arith >> 8;                        // (the PRD_TUNE_* word above the arithmetic)
    for (int k = 0; k < 7; ++k) if (!w_start[k]) return PRD_ERR_ARG;
    for (int k = 0; k < 5; ++k) if (!w_end[k]) return PRD_ERR_ARG;
    if (c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (!prd_tri_attn_pair_supported(N, P, arith)) return PRD_ERR_UNSUPPORTED;
    if ((long)b * N * N > 0x7fffffffL / 2) return PRD_ERR_UNSUPPORTED;
    const int NP = prd_round_up(N, 32);
    const long rows_total = (long)b * N, cap = 256 / H;                 // the grid of prd_tri_attn_core_v2 (heads of a row on one XCD)
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head < 1) per_head = 1;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    if (per_head >= 8) per_head = (per_head + 7) / 8 * 8;
    if (per_head > cap) per_head = cap;
    const int grid = (int)(per_head * H);
    // every workgroup must be resident for the in-kernel barriers: one workgroup of 12 waves + ~150 KB of LDS per CU
    const int cus = prd_cu_count();
    if (cus <= 0 || grid > cus) return PRD_ERR_UNSUPPORTED;
    const size_t lds3 = v3_lds_bytes(N, P);
    const size_t ldso = (size_t)P * 64 * 4 + (size_t)P * 4;
    const size_t lds = lds3 > ldso ? lds3 : ldso;
    hipError_t e = hipMemsetAsync(bar, 0, 32 * sizeof(unsigned), stream);     // counters, timeout flag, memberships: zero before EVERY launch
    const int plain = PRD_TGET_TA2_NO_TAIL_SPLIT(tune) ? 1 : 0;               // (A/B: PRD_TA2_TAIL=0 selects the plain counter barrier here)
    if (e != hipSuccess) return (int)e;
    constexpr int NWV = 12;
    if (P == 64) {
        PRD2_SET_LDS((tri_attn_pair_kernel<64, NWV>));
        hipLaunchKernelGGL((tri_attn_pair_kernel<64, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, w_start[0], w_start[1], w_start[2],
                           w_start[3], w_start[4], w_start[5], w_start[6], w_end[0], w_end[1], w_end[2], w_end[3], w_end[4], b, N, NP, H, bar, plain);
    } else {
        PRD2_SET_LDS((tri_attn_pair_kernel<32, NWV>));
        hipLaunchKernelGGL((tri_attn_pair_kernel<32, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, w_start[0], w_start[1], w_start[2],
                           w_start[3], w_start[4], w_start[5], w_start[6], w_end[0], w_end[1], w_end[2], w_end[3], w_end[4], b, N, NP, H, bar, plain);
    }
    return (int)hipGetLastError();
}

// ---- backward core, split-16 arithmetic (tri_attn_bwd_core_v2_kernel) ----
extern "C" int prd_tri_attn_bwd_core_v2_supported(int N, int P) {
    if (N <= 0 || (P != 32 && P != 64)) return 0;
    const int NP = prd_round_up(N, 32);
    return (NP / 32 <= 12 && b2_layout(P, NP).total <= 160u * 1024u) ? 1 : 0;
}

extern "C" int prd_tri_attn_bwd_core_v2(float* dqkvg, const float* dog, const float* og, const float* pair, const float* mask,
                                        const float* wq, const float* wk, const float* wv, const float* wg, const float* bg,
                                        const float* lse, float* x_out, int ending, int b, int N, int P, int H, int c, hipStream_t stream) {
    if (!dqkvg || !dog || !og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (!prd_tri_attn_bwd_core_v2_supported(N, P)) return PRD_ERR_UNSUPPORTED;
    if ((long)b * N * N > 0x7fffffffL / 4) return PRD_ERR_UNSUPPORTED;      // 32-bit position arithmetic in the kernel
    const int NP = prd_round_up(N, 32);
    const size_t lds = b2_layout(P, NP).total;
    const long rows_total = (long)b * N, cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head >= 8) per_head = per_head / 8 * 8;     // the heads of a row on one XCD
    const int grid = (int)(per_head * H);
    constexpr int NWV = 12;
    if (P == 64) {
        PRD2_SET_LDS((tri_attn_bwd_core_v2_kernel<64, NWV>));
        hipLaunchKernelGGL((tri_attn_bwd_core_v2_kernel<64, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, dqkvg, dog, og, pair, mask, wq, wk, wv,
                           wg, bg, b, N, NP, H, ending, lse, x_out);
    } else {
        PRD2_SET_LDS((tri_attn_bwd_core_v2_kernel<32, NWV>));
        hipLaunchKernelGGL((tri_attn_bwd_core_v2_kernel<32, NWV>), dim3(grid), dim3(NWV * 64), lds, stream, dqkvg, dog, og, pair, mask, wq, wk, wv,
                           wg, bg, b, N, NP, H, ending, lse, x_out);
    }
    return (int)hipGetLastError();
}
