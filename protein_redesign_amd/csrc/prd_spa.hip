// SPAttention core with WIDE heads (head width c = single_dim: 64 ... 512) in ONE launch, split-16 arithmetic:
//     o[b, q, h, :] = gate[b, q, h, :] * sum_k softmax_k(q[b,q,h,:] . k[b,k,h,:] + bias[b,h,q,k]) v[b,k,h,:]
// Replaces, for the head of the trunk (models/AF2_modules.py:421-473 -> 251-293, 613-628: logits, softmax, P V of the gated
// attention; the pair bias of :454-459 comes in as `bias`), three launches of round 4 -- the per-head logits GEMM, the row softmax
// riding in the P V GEMM's operand staging, the P V GEMM (12.8 + 19.6 us at N = 320, c = 512) -- and the [b, H, N, N] logits round
// trip.  The output projection stays a GEMM of its own (the K-slab path).
//
// What bounds this operator is the OPERAND STREAM PER CU (~50 GB/s whatever is in flight): K and V of a head are 1.3 MB
// at N = 320, c = 512, and the first form of this kernel -- a workgroup per (head, 32 queries, quarter of the channels), each reading
// ALL keys -- ran 32 us (0.9 MB per workgroup), no faster than the GEMMs it replaced.  So the keys are SPLIT across workgroups
// (the flash-decoding form): a workgroup of 8 waves owns (complex, head, 32 queries, key part of 1-4 key tiles of 32):
//   phase 0  the Q block (32 x c, pre-scaled by the projection's column scale and by log2 e here) -> fp16 hi | lo planes in LDS.
//   phase A  logits, transposed: S^T[32 keys x 32 queries] = K Q^T on v_mfma_f32_32x32x16_f16 (3 split products per k-step:
//            kh qh + kh ql + kl qh).  Unit (key tile, slice of the head width) -> wave: a wave streams ITS K rows through a
//            wave-private LDS tile (coalesced 256-byte row segments in, row-per-lane operand fragments out: no workgroup barrier
//            in the loop); the slices of a tile are summed through LDS.  A lane of the tile's lead wave ends up with 16 logits of
//            ONE query (the swapped form of csrc/prd_tri2.hip): pair bias added, padded keys excluded, optional key mask as the
//            reference's fill value; tile maxima / sums meet in LDS.
//   phase B  P = 2^(s - max_part + 8) split into fp16 hi | lo stays in the register layout of S^T and goes to LDS exactly so: it IS the
//            B operand of O^T[32 channels x 32 queries] = V^T P.  Channel tiles are dealt to the waves; a wave streams ITS V rows
//            through a private LDS tile written TRANSPOSED ([khalf][channel][8 keys], two keys packed per 32-bit store) in the key
//            order of the S^T register layout, three products per 16 keys (no dead quadrant: 32 channels fill the M side).
//   out      one key part: normalised, gated, stored.  Several: the un-normalised partial O and (max, sum) of the part go to a
//            workspace and spa_attn_merge_kernel combines the parts (w_p = 2^(max_p - max), o = gate * sum w_p O_p / sum w_p l_p).
#include "prd_common.h"
#include "../../include/prd_hip.h"
#include <mutex>

namespace {

constexpr int SPA_NW = 8, SPA_NT = SPA_NW * 64;
constexpr int SPA_KCH = 64;                              // channels of a K tile per staging round
constexpr int SPA_KPITCH = SPA_KCH * 2 + 16;             // bytes per key row of a staged plane (+16: conflict-free fragment reads)
constexpr float SPA_LOG2E = 1.4426950408889634f;
constexpr float SPA_PSHIFT = 8.0f;                       // probabilities are 2^(s - max + 8): a normal lo part down to 2^-11 of the maximum

PRD_DEV f32x16 mfma_h16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
PRD_DEV float xhalf_maxf(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
PRD_DEV float xhalf_addf(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

struct SpaLds { unsigned q_hi, q_lo, qpitch, kst, stat_m, stat_l, total; };
// Phase A: Q planes [32][c fp16 + 16 B] x 2 | K staging [8 waves][2 planes][32][SPA_KPITCH] (re-used for the slice partials) |
// tile maxima [4][32] | tile sums.  Phase B re-uses the Q region: P planes [tile][2 ksteps][2 planes][2 khalf][32 q][16 B] |
// V staging [8 waves][2 buffers][2 planes][2 khalf][32 ch][16 B].
PRD_DEV SpaLds spa_layout(int c) {
    SpaLds L;
    L.qpitch = (unsigned)c * 2u + 16u;
    L.q_hi = 0;
    L.q_lo = 32u * L.qpitch;
    L.kst = 64u * L.qpitch;
    unsigned a_end = L.kst + SPA_NW * 2u * 32u * SPA_KPITCH;
    unsigned b_end = 4u * 4096u + SPA_NW * 2u * 2048u;
    unsigned off = a_end > b_end ? a_end : b_end;
    L.stat_m = off; off += 4u * 128u;
    L.stat_l = off; off += 4u * 128u;
    L.total = off;
    return L;
}
constexpr int SPA_MAXLT = 4;                             // key tiles of a workgroup's key part

__global__ __launch_bounds__(SPA_NT) void spa_attn_part_kernel(
    float* __restrict__ o, float* __restrict__ ws, const float* __restrict__ qkvg, int ldq, const float* __restrict__ bias,
    const float* __restrict__ mask, int b, int N, int H, int c, int KP, int nqb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int ntile = (N + 31) / 32;
    const int HC = H * c;
    const SpaLds L = spa_layout(c);
    // block -> (complex, head, query block, key part): the key parts of a (head, query block) next to each other
    int id = blockIdx.x;
    const int kp = id % KP; id /= KP;
    const int qb = id % nqb; id /= nqb;
    const int h = id % H, bb = id / H;
    const int q0 = qb * 32;
    const int t0 = (ntile * kp) / KP, nlt = (ntile * (kp + 1)) / KP - t0;      // this part's key tiles [t0, t0 + nlt), 1 <= nlt <= 4
    const float* __restrict__ rowbase = qkvg + (size_t)bb * N * ldq;
    float* stat_m = reinterpret_cast<float*>(lds + L.stat_m);
    float* stat_l = reinterpret_cast<float*>(lds + L.stat_l);

    // ---------------- phase 0: Q block -> fp16 hi | lo planes (x log2 e: the softmax runs in the exp2 domain) ----------------
    {
        const int per_row = c / 4;                       // float4 pieces per row
        for (int p = tid; p < 32 * per_row; p += SPA_NT) {
            const int row = p / per_row, f = p - row * per_row;
            const int qrow = q0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qrow < N) v = *reinterpret_cast<const float4*>(rowbase + (size_t)qrow * ldq + h * c + 4 * f);
            unsigned h0, l0, h1, l1;
            split2h(v.x * SPA_LOG2E, v.y * SPA_LOG2E, h0, l0);
            split2h(v.z * SPA_LOG2E, v.w * SPA_LOG2E, h1, l1);
            *reinterpret_cast<u32x2*>(lds + L.q_hi + (unsigned)row * L.qpitch + 8u * f) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(lds + L.q_lo + (unsigned)row * L.qpitch + 8u * f) = u32x2{l0, l1};
        }
    }
    __syncthreads();

    // ---------------- phase A: unit (local key tile, slice of the head width) -> wave ----------------
    const int nch = c / SPA_KCH;                         // 64-channel chunks of the head width
    int csplit = SPA_NW / nlt;                           // slices per tile: 8, 4, 2, 2 for 1 .. 4 tiles, at most one chunk each
    if (csplit > nch) csplit = nch;
    while (nch % csplit) --csplit;
    const int nunits = nlt * csplit;
    const bool unit_ok = wave < nunits;
    const int tl = unit_ok ? wave / csplit : 0, sj = unit_ok ? wave % csplit : 0;
    const int T = t0 + tl;
    unsigned char* kst = lds + L.kst + (unsigned)wave * (2u * 32u * SPA_KPITCH);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // the pair bias of the lane's query x its 16 keys (row-per-lane loads: requested BEFORE the K loop, consumed after it)
    float bq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bq[e] = 0.f;
    if (unit_ok && sj == 0 && bias) {
        const int qq = q0 + r;
        const float* brow = bias + (((size_t)bb * H + h) * N + (qq < N ? qq : 0)) * N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = 32 * T + 8 * g + 4 * hi;
            if ((N & 3) == 0 && k0 + 3 < N) {              // rows of the bias are 16-byte aligned: one load for the lane's four keys
                const float4 t4 = *reinterpret_cast<const float4*>(brow + k0);
                bq[4 * g] = t4.x; bq[4 * g + 1] = t4.y; bq[4 * g + 2] = t4.z; bq[4 * g + 3] = t4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k0 + e < N) bq[4 * g + e] = brow[k0 + e];
            }
        }
    }
    if (unit_ok) {
        const int srow = lane >> 4, scol = lane & 15;    // staging: lane = (row srow + 4 j, 16-byte piece scol of the 256-byte segment)
        const int ch_lo = sj * (nch / csplit), ch_hi = ch_lo + nch / csplit;
        float4 kv[2][8];                                 // two chunks in flight
        auto load_chunk = [&](int ch, float4 (&dst)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 32 * T + srow + 4 * j;
                dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (key < N && ch < ch_hi) dst[j] = *reinterpret_cast<const float4*>(rowbase + (size_t)key * ldq + HC + h * c + ch * SPA_KCH + 4 * scol);
            }
        };
        load_chunk(ch_lo, kv[0]);
        load_chunk(ch_lo + 1, kv[1]);
        for (int ch0 = ch_lo; ch0 < ch_hi; ch0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int ch = ch0 + d;
                if (ch >= ch_hi) break;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned h0, l0, h1, l1;
                    split2h(kv[d][j].x, kv[d][j].y, h0, l0);
                    split2h(kv[d][j].z, kv[d][j].w, h1, l1);
                    const unsigned off = (unsigned)(srow + 4 * j) * SPA_KPITCH + 8u * scol;
                    *reinterpret_cast<u32x2*>(kst + off) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(kst + 32u * SPA_KPITCH + off) = u32x2{l0, l1};
                }
                load_chunk(ch + 2, kv[d]);
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < SPA_KCH / 16; ++s) {
                    const unsigned ko = (unsigned)r * SPA_KPITCH + (unsigned)(16 * s + 8 * hi) * 2u;
                    const u32x4 kh = *reinterpret_cast<const u32x4*>(kst + ko), kl = *reinterpret_cast<const u32x4*>(kst + 32u * SPA_KPITCH + ko);
                    const unsigned qo = (unsigned)r * L.qpitch + (unsigned)(ch * SPA_KCH + 16 * s + 8 * hi) * 2u;
                    const u32x4 qh = *reinterpret_cast<const u32x4*>(lds + L.q_hi + qo), ql = *reinterpret_cast<const u32x4*>(lds + L.q_lo + qo);
                    acc = mfma_h16(kh, qh, acc);
                    acc = mfma_h16(kh, ql, acc);
                    acc = mfma_h16(kl, qh, acc);
                }
                wave_lds_fence();
            }
        }
        if (csplit > 1 && sj > 0) {                      // slice partial -> the wave's own (now idle) staging tile
            float* pp = reinterpret_cast<float*>(kst);
#pragma unroll
            for (int e = 0; e < 16; ++e) pp[e * 64 + lane] = acc[e];
        }
    }
    __syncthreads();                                     // all fragment reads of Q and K done; slice partials visible
    const bool lead = unit_ok && sj == 0;                // the wave that finishes tile tl
    if (lead) {
        for (int j2 = 1; j2 < csplit; ++j2) {
            const float* pp = reinterpret_cast<const float*>(lds + L.kst + (unsigned)(wave + j2) * (2u * 32u * SPA_KPITCH));
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += pp[e * 64 + lane];
        }
        // pair bias (x log2 e), padded keys out, optional key mask as the reference's fill value (modules.py:216-219: -2^15)
        float tmax = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = 32 * T + 8 * g + 4 * hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = k0 + e;
                float v = acc[4 * g + e];
                if (key < N) v += SPA_LOG2E * bq[4 * g + e];
                if (mask && key < N && mask[bb * N + key] < 0.5f) v = -32768.0f * SPA_LOG2E;
                if (key >= N) v = -INFINITY;
                acc[4 * g + e] = v;
                tmax = fmaxf(tmax, v);
            }
        }
        tmax = xhalf_maxf(tmax);
        if (hi == 0) stat_m[tl * 32 + r] = tmax;
    }
    __syncthreads();
    // maximum of the lane's query over this part's keys (every tile has at least one real key: finite)
    float M = -INFINITY;
    for (int t = 0; t < nlt; ++t) M = fmaxf(M, stat_m[t * 32 + r]);
    // ---------------- P = 2^(s - M + shift), sums, fp16 hi | lo -> LDS in the register layout (the B operand of P V) ----------------
    if (lead) {
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[e] = __builtin_amdgcn_exp2f(acc[e] - M + SPA_PSHIFT);
            ls += acc[e];
        }
        ls = xhalf_addf(ls);
        if (hi == 0) stat_l[tl * 32 + r] = ls;
        unsigned char* pb = lds + (unsigned)tl * 4096u + (unsigned)hi * 512u + (unsigned)r * 16u;    // [tile][a][plane][khalf][q][16 B]
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            u32x4 ph, pl;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned x, y;
                split2h(acc[8 * a + 2 * w], acc[8 * a + 2 * w + 1], x, y);
                ph[w] = x; pl[w] = y;
            }
            *reinterpret_cast<u32x4*>(pb + (unsigned)a * 2048u) = ph;
            *reinterpret_cast<u32x4*>(pb + (unsigned)a * 2048u + 1024u) = pl;
        }
    }
    __syncthreads();
    float lsum = 0.f;
    for (int t = 0; t < nlt; ++t) lsum += stat_l[t * 32 + r];

    // ---------------- phase B: O^T[32 channels x 32 queries] per channel tile (tiles wave, wave + 8, ...) ----------------
    unsigned char* vst = lds + 4u * 4096u + (unsigned)wave * 4096u;      // 2 buffers x 2 KB
    const int vpair = lane >> 3, vc4 = lane & 7;         // V staging: lane = (key pair 2 vpair | 2 vpair + 1 of the 16-key step, channels 4 vc4 ..)
    // key of step position (pair member m): kk = 2 vpair + m -> khalf = (kk >> 2) & 1, i = 4 (kk >> 3) + (kk & 3)
    // (key pairs in the order 0, 2, 8, 10 | 4, 6, 12, 14: the 32 lanes of a store half then cover all four dwords w of the khalf they
    // share -- 8 slots x 4 dwords = 32 banks; in natural order a half saw two dwords and two khalf blocks: 2-way conflicts)
    const int kk0 = 2 * ((vpair & 1) | ((vpair & 2) << 1) | ((vpair & 4) >> 1));
    const unsigned vdst = (unsigned)((kk0 >> 2) & 1) * 512u + (unsigned)(4 * (kk0 >> 3) + (kk0 & 3)) * 2u;     // + plane 1024 + ch 16
    // 16-byte slot of channel ch inside a (plane, khalf) block: a bit permutation of ch (slot bits = ch2, ch3, ch4 ^ ch0, ch1, ch0) chosen so
    // that the eight lanes vc4 of a transposed 4-byte store hit eight different slots mod 8 (all 32 banks, two lanes each) AND the
    // sixteen lanes of a 16-byte fragment read hit sixteen different slots mod 16; with slot = ch the stores were 8-way conflicted
    auto vslot = [](int ch) { return ((ch >> 2) & 3) | ((((ch >> 4) ^ ch) & 1) << 2) | (((ch >> 1) & 1) << 3) | ((ch & 1) << 4); };
    unsigned vwr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vwr[j] = (unsigned)vslot(4 * vc4 + j) * 16u;
    const unsigned vrd = (unsigned)vslot(r) * 16u;
    const int nct = c / 32;
    const int step_lo = 2 * t0, step_end = 2 * (t0 + nlt);
    const int q = q0 + r;
    for (int ct = wave; ct < nct; ct += SPA_NW) {
        const int chan0 = h * c + 32 * ct;               // first channel of the tile inside a q | k | v | gate block
        f32x16 oacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
        float4 vav[2], vbv[2];
        auto load_v = [&](int step, float4& va, float4& vb) {                    // step = 2 T + a
            const int key = 16 * step + kk0;
            va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va;
            if (step < step_end && key < N) va = *reinterpret_cast<const float4*>(rowbase + (size_t)key * ldq + 2 * HC + chan0 + 4 * vc4);
            if (step < step_end && key + 1 < N) vb = *reinterpret_cast<const float4*>(rowbase + (size_t)(key + 1) * ldq + 2 * HC + chan0 + 4 * vc4);
        };
        load_v(step_lo, vav[0], vbv[0]);
        load_v(step_lo + 1, vav[1], vbv[1]);
        for (int step0 = step_lo; step0 < step_end; step0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int step = step0 + d;              // (step_end - step_lo is even: whole tiles)
                unsigned char* vbuf = vst + (unsigned)d * 2048u;
                {   // transposed store: channel 4 vc4 + j, keys (kk0, kk0 + 1) packed
                    const float4 va = vav[d], vb = vbv[d];
                    const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned hh, ll;
                        split2h(xa[j], xb[j], hh, ll);
                        *reinterpret_cast<unsigned*>(vbuf + vdst + vwr[j]) = hh;
                        *reinterpret_cast<unsigned*>(vbuf + 1024u + vdst + vwr[j]) = ll;
                    }
                }
                load_v(step + 2, vav[d], vbv[d]);
                wave_lds_fence();
                const u32x4 vh = *reinterpret_cast<const u32x4*>(vbuf + (unsigned)hi * 512u + vrd);
                const u32x4 vl = *reinterpret_cast<const u32x4*>(vbuf + 1024u + (unsigned)hi * 512u + vrd);
                const int sl = step - step_lo;
                const unsigned char* pb = lds + (unsigned)(sl >> 1) * 4096u + (unsigned)(sl & 1) * 2048u + (unsigned)hi * 512u + (unsigned)r * 16u;
                const u32x4 ph = *reinterpret_cast<const u32x4*>(pb), pl = *reinterpret_cast<const u32x4*>(pb + 1024u);
                oacc = mfma_h16(vh, ph, oacc);
                oacc = mfma_h16(vl, ph, oacc);
                oacc = mfma_h16(vh, pl, oacc);
            }
        }
        wave_lds_fence();
        if (q < N) {
            if (KP == 1) {                               // the only key part: normalise, gate, store
                const float il = 1.0f / lsum;
                const float* grow = rowbase + (size_t)q * ldq + 3 * HC + chan0 + 4 * hi;
                float* orow = o + ((size_t)bb * N + q) * HC + chan0 + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 gt = *reinterpret_cast<const float4*>(grow + 8 * g);
                    *reinterpret_cast<float4*>(orow + 8 * g) = make_float4(gt.x * oacc[4 * g] * il, gt.y * oacc[4 * g + 1] * il,
                                                                           gt.z * oacc[4 * g + 2] * il, gt.w * oacc[4 * g + 3] * il);
                }
            } else {                                     // partial of this key part: ws[part][b, q, h c]
                float* orow = ws + ((size_t)kp * b * N + (size_t)bb * N + q) * HC + chan0 + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(orow + 8 * g) = make_float4(oacc[4 * g], oacc[4 * g + 1], oacc[4 * g + 2], oacc[4 * g + 3]);
            }
        }
    }
    if (KP > 1 && wave == 0 && hi == 0 && q < N) {      // (max, sum) of this part for the lane's query
        float* st = ws + (size_t)KP * b * N * HC + (((size_t)kp * b * N + (size_t)bb * N + q) * H + h) * 2;
        st[0] = M;
        st[1] = lsum;
    }
}

// o = gate * sum_p w_p O_p / sum_p w_p l_p with w_p = 2^(max_p - max): one thread per (b, q, h, 4 channels)
__global__ __launch_bounds__(256) void spa_attn_merge_kernel(float* __restrict__ o, const float* __restrict__ ws, const float* __restrict__ qkvg,
                                                             int ldq, int b, int N, int H, int c, int KP) {
    const int HC = H * c;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)b * N * (HC / 4);
    if (idx >= total) return;
    const long row = idx / (HC / 4);
    const int col = (int)(idx - row * (HC / 4)) * 4, h = col / c;
    const float* st = ws + (size_t)KP * b * N * HC;
    float M = -INFINITY;
    for (int p = 0; p < KP; ++p) M = fmaxf(M, st[(((size_t)p * b * N + row) * H + h) * 2]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float l = 0.f;
    for (int p = 0; p < KP; ++p) {
        const float* s2 = st + (((size_t)p * b * N + row) * H + h) * 2;
        const float w = __builtin_amdgcn_exp2f(s2[0] - M);
        const float4 v = *reinterpret_cast<const float4*>(ws + ((size_t)p * b * N + row) * HC + col);
        l += w * s2[1];
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    const float il = 1.0f / l;
    const float4 gt = *reinterpret_cast<const float4*>(qkvg + (size_t)row * ldq + 3 * HC + col);
    *reinterpret_cast<float4*>(o + (size_t)row * HC + col) = make_float4(gt.x * acc.x * il, gt.y * acc.y * il, gt.z * acc.z * il, gt.w * acc.w * il);
}

size_t spa_lds_bytes(int c) {
    const unsigned qpitch = (unsigned)c * 2u + 16u;
    const size_t a_end = (size_t)64 * qpitch + (size_t)SPA_NW * 2 * 32 * SPA_KPITCH;
    const size_t b_end = (size_t)4 * 4096 + (size_t)SPA_NW * 4096;
    return (a_end > b_end ? a_end : b_end) + 1024;
}

// key parts: enough workgroups for ~200+ and at most SPA_MAXLT key tiles per part
int spa_key_parts(int b, int N, int H) {
    const int ntile = (N + 31) / 32, nqb = ntile;
    int kp = (ntile + SPA_MAXLT - 1) / SPA_MAXLT;
    while (kp < ntile && (long)b * H * nqb * kp < 200) ++kp;
    return kp;
}

}  // namespace

// 1 when prd_spa_attn_core serves heads of width c over rows of N positions in this arithmetic (split-16 only; the fp32-MFMA
// arithmetic keeps the three-launch form)
extern "C" int prd_spa_attn_core_supported(int N, int c, int arith) {
    if (arith < 0 || (arith & 0xff) != PRD_ARITH_SPLIT16) return 0;
    if (N <= 0 || c < 64 || c > 512 || (c % 64) != 0) return 0;
    return spa_lds_bytes(c) <= 160 * 1024 ? 1 : 0;
}

// bytes of workspace prd_spa_attn_core needs (0: one key part, no merge launch)
extern "C" size_t prd_spa_attn_core_workspace(int b, int N, int H, int c) {
    if (b <= 0 || N <= 0 || H <= 0 || c <= 0) return 0;
    const int kp = spa_key_parts(b, N, H);
    return kp > 1 ? ((size_t)kp * b * N * H * c + (size_t)kp * b * N * H * 2) * sizeof(float) : 0;
}

extern "C" int prd_spa_attn_core(float* o, const float* qkvg, int ldq, const float* bias, const float* mask,
                                 int b, int N, int H, int c, float* ws, size_t ws_bytes, int arith, hipStream_t stream) {
    if (!o || !qkvg || b <= 0 || N <= 0 || H <= 0) return PRD_ERR_ARG;
    if (!prd_spa_attn_core_supported(N, c, arith)) return PRD_ERR_UNSUPPORTED;
    if (ldq < 4 * H * c || (ldq & 3)) return PRD_ERR_ALIGN;
    if ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(qkvg) | reinterpret_cast<uintptr_t>(ws)) & 15) return PRD_ERR_ALIGN;
    const int nqb = (N + 31) / 32;
    const int KP = spa_key_parts(b, N, H);
    if (KP > 1 && (!ws || ws_bytes < prd_spa_attn_core_workspace(b, N, H, c))) return PRD_ERR_WORKSPACE;
    const int grid = b * H * nqb * KP;
    const size_t lds = spa_lds_bytes(c);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)spa_attn_part_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(spa_attn_part_kernel, dim3(grid), dim3(SPA_NT), lds, stream, o, ws, qkvg, ldq, bias, mask, b, N, H, c, KP, nqb);
    if (KP > 1) {
        const long total = (long)b * N * (H * c / 4);
        hipLaunchKernelGGL(spa_attn_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, o, ws, qkvg, ldq, b, N, H, c, KP);
    }
    return (int)hipGetLastError();
}
