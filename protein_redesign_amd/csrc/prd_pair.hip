// Pair-track kernels, part 1: input stage, pair bias, outer-product update, outer-linear,
// pair transition, coordinate head.  All follow the "lane owns a pair row" scheme of prd_common.h:
// a wave processes 32 pair rows, each row's P channels split over lanes (r, hi); Linear layers run
// as transposed MFMA GEMMs with the weights in LDS, so LayerNorm / gates / residuals stay in
// registers and every pair row is read and written exactly once per operator.
#include "prd_common.h"
#include <cstdlib>
#include "../../include/prd_hip.h"
#include <mutex>

#ifdef PRD_TIMING     // diagnostic builds only (tools/phase_timing.py): cycles per phase summed over the tasks of a wave
__device__ unsigned long long prd_dbg_pair[256 * 16 * 8];
extern "C" int prd_debug_read_pair(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prd_dbg_pair), sizeof(prd_dbg_pair)); }
struct PairPhaseTimer {
    unsigned long long t, acc[8];
    __device__ PairPhaseTimer() { for (int k = 0; k < 8; ++k) acc[k] = 0; t = __builtin_readcyclecounter(); }
    __device__ void mark(int k) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long n = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        acc[k] += n - t;
        t = n;
    }
    __device__ void flush() {
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) prd_dbg_pair[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + k] = acc[k];
    }
};
#else
struct PairPhaseTimer {
    __device__ void mark(int) {}
    __device__ void flush() {}
};
#endif

namespace {

constexpr int WG = 256;   // 4 waves

PRD_DEV void decode_pos(long pos, int N, int& bb, int& i, int& j) {
    const long nn = (long)N * N;
    if (pos < 0x7fffffffL) {                    // 32-bit divisions: a 64-bit one expands to ~130 instructions
        const unsigned p = (unsigned)pos, n2 = (unsigned)nn;
        bb = (int)(p / n2);
        const unsigned rem = p - (unsigned)bb * n2;
        i = (int)(rem / (unsigned)N);
        j = (int)(rem - (unsigned)i * (unsigned)N);
        return;
    }
    bb = (int)(pos / nn);
    const int rem = (int)(pos - (long)bb * nn);
    i = rem / N;
    j = rem - i * N;
}

// ------------------------------------------------------------------------------------------------
// static pair: embeddings gathered once per sample() (model.py:348-358)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void static_pair_kernel(
    float* __restrict__ out, const float* __restrict__ am, const float* __restrict__ rm, const float* __restrict__ bond_mask,
    const int64_t* __restrict__ bond_feats, const int64_t* __restrict__ bond_distance,
    const int64_t* __restrict__ residue_index, const int64_t* __restrict__ chain_index,
    const float* __restrict__ t0, const float* __restrict__ t1, const float* __restrict__ t2,
    const float* __restrict__ tbd, const float* __restrict__ trp, int maxbd, int maxrel, int b, int N, int P) {
    const int F = P / 4;
    const long total = (long)b * N * N * F;
    const float s = (float)(1.0 / sqrt(3.0));
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pos = idx / F;
        const int f = (int)(idx - pos * F);
        int bb, i, j;
        decode_pos(pos, N, bb, i, j);
        const float ami = am[bb * N + i], amj = am[bb * N + j], rmi = rm[bb * N + i], rmj = rm[bb * N + j];
        const float bm = bond_mask[pos];
        const int64_t* bf = bond_feats + pos * 3;
        long bd = bond_distance[pos];
        if (bd > maxbd) bd = maxbd;
        long rel = residue_index[bb * N + i] - residue_index[bb * N + j];
        rel = rel < -maxrel ? -maxrel : (rel > maxrel ? maxrel : rel);
        const float chain = chain_index[bb * N + i] == chain_index[bb * N + j] ? 1.f : 0.f;
        const float4 e0 = *reinterpret_cast<const float4*>(t0 + bf[0] * P + 4 * f);
        const float4 e1 = *reinterpret_cast<const float4*>(t1 + bf[1] * P + 4 * f);
        const float4 e2 = *reinterpret_cast<const float4*>(t2 + bf[2] * P + 4 * f);
        const float4 ed = *reinterpret_cast<const float4*>(tbd + bd * P + 4 * f);
        const float4 er = *reinterpret_cast<const float4*>(trp + (maxrel + rel) * P + 4 * f);
        const float am2 = ami * amj, rm2 = rmi * rmj;
        float4 o;
#define PRD_SP(c) o.c = am2 * (bm * (((0.f + s * e0.c) + s * e1.c) + s * e2.c) + ed.c) + rm2 * (chain * er.c)
        PRD_SP(x); PRD_SP(y); PRD_SP(z); PRD_SP(w);
#undef PRD_SP
        *reinterpret_cast<float4*>(out + pos * P + 4 * f) = o;
    }
}

__global__ __launch_bounds__(256) void atom_embed_kernel(float* __restrict__ out, const int64_t* __restrict__ feats,
                                                         const float* __restrict__ am, const float* __restrict__ tables,
                                                         const int* __restrict__ offsets, int nf, int rows, int S) {
    const long total = (long)rows * S;
    const float s = (float)(1.0 / sqrt((double)nf));
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / S;
        const int c = (int)(idx - row * S);
        float acc = 0.f;
        for (int f = 0; f < nf; ++f) acc += s * tables[(offsets[f] + feats[row * nf + f]) * (long)S + c];
        out[idx] = am[row] * acc;
    }
}

// single = static + rm * relu(W_rt LN(seq_t));  one workgroup per node
__global__ __launch_bounds__(128) void single_init_kernel(float* __restrict__ single, const float* __restrict__ stat,
                                                          const float* __restrict__ seq_t, const float* __restrict__ rm,
                                                          const float* __restrict__ w, int S, int ncls) {
    __shared__ float xs[64];
    const long row = blockIdx.x;
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        const float v = c < ncls ? seq_t[row * ncls + c] : 0.f;
        float s = v;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / ncls;
        const float d = c < ncls ? v - mean : 0.f;
        float q = d * d;
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        xs[c] = d * (1.0f / sqrtf(q / ncls + 1e-5f));
    }
    __syncthreads();
    const float m = rm[row];
    for (int c = threadIdx.x; c < S; c += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < ncls; ++k) acc += xs[k] * w[c * ncls + k];
        single[row * S + c] = stat[row * S + c] + m * relu_nan(acc);
    }
}

__global__ void time_embed_kernel(float* __restrict__ eb, const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                  const float* __restrict__ w, int T, int P, int TD) {
    extern __shared__ float feat[];
    const int bb = blockIdx.x;
    const float tau = (float)t[bb] / (float)T;
    const int half = TD / 2;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        const float wx = freqs[k] * tau;
        feat[k] = sinf(wx);
        feat[half + k] = cosf(wx);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < TD; ++k) acc += feat[k] * w[p * TD + k];
        eb[bb * P + p] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// pair_init: pair = static + m2 * (W_d rbf(d) + ebeta).  B operand (rbf features) generated in
// registers, A operand W_d from LDS; never materialises [N,N,dist_dim].
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(WG) void pair_init_kernel(float* __restrict__ pair, const float* __restrict__ stat,
                                                       const float* __restrict__ z, const float* __restrict__ mask,
                                                       const float* __restrict__ centers, const float* __restrict__ wd,
                                                       const float* __restrict__ ebeta, int b, int N, int DK) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                       // [P][DK+4]
    float* cl = smem + P * (DK + 4);        // [DK]
    stage_weight_plain(Wl, wd, P, DK, DK, 0, threadIdx.x, WG);
    for (int k = threadIdx.x; k < DK; k += WG) cl[k] = centers[k];
    __syncthreads();
    const float scale = (float)((DK - 1) / 2.0);
    const float c2 = -scale * 1.4426950408889634f;
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long rows = (long)b * N * N;
    const long ntask = (rows + 31) / 32;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        int bb = 0, i = 0, j = 0;
        if (valid) decode_pos(pos, N, bb, i, j);
        const float* zi = z + ((long)bb * N + i) * 3;
        const float* zj = z + ((long)bb * N + j) * 3;
        const float dx = zi[0] - zj[0], dy = zi[1] - zj[1], dz = zi[2] - zj[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float m2 = mask[bb * N + i] * mask[bb * N + j];
        f32x16 acc[NB];
        zero_acc(acc);
        const int kb = hi * (DK / 2);
        for (int m = 0; m < DK / 8; ++m) {
            const float4 c4 = *reinterpret_cast<const float4*>(cl + kb + 4 * m);
            float f0 = d - c4.x, f1 = d - c4.y, f2 = d - c4.z, f3 = d - c4.w;
            // exp(-scale x^2) = 2^(c2 x^2) on v_exp_f32 (~1 ulp): expf() expands to ~15 VALU instructions per value, and on
            // gfx950 VALU instructions cost matrix-pipe time (67 -> 22 per 8 MFMAs)
            f0 = __builtin_amdgcn_exp2f(c2 * (f0 * f0)); f1 = __builtin_amdgcn_exp2f(c2 * (f1 * f1));
            f2 = __builtin_amdgcn_exp2f(c2 * (f2 * f2)); f3 = __builtin_amdgcn_exp2f(c2 * (f3 * f3));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 w = *reinterpret_cast<const float4*>(Wl + (nb * 32 + r) * (DK + 4) + kb + 4 * m);
                acc[nb] = mfma32(w.x, f0, acc[nb]);
                acc[nb] = mfma32(w.y, f1, acc[nb]);
                acc[nb] = mfma32(w.z, f2, acc[nb]);
                acc[nb] = mfma32(w.w, f3, acc[nb]);
            }
        }
        float st[KH], eb[KH];
        load_row_cll<P>(stat + pos * P, hi, valid, st);
        load_row_cll<P>(ebeta + bb * P, hi, true, eb);
        float o[KH];
#pragma unroll
        for (int s = 0; s < KH; ++s) o[s] = st[s] + m2 * (acc[s >> 4][s & 15] + eb[s]);
        store_row_cll<P>(pair + pos * P, hi, valid, o);
    }
}

// pair_init on the fp16 matrix pipe (gemm mode 1; dist_dim a multiple of 128): the radial-basis features are generated per K
// step in natural order and split into fp16 hi + lo, W_d sits in LDS as hi | lo planes (prd_common.h: h2_nat_step).
template <int P>
__global__ __launch_bounds__(WG) void pair_init_h2_kernel(float* __restrict__ pair, const float* __restrict__ stat,
                                                          const float* __restrict__ z, const float* __restrict__ mask,
                                                          const float* __restrict__ centers, const float* __restrict__ wd,
                                                          const float* __restrict__ ebeta, int b, int N, int DK) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pi[];
    u32x4* Wh = reinterpret_cast<u32x4*>(smem_pi);          // [2][P][DK/8]
    const int SL = DK / 8;
    float* cl = reinterpret_cast<float*>(Wh + 2 * P * SL);  // [DK]
    stage_weight_h2_nat(Wh, wd, P, DK, DK, 0, threadIdx.x, WG, H2_WSCALE);
    for (int k = threadIdx.x; k < DK; k += WG) cl[k] = centers[k];
    __syncthreads();
    const float scale = (float)((DK - 1) / 2.0);
    const float c2 = -scale * 1.4426950408889634f;
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long rows = (long)b * N * N;
    const long ntask = (rows + 31) / 32;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        int bb = 0, i = 0, j = 0;
        if (valid) decode_pos(pos, N, bb, i, j);
        const float* zi = z + ((long)bb * N + i) * 3;
        const float* zj = z + ((long)bb * N + j) * 3;
        const float dx = zi[0] - zj[0], dy = zi[1] - zj[1], dz = zi[2] - zj[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float m2 = mask[bb * N + i] * mask[bb * N + j];
        f32x16 acc[NB];
        zero_acc(acc);
        for (int st = 0; st < DK / 16; ++st) {
            const float4 c0 = *reinterpret_cast<const float4*>(cl + 16 * st + 8 * hi);
            const float4 c1 = *reinterpret_cast<const float4*>(cl + 16 * st + 8 * hi + 4);
            const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = d - cc[e];
                f[e] = __builtin_amdgcn_exp2f(c2 * (t * t));       // exp(-scale t^2) on v_exp_f32
            }
            h2_nat_step<NB>(Wh, P, SL, st, f, acc, r, hi);
        }
        float st_[KH], eb[KH];
        load_row_cll<P>(stat + pos * P, hi, valid, st_);
        load_row_cll<P>(ebeta + bb * P, hi, true, eb);
        float o[KH];
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) o[s_] = st_[s_] + m2 * (acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + eb[s_]);
        store_row_cll<P>(pair + pos * P, hi, valid, o);
    }
}

// OPM tail on the fp16 matrix pipe (gemm mode 1; C a multiple of 128): the products a_i * b_j are generated per K step.
template <int P>
__global__ __launch_bounds__(WG) void opm_pair_h2_kernel(float* out, const float* pair, const float* __restrict__ ab,
                                                         const float* __restrict__ mask, const float* __restrict__ wo,
                                                         const float* __restrict__ bo, int b, int N, int C, int flags) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_op[];
    u32x4* Wh = reinterpret_cast<u32x4*>(smem_op);          // [2][P][C/8]
    const int SL = C / 8;
    float* bl = reinterpret_cast<float*>(Wh + 2 * P * SL);  // [P] CLL
    stage_weight_h2_nat(Wh, wo, P, C, C, 0, threadIdx.x, WG, H2_WSCALE);
    stage_vec_cll(bl, bo, P, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;            // bb*N + i
        const int bb = (int)(bi / N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const float* ai = ab + bi * 2 * C + 8 * hi;
        const float* bj = ab + ((long)bb * N + jj) * 2 * C + C + 8 * hi;
        f32x16 acc[NB];
        zero_acc(acc);
        float4 a0 = *reinterpret_cast<const float4*>(ai), a1 = *reinterpret_cast<const float4*>(ai + 4);
        float4 b0 = *reinterpret_cast<const float4*>(bj), b1 = *reinterpret_cast<const float4*>(bj + 4);
        const int nst = C / 16;
        for (int st = 0; st < nst; ++st) {
            const int sn = st + 1 < nst ? st + 1 : st;                   // unconditional prefetch (clamped)
            const float4 na0 = *reinterpret_cast<const float4*>(ai + 16 * sn), na1 = *reinterpret_cast<const float4*>(ai + 16 * sn + 4);
            const float4 nb0 = *reinterpret_cast<const float4*>(bj + 16 * sn), nb1 = *reinterpret_cast<const float4*>(bj + 16 * sn + 4);
            const float f[8] = {a0.x * b0.x, a0.y * b0.y, a0.z * b0.z, a0.w * b0.w, a1.x * b1.x, a1.y * b1.y, a1.z * b1.z, a1.w * b1.w};
            h2_nat_step<NB>(Wh, P, SL, st, f, acc, r, hi);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        const float m2 = mask[bi] * mask[(long)bb * N + jj];
        const float norm = m2 + 1e-3f;
        const long off = (bi * N + jj) * P;
        float x[KH];
        load_row_cll<P>(pair + off, hi, valid && (flags & 1), x);
        const float mm = (flags & 2) ? m2 : 1.f;
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) x[s_] = x[s_] + mm * ((acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + bl[hi * KH + s_]) / norm);
        store_row_cll<P>(out + off, hi, valid, x);
    }
}

// ------------------------------------------------------------------------------------------------
// pair bias: [b,H,N,N] = Linear(LN(pair))  (HBM bound: one read of pair, H/P of it written)
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(WG) void pair_bias_kernel(float* __restrict__ out, const float* __restrict__ pair,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ w, const float* __restrict__ bvec,
                                                       float* __restrict__ out2, const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                       const float* __restrict__ w2, const float* __restrict__ bvec2, int H2,
                                                       int b, int N, int H) {
    // (out2 != NULL: a second head set on the same normalised rows -- SPAttention's pair bias and the first folding block's
    // attention bias are both functions of the pair tensor after the outer-product update)
    constexpr int KH = P / 2;
    __shared__ __attribute__((aligned(16))) float wl[2][8 * P];
    __shared__ __attribute__((aligned(16))) float gl[2][P];
    __shared__ __attribute__((aligned(16))) float bl[2][P];
    for (int h = 0; h < H; ++h) stage_vec_cll(wl[0] + h * P, w + h * P, P, threadIdx.x, WG);
    stage_vec_cll(gl[0], gamma, P, threadIdx.x, WG);
    stage_vec_cll(bl[0], beta, P, threadIdx.x, WG);
    if (out2) {
        for (int h = 0; h < H2; ++h) stage_vec_cll(wl[1] + h * P, w2 + h * P, P, threadIdx.x, WG);
        stage_vec_cll(gl[1], gamma2, P, threadIdx.x, WG);
        stage_vec_cll(bl[1], beta2, P, threadIdx.x, WG);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long nn = (long)N * N, rows = (long)b * nn, ntask = (rows + 31) / 32;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float xn[KH];
        load_row_cll<P>(pair + pos * P, hi, valid, xn);
        ln_cll<KH>(xn);
        long bb = (task * 32) / nn, rem = task * 32 - bb * nn + r;       // the division on the scalar unit (wave-uniform task), not per lane
        if (rem >= nn) { rem -= nn; ++bb; }
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            if (set == 1 && !out2) break;
            const float* ga = set ? gamma2 : gamma;
            const float* bv = set ? bvec2 : bvec;
            float* o = set ? out2 : out;
            const int nh = set ? H2 : H;
            float x[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) x[s] = ga ? xn[s] * gl[set][hi * KH + s] + bl[set][hi * KH + s] : xn[s];
            for (int h = 0; h < nh; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < KH; ++s) acc += x[s] * wl[set][h * P + hi * KH + s];
                acc = xhalf_sum(acc);
                if (bv) acc += bv[h];
                if (valid && hi == 0) o[(bb * nh + h) * nn + rem] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Head of the pair track in ONE row pass (gemm mode 1): pair_init -> OuterProductUpdate tail -> the two attention-bias heads
// (model.py:339-361 + models/AF2_modules.py:532-545 + modules.py:395-397 + AF2_modules.py:454-459 + modules.py:300-304).
// All three are row-local in (i, j): as separate launches the pair tensor is written, read + written, and read again (4 U =
// 105 MB at N = 320) and three prologues / tails are paid; here the row stays in registers from the radial-basis GEMM to the
// bias heads and is written once (1 U + the two [H,N,N] outputs).  The arithmetic of each stage is that of its own kernel
// (pair_init_h2_kernel, opm_pair_h2_kernel, pair_bias_kernel), in the same order.  A task = (batch row bi = bb N + i, 32 columns j).
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void pair_head_h2_kernel(
    float* __restrict__ pair, const float* __restrict__ stat, const float* __restrict__ z, const float* __restrict__ mask,
    const float* __restrict__ centers, const float* __restrict__ wd, const float* __restrict__ ebeta, int DK,
    const float* __restrict__ ab, const float* __restrict__ wo, const float* __restrict__ bo, int C, int apply_mask,
    float* __restrict__ out_a, const float* __restrict__ gamma_a, const float* __restrict__ beta_a, const float* __restrict__ w_a,
    const float* __restrict__ bvec_a, int Ha,
    float* __restrict__ out_b, const float* __restrict__ gamma_b, const float* __restrict__ beta_b, const float* __restrict__ w_b,
    const float* __restrict__ bvec_b, int Hb, int b, int N) {
    constexpr int NB = P / 32, KH = P / 2, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ph[];
    u32x4* Wd = reinterpret_cast<u32x4*>(smem_ph);                      // [2][P][DK/8]
    const int SLd = DK / 8, SLo = C / 8;
    u32x4* Wo = Wd + 2 * P * SLd;                                        // [2][P][C/8]
    float* cl = reinterpret_cast<float*>(Wo + 2 * P * SLo);              // [DK]
    float* bl = cl + DK;                                                 // [P] CLL
    float* wl = bl + P;                                                  // [2][8 P] CLL rows of the bias heads
    float* gl = wl + 16 * P;                                             // [2][P]
    float* sl_ = gl + 2 * P;                                             // [2][P]
    stage_weight_h2_nat(Wd, wd, P, DK, DK, 0, threadIdx.x, NT, H2_WSCALE);
    stage_weight_h2_nat(Wo, wo, P, C, C, 0, threadIdx.x, NT, H2_WSCALE);
    for (int k = threadIdx.x; k < DK; k += NT) cl[k] = centers[k];
    stage_vec_cll(bl, bo, P, threadIdx.x, NT);
    for (int h = 0; h < Ha; ++h) stage_vec_cll(wl + h * P, w_a + h * P, P, threadIdx.x, NT);
    for (int h = 0; h < Hb; ++h) stage_vec_cll(wl + 8 * P + h * P, w_b + h * P, P, threadIdx.x, NT);
    if (gamma_a) { stage_vec_cll(gl, gamma_a, P, threadIdx.x, NT); stage_vec_cll(sl_, beta_a, P, threadIdx.x, NT); }
    if (gamma_b) { stage_vec_cll(gl + P, gamma_b, P, threadIdx.x, NT); stage_vec_cll(sl_ + P, beta_b, P, threadIdx.x, NT); }
    __syncthreads();
    const float scale = (float)((DK - 1) / 2.0);
    const float c2 = -scale * 1.4426950408889634f;
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb, nn = (long)N * N;
    for (long task = (long)blockIdx.x * NW + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * NW) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;            // bb*N + i
        const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const long off = (bi * N + jj) * P;
        const float m2 = mask[bi] * mask[(long)bb * N + jj];
        float x[KH];
        {   // ---- pair_init: static part + m2 (W_d rbf(|z_i - z_j|) + ebeta) ----
            const float* zi = z + bi * 3;
            const float* zj = z + ((long)bb * N + jj) * 3;
            const float dx = zi[0] - zj[0], dy = zi[1] - zj[1], dz = zi[2] - zj[2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            f32x16 acc[NB];
            zero_acc(acc);
            for (int st = 0; st < DK / 16; ++st) {
                const float4 c0 = *reinterpret_cast<const float4*>(cl + 16 * st + 8 * hi);
                const float4 c1 = *reinterpret_cast<const float4*>(cl + 16 * st + 8 * hi + 4);
                const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = d - cc[e];
                    f[e] = __builtin_amdgcn_exp2f(c2 * (t * t));
                }
                h2_nat_step<NB>(Wd, P, SLd, st, f, acc, r, hi);
            }
            float st_[KH], eb[KH];                                       // (requested after the GEMM: 64 registers the loop needs; the
            load_row_cll<P>(stat + off, hi, valid, st_);                 // other waves of the SIMD cover the latency)
            load_row_cll<P>(ebeta + bb * P, hi, true, eb);
#pragma unroll
            for (int s_ = 0; s_ < KH; ++s_) x[s_] = st_[s_] + m2 * (acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + eb[s_]);
        }
        {   // ---- OuterProductUpdate tail: x += [m2] (W_o (a_i * b_j) + b_o) / (m2 + 1e-3) ----
            const float* ai = ab + bi * 2 * C + 8 * hi;
            const float* bj = ab + ((long)bb * N + jj) * 2 * C + C + 8 * hi;
            f32x16 acc[NB];
            zero_acc(acc);
            float4 a0 = *reinterpret_cast<const float4*>(ai), a1 = *reinterpret_cast<const float4*>(ai + 4);
            float4 b0 = *reinterpret_cast<const float4*>(bj), b1 = *reinterpret_cast<const float4*>(bj + 4);
            const int nst = C / 16;
            for (int st = 0; st < nst; ++st) {
                const int sn = st + 1 < nst ? st + 1 : st;
                const float4 na0 = *reinterpret_cast<const float4*>(ai + 16 * sn), na1 = *reinterpret_cast<const float4*>(ai + 16 * sn + 4);
                const float4 nb0 = *reinterpret_cast<const float4*>(bj + 16 * sn), nb1 = *reinterpret_cast<const float4*>(bj + 16 * sn + 4);
                const float f[8] = {a0.x * b0.x, a0.y * b0.y, a0.z * b0.z, a0.w * b0.w, a1.x * b1.x, a1.y * b1.y, a1.z * b1.z, a1.w * b1.w};
                h2_nat_step<NB>(Wo, P, SLo, st, f, acc, r, hi);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
            const float norm = m2 + 1e-3f;
            const float mm = apply_mask ? m2 : 1.f;
#pragma unroll
            for (int s_ = 0; s_ < KH; ++s_) x[s_] = x[s_] + mm * ((acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + bl[hi * KH + s_]) / norm);
        }
        store_row_cll<P>(pair + off, hi, valid, x);
        // ---- the two bias heads on LN(x) ----
        ln_cll<KH>(x);
        const long rem = (long)i * N + jj;
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            const float* ga = set ? gamma_b : gamma_a;
            const float* bv = set ? bvec_b : bvec_a;
            float* o = set ? out_b : out_a;
            const int nh = set ? Hb : Ha;
            float y[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) y[s] = ga ? x[s] * gl[set * P + hi * KH + s] + sl_[set * P + hi * KH + s] : x[s];
            for (int h = 0; h < nh; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < KH; ++s) acc += y[s] * wl[set * 8 * P + h * P + hi * KH + s];
                acc = xhalf_sum(acc);
                if (bv) acc += bv[h];
                if (valid && hi == 0) o[((long)bb * nh + h) * nn + rem] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// OPM tail: pair[i,j,:] += m2 * (W_o (a_i*b_j) + b_o) / (m2 + 1e-3)
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(WG) void opm_pair_kernel(float* out, const float* pair, const float* __restrict__ ab,
                                                      const float* __restrict__ mask, const float* __restrict__ wo,
                                                      const float* __restrict__ bo, int b, int N, int C, int flags) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                    // [P][C+4]
    float* bl = smem + P * (C + 4);      // [P] CLL
    stage_weight_plain(Wl, wo, P, C, C, 0, threadIdx.x, WG);
    stage_vec_cll(bl, bo, P, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    for (long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntask; task += (long)gridDim.x * 4) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;            // bb*N + i
        const int bb = (int)(bi / N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const float* ai = ab + bi * 2 * C + hi * (C / 2);
        const float* bj = ab + ((long)bb * N + jj) * 2 * C + C + hi * (C / 2);
        f32x16 acc[NB];
        zero_acc(acc);
        // the a_i / b_j values of step m+1 are in flight while step m is multiplied (unconditional, clamped index)
        float4 a4 = *reinterpret_cast<const float4*>(ai), b4 = *reinterpret_cast<const float4*>(bj);
        const int nm = C / 8;
        for (int m = 0; m < nm; ++m) {
            const int mn = m + 1 < nm ? m + 1 : m;
            const float4 an = *reinterpret_cast<const float4*>(ai + 4 * mn);
            const float4 bn = *reinterpret_cast<const float4*>(bj + 4 * mn);
            const float f0 = a4.x * b4.x, f1 = a4.y * b4.y, f2 = a4.z * b4.z, f3 = a4.w * b4.w;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 w = *reinterpret_cast<const float4*>(Wl + (nb * 32 + r) * (C + 4) + hi * (C / 2) + 4 * m);
                acc[nb] = mfma32(w.x, f0, acc[nb]);
                acc[nb] = mfma32(w.y, f1, acc[nb]);
                acc[nb] = mfma32(w.z, f2, acc[nb]);
                acc[nb] = mfma32(w.w, f3, acc[nb]);
            }
            a4 = an;
            b4 = bn;
        }
        const float m2 = mask[bi] * mask[(long)bb * N + jj];
        const float norm = m2 + 1e-3f;
        const long off = (bi * N + jj) * P;
        float x[KH];
        load_row_cll<P>(pair + off, hi, valid && (flags & 1), x);
        const float mm = (flags & 2) ? m2 : 1.f;
#pragma unroll
        for (int s = 0; s < KH; ++s) x[s] = x[s] + mm * ((acc[s >> 4][s & 15] + bl[hi * KH + s]) / norm);
        store_row_cll<P>(out + off, hi, valid, x);
    }
}

// ------------------------------------------------------------------------------------------------
// outer-linear: pair[i,j,:] += W1 (x_i*x_j) + u_i - u_j + bias.  The [N,N,2S] concat of the
// reference (419 MB at N=320) is never formed: the product operand is generated per MFMA step.
// K = S streamed in chunks of KCH through LDS (weights); x_i / x_j come straight from L2.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(WG) void outer_linear_kernel(float* out, const float* pair, const float* __restrict__ x,
                                                          const float* __restrict__ u, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int b, int N, int S, int residual, int ldu) {
    constexpr int NB = P / 32, KH = P / 2, KCH = 128;
    __shared__ __attribute__((aligned(16))) float Wl[P * (KCH + 4)];
    __shared__ __attribute__((aligned(16))) float bl[P];
    stage_vec_cll(bl, bias, P, threadIdx.x, WG);
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    const long ngroups = (ntask + 3) / 4;
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long task = grp * 4 + (threadIdx.x >> 6);
        const bool live = task < ntask;
        const long tk = live ? task : 0;
        const int vb = (int)(tk % nvb);
        const long bi = tk / nvb;
        const int bb = (int)(bi / N);
        const int j = vb * 32 + r;
        const bool valid = live && j < N;
        const int jj = (j < N) ? j : 0;
        const float* xi = x + bi * S;
        const float* xj = x + ((long)bb * N + jj) * S;
        f32x16 acc[NB];
        zero_acc(acc);
        for (int k0 = 0; k0 < S; k0 += KCH) {
            const int kc = (S - k0 < KCH) ? (S - k0) : KCH;     // multiple of 8
            __syncthreads();
            stage_weight_plain(Wl, w, P, kc, 2 * S, k0, threadIdx.x, WG);
            __syncthreads();
            const int kb = hi * (kc / 2);
            for (int m = 0; m < kc / 8; ++m) {
                const float4 a4 = *reinterpret_cast<const float4*>(xi + k0 + kb + 4 * m);
                const float4 b4 = *reinterpret_cast<const float4*>(xj + k0 + kb + 4 * m);
                const float f0 = a4.x * b4.x, f1 = a4.y * b4.y, f2 = a4.z * b4.z, f3 = a4.w * b4.w;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float4 wv = *reinterpret_cast<const float4*>(Wl + (nb * 32 + r) * (kc + 4) + kb + 4 * m);
                    acc[nb] = mfma32(wv.x, f0, acc[nb]);
                    acc[nb] = mfma32(wv.y, f1, acc[nb]);
                    acc[nb] = mfma32(wv.z, f2, acc[nb]);
                    acc[nb] = mfma32(wv.w, f3, acc[nb]);
                }
            }
        }
        float ui[KH], uj[KH], pr[KH];
        load_row_cll<P>(u + bi * ldu, hi, true, ui);
        load_row_cll<P>(u + ((long)bb * N + jj) * ldu, hi, true, uj);
        const long off = (bi * N + jj) * P;
        load_row_cll<P>(pair + off, hi, valid && residual, pr);
#pragma unroll
        for (int s = 0; s < KH; ++s) pr[s] = pr[s] + (((acc[s >> 4][s & 15] + ui[s]) - uj[s]) + bl[hi * KH + s]);
        store_row_cll<P>(out + off, hi, valid, pr);
    }
}

// outer-linear, resident-weight variant: all of W1 [P][S] stays in LDS (132 KB at S=512, P=64) for the
// lifetime of a persistent workgroup, so the task loop has no workgroup barrier and every wave pulls
// tasks from the device queue on its own.
// The product term W1 (x_i * x_j) is SYMMETRIC in (i, j): a task (i, 32-block of j >= block of i) computes it once and
// writes both out[i,j] = S + u_i - u_j + b and out[j,i] = S + u_j - u_i + b  (55 % of the MFMAs of the full square).
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void outer_linear_res_kernel(int* queue, float* out, const float* pair,
                                                                   const float* __restrict__ x, const float* __restrict__ u,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   int b, int N, int S, int residual, int ldu) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                      // [P][S+4]
    float* bl = smem + P * (S + 4);        // [P] CLL
    stage_weight_plain(Wl, w, P, S, 2 * S, 0, threadIdx.x, NW * 64);
    stage_vec_cll(bl, bias, P, threadIdx.x, NW * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const int npairs = nvb * (nvb + 1) / 2;                // block pairs (ib <= jb)
    const long ntask = (long)b * npairs * 32;              // 32 rows i per block pair
    const int kb = hi * (S / 2);
    WaveTasks tasks(queue, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const int ii = (int)(task & 31);
        const long t2 = task >> 5;
        const int bb = (int)(t2 / npairs);
        int pidx = (int)(t2 - (long)bb * npairs);
        int ib = 0;
        while (pidx >= nvb - ib) { pidx -= nvb - ib; ++ib; }
        const int jb = ib + pidx;
        const int i = ib * 32 + ii;
        if (i >= N) continue;                              // wave-uniform
        const long bi = (long)bb * N + i;
        const int j = jb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const float* xi = x + bi * S + kb;
        const float* xj = x + ((long)bb * N + jj) * S + kb;
        const float* wl = Wl + r * (S + 4) + kb;
        f32x16 acc[NB];
        zero_acc(acc);
        // software pipeline: the x_i / x_j operands of K group g+1 (4 x 16 B per lane and operand) are in
        // flight while the 16*NB MFMAs of group g execute (2048+ cycles >> L2 latency)
        constexpr int G = 4;                               // 16-byte steps per group
        float4 ca[G], cb[G], na[G], nb4[G];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            ca[t] = *reinterpret_cast<const float4*>(xi + 4 * t);
            cb[t] = *reinterpret_cast<const float4*>(xj + 4 * t);
        }
        const int ngroups = S / (8 * G);                   // S is a multiple of 64 on this path
        for (int g = 0; g < ngroups; ++g) {
            // UNCONDITIONAL prefetch (the last iteration re-reads its own group): with the loads under an `if`
            // hipcc merges the two paths into `s_waitcnt vmcnt(0)` in front of the MFMAs and the prefetch is lost
            const int gn = (g + 1 < ngroups) ? g + 1 : g;
#pragma unroll
            for (int t = 0; t < G; ++t) {
                na[t] = *reinterpret_cast<const float4*>(xi + 4 * (G * gn + t));
                nb4[t] = *reinterpret_cast<const float4*>(xj + 4 * (G * gn + t));
            }
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const float f0 = ca[t].x * cb[t].x, f1 = ca[t].y * cb[t].y, f2 = ca[t].z * cb[t].z, f3 = ca[t].w * cb[t].w;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float4 wv = *reinterpret_cast<const float4*>(wl + nb * 32 * (S + 4) + 4 * (G * g + t));
                    acc[nb] = mfma32(wv.x, f0, acc[nb]);
                    acc[nb] = mfma32(wv.y, f1, acc[nb]);
                    acc[nb] = mfma32(wv.z, f2, acc[nb]);
                    acc[nb] = mfma32(wv.w, f3, acc[nb]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < G; ++t) { ca[t] = na[t]; cb[t] = nb4[t]; }
        }
        // epilogue: (i,j) for j >= i ... and the mirrored (j,i) for j > i; positions j < i of the diagonal block belong
        // to the task of row j (which has i in ITS block, i > j)
        float ui[KH], uj[KH], pr[KH];
        load_row_cll<P>(u + bi * ldu, hi, true, ui);
        load_row_cll<P>(u + ((long)bb * N + jj) * ldu, hi, true, uj);
        const bool upper = valid && j >= i, mirror = valid && j > i;
        const long off = (bi * N + jj) * P;
        load_row_cll<P>(pair + off, hi, upper && residual, pr);
#pragma unroll
        for (int s = 0; s < KH; ++s) pr[s] = pr[s] + (((acc[s >> 4][s & 15] + ui[s]) - uj[s]) + bl[hi * KH + s]);
        store_row_cll<P>(out + off, hi, upper, pr);
        const long offm = (((long)bb * N + jj) * N + i) * P;
        load_row_cll<P>(pair + offm, hi, mirror && residual, pr);
#pragma unroll
        for (int s = 0; s < KH; ++s) pr[s] = pr[s] + (((acc[s >> 4][s & 15] + uj[s]) - ui[s]) + bl[hi * KH + s]);
        store_row_cll<P>(out + offm, hi, mirror, pr);
    }
}

// The same on the fp16 matrix pipe (gemm mode 1): W1 as fp16 hi | lo planes (the size of the fp32 image, so it stays resident),
// the generated operand x_i * x_j split per K step, three products per step (prd_common.h: rowgemm_h2 scheme).  K runs in
// natural order: K step s of lane (r, hi) covers k = 16 s + 8 hi .. + 7.  S must be a multiple of 128.
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void outer_linear_res_h2_kernel(float* out, const float* pair,
                                                                      const float* __restrict__ x, const float* __restrict__ u,
                                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                                      int b, int N, int S, int residual, int ldu) {
    constexpr int NB = P / 32, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ol[];
    u32x4* Wh = reinterpret_cast<u32x4*>(smem_ol);          // [2 planes][P rows][S/8 slots]
    const int SL = S / 8;
    float* bl = reinterpret_cast<float*>(Wh + 2 * P * SL);  // [P] CLL
    PairPhaseTimer pt;
    stage_weight_h2_nat(Wh, w, P, S, 2 * S, 0, threadIdx.x, NW * 64, H2_WSCALE);
    stage_vec_cll(bl, bias, P, threadIdx.x, NW * 64);
    __syncthreads();
    pt.mark(6);                                             // 6: prologue (W1 staging, barrier)
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = (N + 31) / 32;
    const int npairs = nvb * (nvb + 1) / 2;                // block pairs (ib <= jb)
    const long ntask = (long)b * npairs * 32;              // 32 rows i per block pair
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const int ii = (int)(task & 31);
        const long t2 = task >> 5;
        const int bb = (int)(t2 / npairs);
        int pidx = (int)(t2 - (long)bb * npairs);
        int ib = 0;
        while (pidx >= nvb - ib) { pidx -= nvb - ib; ++ib; }
        const int jb = ib + pidx;
        const int i = ib * 32 + ii;
        if (i >= N) continue;                              // wave-uniform
        const long bi = (long)bb * N + i;
        const int j = jb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const float* xi = x + bi * S + 8 * hi;
        const float* xj = x + ((long)bb * N + jj) * S + 8 * hi;
        f32x16 acc[NB];
        zero_acc(acc);
        pt.mark(0);                                         // 0: task decode
        // software pipeline: the operands of K step s+1 are in flight while step s is split and multiplied
        float4 ca0 = *reinterpret_cast<const float4*>(xi), ca1 = *reinterpret_cast<const float4*>(xi + 4);
        float4 cb0 = *reinterpret_cast<const float4*>(xj), cb1 = *reinterpret_cast<const float4*>(xj + 4);
        const int nsteps = S / 16;
        for (int st = 0; st < nsteps; ++st) {
            const int sn = (st + 1 < nsteps) ? st + 1 : st;             // unconditional prefetch (clamped)
            const float4 na0 = *reinterpret_cast<const float4*>(xi + 16 * sn), na1 = *reinterpret_cast<const float4*>(xi + 16 * sn + 4);
            const float4 nb0 = *reinterpret_cast<const float4*>(xj + 16 * sn), nb1 = *reinterpret_cast<const float4*>(xj + 16 * sn + 4);
            const float f[8] = {ca0.x * cb0.x, ca0.y * cb0.y, ca0.z * cb0.z, ca0.w * cb0.w,
                                ca1.x * cb1.x, ca1.y * cb1.y, ca1.z * cb1.z, ca1.w * cb1.w};
            h2_nat_step<NB>(Wh, P, SL, st, f, acc, r, hi);
            __builtin_amdgcn_sched_barrier(0);
            ca0 = na0; ca1 = na1; cb0 = nb0; cb1 = nb1;
        }
        pt.mark(1);                                         // 1: K loop (x_i x_j products, splits, MFMAs)
        // epilogue: (i,j) for j >= i and the mirrored (j,i) for j > i (see outer_linear_res_kernel)
        float ui[KH], uj[KH], pr[KH];
        load_row_cll<P>(u + bi * ldu, hi, true, ui);
        load_row_cll<P>(u + ((long)bb * N + jj) * ldu, hi, true, uj);
        const bool upper = valid && j >= i, mirror = valid && j > i;
        const long off = (bi * N + jj) * P;
        load_row_cll<P>(pair + off, hi, upper && residual, pr);
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) pr[s_] = pr[s_] + (((acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + ui[s_]) - uj[s_]) + bl[hi * KH + s_]);
        store_row_cll<P>(out + off, hi, upper, pr);
        pt.mark(2);                                         // 2: u / pair rows, first store
        const long offm = (((long)bb * N + jj) * N + i) * P;
        load_row_cll<P>(pair + offm, hi, mirror && residual, pr);
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) pr[s_] = pr[s_] + (((acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + uj[s_]) - ui[s_]) + bl[hi * KH + s_]);
        store_row_cll<P>(out + offm, hi, mirror, pr);
        pt.mark(3);                                         // 3: mirrored pair rows + store
    }
    pt.mark(7);
    pt.flush();
}

// ------------------------------------------------------------------------------------------------
// pair transition: pair += W2 relu(W1 LN(pair) + b1) + b2  (hidden 4P kept in registers)
// ------------------------------------------------------------------------------------------------
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void pair_transition_kernel(int* queue, float* out, const float* pair, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, long rows, int residual) {
    constexpr int KH = P / 2, HID = 4 * P, HH = HID / 2, NB = P / 32;
    constexpr int PASSES = 4, HBP = HID / 32 / PASSES, HHP = HH / PASSES;     // hidden units handled per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1l = smem;                         // [HID][P+4]
    float* W2l = W1l + HID * (P + 4);          // [P][HID+4]
    float* b1l = W2l + P * (HID + 4);          // [HID] CLL
    float* b2l = b1l + HID;                    // [P] CLL
    stage_weight_cll<P>(W1l, w1, HID, P, threadIdx.x, NW * 64);
    stage_weight_cll<HID>(W2l, w2, P, HID, threadIdx.x, NW * 64);
    stage_vec_cll(b1l, b1, HID, threadIdx.x, NW * 64);
    stage_vec_cll(b2l, b2, P, threadIdx.x, NW * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long ntask = (rows + 31) / 32;
    WaveTasks tasks(queue, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float x[KH];
        load_row_cll<P>(pair + pos * P, hi, valid, x);
        ln_cll<KH>(x);
        // the 4P hidden units are produced and consumed a quarter at a time (CLL elements [q*HHP, (q+1)*HHP)),
        // which keeps the live registers low enough for three waves per SIMD
        f32x16 acc2[NB];
        zero_acc(acc2);
#define PRD_PT_PASS(Q)                                                                                              \
        {                                                                                                           \
            float h[HHP];                                                                                           \
            f32x16 acc[HBP];                                                                                        \
            zero_acc(acc);                                                                                          \
            rowgemm<P, HBP>(W1l + (Q) * HBP * 32 * (P + 4), x, acc, r, hi);                                         \
            _Pragma("unroll") for (int s = 0; s < HHP; ++s)                                                         \
                h[s] = relu_nan(acc[s >> 4][s & 15] + b1l[hi * HH + (Q) * HHP + s]);                             \
            rowgemm_part<HID, NB, (Q) * HHP / 4, ((Q) + 1) * HHP / 4>(W2l, h, acc2, r, hi);                         \
        }
        PRD_PT_PASS(0) PRD_PT_PASS(1) PRD_PT_PASS(2) PRD_PT_PASS(3)
#undef PRD_PT_PASS
        load_row_cll<P>(pair + pos * P, hi, valid && residual, x);      // raw row again (cache hit) for the residual
#pragma unroll
        for (int s = 0; s < KH; ++s) x[s] = x[s] + (acc2[s >> 4][s & 15] + b2l[hi * KH + s]);
        store_row_cll<P>(out + pos * P, hi, valid, x);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused tail of a folding block (reference modules.py:341-342 and the next block's :300-304):
//   pair += W_o og + b_o                     (output projection of the ending triangle attention)
//   pair += W_2 relu(W_1 LN(pair) + b_1) + b_2                                  (pair transition)
//   bias_out[b,h,i,j] = Linear_h(LN(pair))   (attention bias of the NEXT block's single attention; optional)
// All three are row-local, so one pass reads og + pair and writes pair (+ the H-channel bias) instead of three
// kernels reading / writing the pair tensor three times.
// ------------------------------------------------------------------------------------------------
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void block_tail_kernel(int* queue, float* pair, const float* __restrict__ og,
                                                             const float* __restrict__ wo, const float* __restrict__ bo,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const float* __restrict__ wb, const float* __restrict__ bb_,
                                                             float* __restrict__ bias_out, int H, long rows, long nn) {
    constexpr int KH = P / 2, HID = 4 * P, HH = HID / 2, NB = P / 32, HC = 64;
    constexpr int PASSES = 4, HBP = HID / 32 / PASSES, HHP = HH / PASSES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1l = smem;                         // [HID][P+4]
    float* W2l = W1l + HID * (P + 4);          // [P][HID+4]
    float* Wol = W2l + P * (HID + 4);          // [P][HC+4]
    float* b1l = Wol + P * (HC + 4);           // [HID] CLL
    float* b2l = b1l + HID;                    // [P] CLL
    float* bol = b2l + P;                      // [P] CLL
    float* wbl = bol + P;                      // [8][P] CLL (bias head)
    const int NT = NW * 64;
    stage_weight_cll<P>(W1l, w1, HID, P, threadIdx.x, NT);
    stage_weight_cll<HID>(W2l, w2, P, HID, threadIdx.x, NT);
    stage_weight_cll<HC>(Wol, wo, P, HC, threadIdx.x, NT);
    stage_vec_cll(b1l, b1, HID, threadIdx.x, NT);
    stage_vec_cll(b2l, b2, P, threadIdx.x, NT);
    stage_vec_cll(bol, bo, P, threadIdx.x, NT);
    if (bias_out)
        for (int h = 0; h < H; ++h) stage_vec_cll(wbl + h * P, wb + h * P, P, threadIdx.x, NT);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
    const long ntask = (rows + 31) / 32;

    // raw = pair + Wo og + bo for the 32 rows of a task; x = LN(raw)
    auto head = [&](long pos, bool valid, float (&raw)[KH], float (&x)[KH]) {
        float xo[HC / 2];
        load_row_cll<HC>(og + pos * HC, hi, valid, xo);
        f32x16 acc[NB];
        zero_acc(acc);
        rowgemm<HC, NB>(Wol, xo, acc, r, hi);
        load_row_cll<P>(pair + pos * P, hi, valid, raw);
#pragma unroll
        for (int s = 0; s < KH; ++s) raw[s] = raw[s] + (acc[s >> 4][s & 15] + bol[hi * KH + s]);
#pragma unroll
        for (int s = 0; s < KH; ++s) x[s] = raw[s];
        ln_cll<KH>(x);
    };
    // pair row out (raw + transition) and, if asked for, the next block's attention bias from its LayerNorm
    auto tail = [&](long pos, bool valid, float (&raw)[KH], const f32x16 (&acc2)[NB]) {
#pragma unroll
        for (int s = 0; s < KH; ++s) raw[s] = raw[s] + (acc2[s >> 4][s & 15] + b2l[hi * KH + s]);
        store_row_cll<P>(pair + pos * P, hi, valid, raw);
        if (bias_out) {
            ln_cll<KH>(raw);
            const long bb = pos / nn, rem = pos - bb * nn;
            for (int h = 0; h < H; ++h) {
                float a = 0.f;
#pragma unroll
                for (int s = 0; s < KH; ++s) a += raw[s] * wbl[h * P + hi * KH + s];
                a = xhalf_sum(a);
                if (bb_) a += bb_[h];
                if (valid && hi == 0) bias_out[(bb * H + h) * nn + rem] = a;
            }
        }
    };
    // hidden units [Q*HID/PASSES, (Q+1)*HID/PASSES): acc2 += W2[:, q] relu(W1[q, :] x + b1[q])
#define PRD_PT_PASS(Q)                                                                                              \
    {                                                                                                               \
        float h[HHP];                                                                                               \
        f32x16 acc[HBP];                                                                                            \
        zero_acc(acc);                                                                                              \
        rowgemm<P, HBP>(W1l + (Q) * HBP * 32 * (P + 4), x, acc, r, hi);                                             \
        _Pragma("unroll") for (int s = 0; s < HHP; ++s)                                                             \
            h[s] = relu_nan(acc[s >> 4][s & 15] + b1l[hi * HH + (Q) * HHP + s]);                                 \
        rowgemm_part<HID, NB, (Q) * HHP / 4, ((Q) + 1) * HHP / 4>(W2l, h, acc2, r, hi);                             \
    }

    // Static schedule per SIMD (waves w and w+4 share SIMD w): whole 32-row tasks in rounds of 4*gridDim.x, alternating
    // between the two waves of a SIMD.  What is left after the whole rounds -- at N = 320: 128 of 3200 tasks, which used
    // to keep one SIMD of 128 CUs busy for a full task (15.7 us) while everything else idled -- is computed by the four
    // SIMDs of a workgroup TOGETHER: each of waves 0-3 does the head redundantly and one quarter of the hidden units
    // (192 of the task's 576 MFMAs); the partial outputs meet in LDS (over the W1 image, no longer needed) and wave 0
    // finishes the rows.  Chosen only when it is cheaper than one more whole round (leftover <= 2 tasks per workgroup).
    const long slots = (long)gridDim.x * 4;
    const long nfull = ntask / slots;
    const long left = ntask - nfull * slots;
    const bool coop = queue == nullptr && left > 0 && left <= 2 * (long)gridDim.x && NW == 8;
    const long nwhole = coop ? nfull * slots : ntask;
    {
        WaveTasks tasks(queue, ntask, NW);                      // used with a queue only
        const int simd = wave & 3, par = wave >> 2;             // static: rounds 0, 2, .. to wave simd, rounds 1, 3, .. to wave simd + 4
        long rnd = par;
        while (true) {
            long task;
            if (queue) {
                task = tasks.next();
            } else {
                task = rnd * slots < nwhole ? (long)__builtin_amdgcn_readfirstlane((int)(rnd * slots + (long)simd * gridDim.x + blockIdx.x)) : -1;
                if (task >= nwhole) task = -1;
                rnd += NW / 4;
            }
            if (task < 0) break;
            const long pos = task * 32 + r;
            const bool valid = pos < rows;
            float raw[KH], x[KH];
            head(pos, valid, raw, x);
            f32x16 acc2[NB];
            zero_acc(acc2);
            PRD_PT_PASS(0) PRD_PT_PASS(1) PRD_PT_PASS(2) PRD_PT_PASS(3)
            tail(pos, valid, raw, acc2);
        }
    }
    if (coop) {
        float* part = W1l;                                      // [4 waves][64 lanes][KH + 1]
        for (long task = nwhole + blockIdx.x; task < ntask; task += gridDim.x) {     // uniform over the workgroup
            const long pos = task * 32 + r;
            const bool valid = pos < rows;
            float raw[KH], x[KH];
            f32x16 acc2[NB];
            zero_acc(acc2);
            if (wave < 4) {
                head(pos, valid, raw, x);
                if (wave == 0) PRD_PT_PASS(0)
                else if (wave == 1) PRD_PT_PASS(1)
                else if (wave == 2) PRD_PT_PASS(2)
                else PRD_PT_PASS(3)
            }
            __syncthreads();                                    // every wave is done with W1 (whole tasks and this quarter)
            if (wave >= 1 && wave < 4) {
#pragma unroll
                for (int s = 0; s < KH; ++s) part[((wave - 1) * 64 + lane) * (KH + 1) + s] = acc2[s >> 4][s & 15];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int s = 0; s < KH; ++s) acc2[s >> 4][s & 15] += part[(w * 64 + lane) * (KH + 1) + s];
                tail(pos, valid, raw, acc2);
            }
            __syncthreads();                                    // partials consumed before the next cooperative task
            // W1 is overwritten: restore the quarter images for a following cooperative task
            if (task + gridDim.x < ntask) {
                stage_weight_cll<P>(W1l, w1, HID, P, threadIdx.x, NT);
                __syncthreads();
            }
        }
    }
#undef PRD_PT_PASS
}

// ------------------------------------------------------------------------------------------------
// The same row pass on the fp16 matrix pipe (gemm mode 1): every operand split into fp16 hi + lo, three products per GEMM
// (rowgemm_h2, prd_common.h).  The weight images have the size of the fp32 ones, so W1, W2 and W_o stay resident together
// (the bf16 x 3 image, 1.5x larger, would not fit the 160 KB of a CU).  Serves prd_block_tail (og given) and
// prd_pair_transition (og == nullptr: no attention projection, optional residual, out may differ from pair).
//   raw = pair (+ W_o og + b_o);  out = (residual ? raw : 0) + W_2 relu(W_1 LN(raw) + b_1) + b_2;  bias_out = Linear_h(LN(out))
// ------------------------------------------------------------------------------------------------
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void pair_tail_h2_kernel(float* out, const float* pair, const float* __restrict__ og,
                                                               const float* __restrict__ wo, const float* __restrict__ bo,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               const float* __restrict__ wb, const float* __restrict__ bb_,
                                                               float* __restrict__ bias_out, int H, long rows, long nn, int residual) {
    constexpr int KH = P / 2, HID = 4 * P, HH = HID / 2, NB = P / 32, HC = 64;
    constexpr int PASSES = P == 64 ? 8 : 4, HBP = HID / 32 / PASSES, HHP = HH / PASSES;   // hidden units per pass: 32 HBP, per lane HHP
    // (P = 64: eight passes of one 32-unit block keep the kernel inside the 168 registers of a 12-wave workgroup)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    u32x4* W1h = reinterpret_cast<u32x4*>(smem_h2);                 // [2][HID][P/8]
    u32x4* W2h = W1h + 2 * HID * (P / 8);                           // [2][P][HID/8]
    u32x4* Woh = W2h + 2 * P * (HID / 8);                           // [2][P][HC/8]
    float* b1l = reinterpret_cast<float*>(Woh + 2 * P * (HC / 8));  // [HID] CLL
    float* b2l = b1l + HID;                                         // [P] CLL
    float* bol = b2l + P;                                           // [P] CLL
    float* wbl = bol + P;                                           // [8][P] CLL (bias head)
    const int NT = NW * 64;
    PairPhaseTimer pt;
    stage_weight_h2<P>(W1h, w1, HID, P, threadIdx.x, NT, H2_WSCALE);
    stage_weight_h2<HID>(W2h, w2, P, HID, threadIdx.x, NT, H2_WSCALE);
    if (og) stage_weight_h2<HC>(Woh, wo, P, HC, threadIdx.x, NT, H2_WSCALE);
    stage_vec_cll(b1l, b1, HID, threadIdx.x, NT);
    stage_vec_cll(b2l, b2, P, threadIdx.x, NT);
    stage_vec_cll(bol, og ? bo : nullptr, P, threadIdx.x, NT);
    if (bias_out)
        for (int h = 0; h < H; ++h) stage_vec_cll(wbl + h * P, wb + h * P, P, threadIdx.x, NT);
    __syncthreads();
    pt.mark(6);                                         // 6: prologue (weight staging, barrier)
    const int lane = threadIdx.x & 63, r_w = lane & 31, hi_w = lane >> 5;
    const long ntask = (rows + 31) / 32;
    WaveTasks tasks(nullptr, ntask, NW);
    // (requesting the first task's rows before the weight staging was tried: the staging loop then spills, 38.7 -> 50.9 us)
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        // lane coordinates opaque per task: hipcc otherwise hoists lane-dependent 64-bit addresses out of the task loop and, at the
        // 168-register limit of 12 waves, parks them in scratch (12 B / lane, one reload per task)
        int r = r_w, hi = hi_w;
        asm volatile("" : "+v"(r), "+v"(hi));
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float raw[KH];
        load_row_cll<P>(pair + pos * P, hi, valid, raw);
        pt.mark(0);                                     // 0: decode + load issue
        if (og) {
            float xo[HC / 2];
            load_row_cll<HC>(og + pos * HC, hi, valid, xo);
            u32x4 os[2][HC / 16];
            split2h_cll<HC / 2>(xo, os);
            f32x16 acc[NB];
            zero_acc(acc);
            rowgemm_h2<HC, NB>(Woh, P, 0, os, acc, r, hi);
#pragma unroll
            for (int s_ = 0; s_ < KH; ++s_) raw[s_] = raw[s_] + (acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + bol[hi * KH + s_]);
        }
        pt.mark(1);                                     // 1: wait for the rows + attention out-projection
        float x[KH];
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) x[s_] = raw[s_];
        ln_cll<KH>(x);
        u32x4 xs[2][P / 16];
        split2h_cll<KH>(x, xs);
        pt.mark(2);                                     // 2: LayerNorm + split
        f32x16 acc2[NB];
        zero_acc(acc2);
#define PRD_H2_PASS(Q)                                                                                              \
        {                                                                                                           \
            float h[HHP];                                                                                           \
            f32x16 acc[HBP];                                                                                        \
            zero_acc(acc);                                                                                          \
            rowgemm_h2<P, HBP>(W1h, HID, (Q) * HBP * 32, xs, acc, r, hi);                                           \
            _Pragma("unroll") for (int s_ = 0; s_ < HHP; ++s_)                                                      \
                h[s_] = relu_nan(acc[s_ >> 4][s_ & 15] * H2_INV_WSCALE + b1l[hi * HH + (Q) * HHP + s_]);          \
            u32x4 hs[2][HHP / 8];                                                                                   \
            split2h_cll<HHP>(h, hs);                                                                                \
            rowgemm_h2_part<HID, NB, (Q) * HHP / 8, ((Q) + 1) * HHP / 8>(W2h, P, 0, hs, acc2, r, hi);               \
        }
        PRD_H2_PASS(0) PRD_H2_PASS(1) PRD_H2_PASS(2) PRD_H2_PASS(3)
        if constexpr (PASSES == 8) { PRD_H2_PASS(4) PRD_H2_PASS(5) PRD_H2_PASS(6) PRD_H2_PASS(7) }
#undef PRD_H2_PASS
        pt.mark(3);                                     // 3: transition GEMMs
#pragma unroll
        for (int s_ = 0; s_ < KH; ++s_) raw[s_] = (residual ? raw[s_] : 0.f) + (acc2[s_ >> 4][s_ & 15] * H2_INV_WSCALE + b2l[hi * KH + s_]);
        store_row_cll<P>(out + pos * P, hi, valid, raw);
        pt.mark(4);                                     // 4: epilogue + store issue
        if (bias_out) {
            ln_cll<KH>(raw);
            // batch element / position inside it: the 64-bit division once per TASK on the scalar unit (the task id is wave-uniform),
            // a compare per lane -- not a 64-bit VALU division per lane
            const long pos0 = task * 32;
            const long bb0 = pos0 / nn;
            long rem = pos0 - bb0 * nn + r;
            long bbl = bb0;
            if (rem >= nn) { rem -= nn; ++bbl; }
            float* bo_ = bias_out + (bbl * H) * nn + rem;
            for (int h = 0; h < H; ++h) {
                const float4* wv = reinterpret_cast<const float4*>(wbl + h * P + hi * KH);      // 16-byte LDS reads, two addresses per wave
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < KH / 4; ++s4) {
                    const float4 w = wv[s4];
                    a0 += raw[4 * s4] * w.x; a1 += raw[4 * s4 + 1] * w.y; a2 += raw[4 * s4 + 2] * w.z; a3 += raw[4 * s4 + 3] * w.w;
                }
                float a = xhalf_sum((a0 + a1) + (a2 + a3));
                if (bb_) a += bb_[h];
                if (valid && hi == 0) bo_[(long)h * nn] = a;
            }
        }
        pt.mark(5);                                     // 5: next block's attention bias
    }
    pt.mark(7);
    pt.flush();
}

// ------------------------------------------------------------------------------------------------
// coordinate head: one workgroup per (b,i); 4 waves split the j blocks; fixed-order reduction
// ------------------------------------------------------------------------------------------------
template <int P, bool B3>                // B3: the hidden layer on the fp16 matrix pipe (split operands; prd_common.h: rowgemm_h2)
__global__ __launch_bounds__(WG) void coord_head_kernel(float* __restrict__ eps, const float* __restrict__ pair,
                                                        const float* __restrict__ z, const float* __restrict__ mask,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, int b, int N) {
    constexpr int NB = P / 32, KH = P / 2;
    __shared__ __attribute__((aligned(16))) float W1l[P * (P + 4)];      // B3: hi | lo planes of P x P fp16 (the same bytes + padding)
    __shared__ __attribute__((aligned(16))) float b1l[P];
    __shared__ __attribute__((aligned(16))) float w2l[P];
    __shared__ float red[4][3];
    if (B3) stage_weight_h2<P>(reinterpret_cast<u32x4*>(W1l), w1, P, P, threadIdx.x, WG, H2_WSCALE);
    else stage_weight_cll<P>(W1l, w1, P, P, threadIdx.x, WG);
    stage_vec_cll(b1l, b1, P, threadIdx.x, WG);
    stage_vec_cll(w2l, w2, P, threadIdx.x, WG);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
    const int nvb = (N + 31) / 32;
    for (long bi = blockIdx.x; bi < (long)b * N; bi += gridDim.x) {
        const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
        const float zi0 = z[bi * 3], zi1 = z[bi * 3 + 1], zi2 = z[bi * 3 + 2];
        const float mi = mask[bi];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int vb = wave; vb < nvb; vb += 4) {
            const int j = vb * 32 + r;
            const bool valid = j < N;
            const int jj = valid ? j : 0;
            float a[KH], t[KH];
            load_row_cll<P>(pair + (bi * N + jj) * P, hi, valid, a);
            load_row_cll<P>(pair + (((long)bb * N + jj) * N + i) * P, hi, valid, t);
#pragma unroll
            for (int s = 0; s < KH; ++s) a[s] = 0.5f * (a[s] + t[s]);
            ln_cll<KH>(a);
            f32x16 acc[NB];
            zero_acc(acc);
            constexpr float ASC = B3 ? H2_INV_WSCALE : 1.0f;
            if (B3) {
                u32x4 xs[2][P / 16];
                split2h_cll<KH>(a, xs);
                rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(W1l), P, 0, xs, acc, r, hi);
            } else {
                rowgemm<P, NB>(W1l, a, acc, r, hi);
            }
            float wsum = 0.f;
#pragma unroll
            for (int s = 0; s < KH; ++s) wsum += relu_nan(acc[s >> 4][s & 15] * ASC + b1l[hi * KH + s]) * w2l[hi * KH + s];
            wsum = xhalf_sum(wsum);
            const float* zj = z + ((long)bb * N + jj) * 3;
            const float d0 = zi0 - zj[0], d1 = zi1 - zj[1], d2 = zi2 - zj[2];
            const float inv = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-4f);
            const float m2 = (valid && hi == 0) ? mi * mask[(long)bb * N + jj] : 0.f;
            const float f = m2 * wsum;
            s0 += f * (d0 * inv); s1 += f * (d1 * inv); s2 += f * (d2 * inv);
        }
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        __syncthreads();
        if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; }
        __syncthreads();
        if (threadIdx.x < 3)
            eps[bi * 3 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// remove_mean over the node axis; one workgroup per sample
__global__ __launch_bounds__(256) void remove_mean_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                          const float* __restrict__ mask, int N, int D) {
    __shared__ float red[256];
    __shared__ float mean[64];
    const int bb = blockIdx.x;
    float cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) cnt += mask[bb * N + i];
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float norm = red[0];
    __syncthreads();
    for (int d = 0; d < D; ++d) {
        float s = 0.f;
        for (int i = threadIdx.x; i < N; i += 256) s += mask[bb * N + i] * x[((long)bb * N + i) * D + d];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) mean[d] = red[0];
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < N * D; idx += 256) {
        const int i = idx / D, d = idx - i * D;
        const float m = mask[bb * N + i];
        out[(long)bb * N * D + idx] = x[(long)bb * N * D + idx] - m * mean[d] / norm;
    }
}

// reverse-diffusion update; one workgroup per sample.  Also advances the device-side step counter.
__global__ __launch_bounds__(256) void reverse_update_kernel(float* __restrict__ z, float* __restrict__ seq_t,
                                                             int64_t* __restrict__ t, const float* __restrict__ eps,
                                                             const float* __restrict__ seq_pred, const float* __restrict__ noise,
                                                             const float* __restrict__ mask, const float* __restrict__ coef,
                                                             int N, int ncls, int num_steps) {
    __shared__ float red[4][4];
    __shared__ float mean[4];
    const int bb = blockIdx.x;
    const long tt = t[bb];
    const float wn = coef[tt * 4 + 0], isa = coef[tt * 4 + 1], sb = coef[tt * 4 + 2];
    // noise table [T-1][b][N][3]: row (T-1-t) is the draw consumed at step t (t > 0)
    const float* nz = noise + ((long)(tt > 0 ? num_steps - 1 - tt : 0) * gridDim.x + bb) * N * 3;
    if (tt > 0) {       // masked sums of the three noise coordinates and the node count: one pass, wave shuffles, one barrier
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = threadIdx.x; i < N; i += 256) {
            const float m = mask[bb * N + i];
            s[0] += m * nz[i * 3];
            s[1] += m * nz[i * 3 + 1];
            s[2] += m * nz[i * 3 + 2];
            s[3] += m;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[d] += __shfl_xor(s[d], o);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = s[d];
        }
        __syncthreads();
        if (threadIdx.x < 4) mean[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < N * 3; idx += 256) {
        const int i = idx / 3, d = idx - i * 3;
        const long g = (long)bb * N * 3 + idx;
        const float mu = isa * (z[g] - wn * eps[g]);
        float out = mu;
        if (tt > 0) out = mu + sb * (nz[idx] - mask[bb * N + i] * mean[d] / mean[3]);
        z[g] = out;
    }
    for (int i = threadIdx.x; i < N; i += 256) {
        const float* lp = seq_pred + ((long)bb * N + i) * ncls;
        float m = -INFINITY;
        for (int c = 0; c < ncls; ++c) m = fmaxf(m, lp[c]);
        float s = 0.f;
        float* sp = seq_t + ((long)bb * N + i) * ncls;
        for (int c = 0; c < ncls; ++c) { const float e = expf(lp[c] - m); sp[c] = e; s += e; }
        for (int c = 0; c < ncls; ++c) sp[c] = sp[c] / s * 2.f - 1.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) t[bb] = tt - 1;
}

// Step boundary of the reverse-diffusion loop in ONE multi-workgroup launch (instead of remove_mean, reverse_update on a single
// workgroup, and the next step's single_init and time_embed -- four dependent launches):
//   noise_pred = remove_mean(eps_raw)                                   (utils.py:32-36; model.py:373)
//   z <- (z - w_t noise_pred) / sqrt(alpha_t) [+ sqrt(beta_t) remove_mean(noise_t)],  seq_t <- 2 softmax(seq_pred) - 1   (model.py:405-420)
//   t <- t - 1;   ebeta <- W_beta sinus(t / T);   single <- static + rm * relu(W_rt LN(seq_t))    (model.py:341-346, the NEXT step's inputs)
// Workgroups [0, nblk) of a sample own 8 nodes each (a 32-node version ran 42 us: one long chain of dependent L2 round trips
// per workgroup -- the work has to be wide, not deep); the masked means over all nodes are recomputed by every one of them
// (N x 7 loads).  Workgroups [nblk, nblk + P/4) compute four time-embedding outputs each.  The step counter is advanced by the
// LAST workgroup to arrive at `sync[0]` (zero before the first launch; every workgroup reads t before it arrives), which
// also resets the counter for the next replay of the graph.  `sync[1]` is a STICKY flag: set to 1 (never cleared here) when a new
// coordinate or sequence entry is inf / NaN -- under PRD_ARITH_SPLIT16 that is how an operand beyond the fp16 range shows
// (prd_hip.h, OPERAND RANGE); the host reads it once per sample() and re-runs the call under PRD_ARITH_FP32 or raises.
constexpr int SB_NODES = 8;
template <int NCLS>
__global__ __launch_bounds__(256) void step_boundary_kernel(
    float* __restrict__ z, float* __restrict__ seq_t, int64_t* __restrict__ t, const float* __restrict__ eps_raw,
    float* __restrict__ seq_pred, const float* __restrict__ noise, const float* __restrict__ mask, const float* __restrict__ coef,
    float* __restrict__ single_next, const float* __restrict__ stat, const float* __restrict__ rm, const float* __restrict__ w_rt,
    float* __restrict__ ebeta_next, const float* __restrict__ freqs, const float* __restrict__ w_beta, int* __restrict__ sync,
    int b, int N, int num_steps, int S, int P, int TD, int nblk, int neb,
    const float* __restrict__ seq_h, int ldh, const float* __restrict__ w_seq, int Sh) {
    constexpr int ncls = NCLS, CPL = (NCLS + 7) / 8;      // classes per lane in the 8-lanes-per-node softmax
    __shared__ float red[4][8];
    __shared__ float lg[SB_NODES][NCLS + 3];      // seq_h given: the logits of the workgroup's nodes (computed here)
    __shared__ float mean[8];
    __shared__ float xs[SB_NODES][NCLS + 3];     // LayerNorm-ed new seq_t of the workgroup's nodes
    __shared__ float feat[512];
    __shared__ int last;
    const int per = nblk + neb;
    const int bb = blockIdx.x / per, blk = blockIdx.x - bb * per;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long tt = t[bb];
    if (blk >= nblk) {                           // ---- time embedding of the NEXT step (t - 1; unused after the last step) ----
        const float tau = (float)(tt - 1) / (float)num_steps;
        const int half = TD / 2;
        for (int k = tid; k < half; k += 256) {
            const float wx = freqs[k] * tau;
            feat[k] = sinf(wx);
            feat[half + k] = cosf(wx);
        }
        __syncthreads();
        const int p = (blk - nblk) * 4 + wave;   // one wave per output: coalesced reads of W_beta's row, shuffle reduction
        if (p < P) {
            float acc = 0.f;
            for (int k = lane; k < TD; k += 64) acc += feat[k] * w_beta[p * TD + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) ebeta_next[bb * P + p] = acc;
        }
    } else {                                     // ---- 8 nodes: z, seq_t, and the next step's single input ----
        const int i0 = blk * SB_NODES;
        const float wn = coef[tt * 4 + 0], isa = coef[tt * 4 + 1], sb = coef[tt * 4 + 2];
        const float* nz = noise + ((long)(tt > 0 ? num_steps - 1 - tt : 0) * b + bb) * N * 3;
        const float* ep = eps_raw + (long)bb * N * 3;
        {   // masked sums over ALL nodes: predicted noise (3), drawn noise (3), node count
            float s7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < N; i += 256) {
                const float m = mask[bb * N + i];
                s7[0] += m * ep[i * 3]; s7[1] += m * ep[i * 3 + 1]; s7[2] += m * ep[i * 3 + 2];
                s7[3] += m * nz[i * 3]; s7[4] += m * nz[i * 3 + 1]; s7[5] += m * nz[i * 3 + 2];
                s7[6] += m;
            }
#pragma unroll
            for (int d = 0; d < 7; ++d) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s7[d] += __shfl_xor(s7[d], o);
                if (lane == 0) red[wave][d] = s7[d];
            }
        }
        if (seq_h) {
            // the sequence head's LAST layer (model.py:117-122: Linear(S, 21, bias=False) on the ReLU hidden units seq_h) for the
            // workgroup's 8 nodes: 32 threads per node, 16 hidden units each per 512, shuffle reduction -- instead of a GEMM launch
            // of its own for 0.007 GF.  The logits are also written to seq_pred (the loop returns them after the last step).
            const int n = tid >> 5, part = tid & 31, i = i0 + n;
            const bool ok = i < N;
            const float* hr = seq_h + ((long)bb * N + (ok ? i : 0)) * ldh;
            float acc[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) acc[c] = 0.f;
            for (int k = 4 * part; k < Sh; k += 128) {
                const float4 hv = *reinterpret_cast<const float4*>(hr + k);
#pragma unroll
                for (int c = 0; c < NCLS; ++c) {
                    const float4 wv = *reinterpret_cast<const float4*>(w_seq + (long)c * Sh + k);
                    acc[c] += (hv.x * wv.x + hv.y * wv.y) + (hv.z * wv.z + hv.w * wv.w);
                }
            }
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) acc[c] += __shfl_xor(acc[c], o);
            }
            if (part == 0) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c) {
                    lg[n][c] = acc[c];
                    if (ok) seq_pred[((long)bb * N + i) * ncls + c] = acc[c];
                }
            }
            __syncthreads();
        }
        if (tid < 8 * SB_NODES) {   // seq_t: 2 softmax - 1 and its LayerNorm; 8 lanes per node, classes l, l+8, ...
            const int n = tid >> 3, l = tid & 7, i = i0 + n;
            const bool ok = i < N;
            const float* lp = seq_h ? &lg[n][0] : seq_pred + ((long)bb * N + (ok ? i : 0)) * ncls;
            float v[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) v[q] = (l + 8 * q < ncls) ? lp[l + 8 * q] : -INFINITY;
            float mx = v[0];
#pragma unroll
            for (int q = 1; q < CPL; ++q) mx = fmaxf(mx, v[q]);
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { v[q] = (l + 8 * q < ncls) ? expf(v[q] - mx) : 0.f; sum += v[q]; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
            float mu = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { v[q] = (l + 8 * q < ncls) ? v[q] / sum * 2.f - 1.f : 0.f; mu += v[q]; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) mu += __shfl_xor(mu, o);
            mu /= ncls;
            float var = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { const float dlt = (l + 8 * q < ncls) ? v[q] - mu : 0.f; var += dlt * dlt; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) var += __shfl_xor(var, o);
            const float rstd = 1.0f / sqrtf(var / ncls + 1e-5f);
            float* sp = seq_t + ((long)bb * N + (ok ? i : 0)) * ncls;
            bool bad = false;
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                if (l + 8 * q < ncls) {
                    if (ok) sp[l + 8 * q] = v[q];
                    bad |= ok && !(fabsf(v[q]) <= 1.5f);        // 2 softmax - 1 lies in [-1, 1]: anything else is inf / NaN logits
                    xs[n][l + 8 * q] = ok ? (v[q] - mu) * rstd : 0.f;
                }
            if (bad) sync[1] = 1;
        }
        __syncthreads();
        if (tid < 7) mean[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        __syncthreads();
        if (tid >= 64 && tid < 64 + 3 * SB_NODES) {      // z of the workgroup's nodes
            const int q = tid - 64, i = i0 + q / 3, d = q - (q / 3) * 3;
            if (i < N) {
                const long g = ((long)bb * N + i) * 3 + d;
                const float m = mask[bb * N + i];
                const float npred = eps_raw[g] - m * mean[d] / mean[6];
                float out = isa * (z[g] - wn * npred);
                if (tt > 0) out = out + sb * (nz[i * 3 + d] - m * mean[3 + d] / mean[6]);
                z[g] = out;
                if (!(fabsf(out) <= 3.0e38f)) sync[1] = 1;      // sticky: inf / NaN coordinates (prd_hip.h, OPERAND RANGE)
            }
        }
        // single input of the next step: thread -> channels c, c + 256, ...; W_rt row in registers; the static-single / mask loads
        // of the 8 nodes are issued together
        for (int c = tid; c < S; c += 256) {
            float w[NCLS], st8[SB_NODES], rm8[SB_NODES];
#pragma unroll
            for (int u = 0; u < SB_NODES; ++u) {
                const int i = i0 + u;
                const long row = (long)bb * N + (i < N ? i : N - 1);
                st8[u] = stat[row * S + c];
                rm8[u] = rm[row];
            }
#pragma unroll
            for (int k = 0; k < NCLS; ++k) w[k] = w_rt[c * NCLS + k];
#pragma unroll
            for (int u = 0; u < SB_NODES; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NCLS; ++k) acc += xs[u][k] * w[k];
                if (i0 + u < N) single_next[((long)bb * N + i0 + u) * S + c] = st8[u] + rm8[u] * relu_nan(acc);
            }
        }
    }
    // advance the step counters once every workgroup has read them
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const int arrived = atomicAdd(sync, 1);
        last = (arrived == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (last) {
        for (int k = tid; k < b; k += 256) t[k] = t[k] - 1;
        if (tid == 0) atomicExch(sync, 0);
    }
}

int grid_for(long tasks, int per_wg, int cap) {
    long g = (tasks + per_wg - 1) / per_wg;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// raise the dynamic-LDS limit of a kernel to the hardware maximum, once per process and kernel (thread-safe: std::call_once;
// not a stream operation, so it is legal during hipGraph capture)
#define PRD_SET_LDS(kernel, bytes)                                                                              \
    do {                                                                                                        \
        static std::once_flag prd_lds_once;                                                                     \
        std::call_once(prd_lds_once, [] {                                                                       \
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                     \
        (void)(bytes);                                                                                          \
    } while (0)

#define PRD_CHECK_P(P) if ((P) != 32 && (P) != 64) return PRD_ERR_UNSUPPORTED

extern "C" int prd_version(void) { return PRD_VERSION; }

extern "C" int prd_static_pair(float* out, const float* atom_mask, const float* residue_mask, const float* bond_mask,
                               const int64_t* bond_feats, const int64_t* bond_distance,
                               const int64_t* residue_index, const int64_t* chain_index,
                               const float* tab_b0, const float* tab_b1, const float* tab_b2,
                               const float* tab_bdist, const float* tab_relpos,
                               int max_bond_distance, int max_relpos, int b, int N, int P, hipStream_t stream) {
    if (!out || !atom_mask || !residue_mask || !bond_mask || !bond_feats || !bond_distance || !residue_index ||
        !chain_index || !tab_b0 || !tab_b1 || !tab_b2 || !tab_bdist || !tab_relpos || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P & 3) return PRD_ERR_ALIGN;
    const long total = (long)b * N * N * (P / 4);
    hipLaunchKernelGGL(static_pair_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, stream, out, atom_mask,
                       residue_mask, bond_mask, bond_feats, bond_distance, residue_index, chain_index, tab_b0, tab_b1,
                       tab_b2, tab_bdist, tab_relpos, max_bond_distance, max_relpos, b, N, P);
    return (int)hipGetLastError();
}

extern "C" int prd_atom_embed(float* out, const int64_t* atom_feats, const float* atom_mask, const float* tables,
                              const int* offsets, int n_feats, int b, int N, int S, hipStream_t stream) {
    if (!out || !atom_feats || !atom_mask || !tables || !offsets || n_feats <= 0 || b <= 0 || N <= 0 || S <= 0) return PRD_ERR_ARG;
    hipLaunchKernelGGL(atom_embed_kernel, dim3(grid_for((long)b * N * S, 256, 2048)), dim3(256), 0, stream, out,
                       atom_feats, atom_mask, tables, offsets, n_feats, b * N, S);
    return (int)hipGetLastError();
}

extern "C" int prd_single_init(float* single, const float* static_single, const float* seq_t, const float* residue_mask,
                               const float* w_rt, int rows, int S, int n_cls, hipStream_t stream) {
    if (!single || !static_single || !seq_t || !residue_mask || !w_rt || rows <= 0 || S <= 0 || n_cls <= 0 || n_cls > 64) return PRD_ERR_ARG;
    hipLaunchKernelGGL(single_init_kernel, dim3(rows), dim3(128), 0, stream, single, static_single, seq_t, residue_mask, w_rt, S, n_cls);
    return (int)hipGetLastError();
}

extern "C" int prd_time_embed(float* ebeta, const int64_t* t, const float* freqs, const float* w_beta,
                              int num_steps, int b, int P, int time_dim, hipStream_t stream) {
    if (!ebeta || !t || !freqs || !w_beta || num_steps <= 0 || b <= 0 || P <= 0 || time_dim <= 0 || (time_dim & 1)) return PRD_ERR_ARG;
    hipLaunchKernelGGL(time_embed_kernel, dim3(b), dim3(64), time_dim * sizeof(float), stream, ebeta, t, freqs, w_beta, num_steps, P, time_dim);
    return (int)hipGetLastError();
}

extern "C" int prd_pair_init(float* pair, const float* static_pair, const float* z, const float* mask,
                             const float* centers, const float* w_dist, const float* ebeta,
                             int b, int N, int P, int dist_dim, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!pair || !static_pair || !z || !mask || !centers || !w_dist || !ebeta || b <= 0 || N <= 0) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    if (dist_dim <= 0 || (dist_dim & 7)) return PRD_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)P * (dist_dim + 4) + dist_dim) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const long ntask = ((long)b * N * N + 31) / 32;
    const int grid = grid_for(ntask, 4, 512);
    if (arith == PRD_ARITH_SPLIT16 && (dist_dim % 128) == 0) {        // fp16 x 2 split operands
        const size_t lds2 = (size_t)4 * P * dist_dim + (size_t)dist_dim * 4;
        if (P == 64) {
            PRD_SET_LDS(pair_init_h2_kernel<64>, lds2);
            hipLaunchKernelGGL(pair_init_h2_kernel<64>, dim3(grid), dim3(WG), lds2, stream, pair, static_pair, z, mask, centers, w_dist, ebeta, b, N, dist_dim);
        } else {
            PRD_SET_LDS(pair_init_h2_kernel<32>, lds2);
            hipLaunchKernelGGL(pair_init_h2_kernel<32>, dim3(grid), dim3(WG), lds2, stream, pair, static_pair, z, mask, centers, w_dist, ebeta, b, N, dist_dim);
        }
        return (int)hipGetLastError();
    }
    if (P == 64) {
        PRD_SET_LDS(pair_init_kernel<64>, lds);
        hipLaunchKernelGGL(pair_init_kernel<64>, dim3(grid), dim3(WG), lds, stream, pair, static_pair, z, mask, centers, w_dist, ebeta, b, N, dist_dim);
    } else {
        PRD_SET_LDS(pair_init_kernel<32>, lds);
        hipLaunchKernelGGL(pair_init_kernel<32>, dim3(grid), dim3(WG), lds, stream, pair, static_pair, z, mask, centers, w_dist, ebeta, b, N, dist_dim);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_pair_bias(float* bias_out, const float* pair, const float* gamma, const float* beta,
                             const float* w, const float* bvec, int b, int N, int P, int H, hipStream_t stream) {
    if (!bias_out || !pair || !w || b <= 0 || N <= 0 || H <= 0 || H > 8 || (gamma && !beta)) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    const long ntask = ((long)b * N * N + 31) / 32;
    const int grid = grid_for(ntask, 4, 2048);
    const float* nul = nullptr;
    if (P == 64) hipLaunchKernelGGL(pair_bias_kernel<64>, dim3(grid), dim3(WG), 0, stream, bias_out, pair, gamma, beta, w, bvec, (float*)nullptr, nul, nul, nul, nul, 0, b, N, H);
    else hipLaunchKernelGGL(pair_bias_kernel<32>, dim3(grid), dim3(WG), 0, stream, bias_out, pair, gamma, beta, w, bvec, (float*)nullptr, nul, nul, nul, nul, 0, b, N, H);
    return (int)hipGetLastError();
}

extern "C" int prd_pair_bias2(float* bias_a, const float* pair, const float* gamma_a, const float* beta_a, const float* w_a,
                              const float* bvec_a, int Ha, float* bias_b, const float* gamma_b, const float* beta_b, const float* w_b,
                              const float* bvec_b, int Hb, int b, int N, int P, hipStream_t stream) {
    if (!bias_a || !bias_b || !pair || !w_a || !w_b || b <= 0 || N <= 0 || Ha <= 0 || Ha > 8 || Hb <= 0 || Hb > 8 ||
        (gamma_a && !beta_a) || (gamma_b && !beta_b)) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    const long ntask = ((long)b * N * N + 31) / 32;
    const int grid = grid_for(ntask, 4, 2048);
    if (P == 64) hipLaunchKernelGGL(pair_bias_kernel<64>, dim3(grid), dim3(WG), 0, stream, bias_a, pair, gamma_a, beta_a, w_a, bvec_a,
                                    bias_b, gamma_b, beta_b, w_b, bvec_b, Hb, b, N, Ha);
    else hipLaunchKernelGGL(pair_bias_kernel<32>, dim3(grid), dim3(WG), 0, stream, bias_a, pair, gamma_a, beta_a, w_a, bvec_a,
                            bias_b, gamma_b, beta_b, w_b, bvec_b, Hb, b, N, Ha);
    return (int)hipGetLastError();
}

extern "C" int prd_pair_head_supported(int P, int dist_dim, int C, int arith) {
    if (arith < 0 || (arith & 0xff) != PRD_ARITH_SPLIT16 || (P != 32 && P != 64)) return 0;
    if (dist_dim <= 0 || (dist_dim % 128) || C <= 0 || (C % 128)) return 0;
    const size_t lds = (size_t)4 * P * dist_dim + (size_t)4 * P * C + ((size_t)dist_dim + P + 16 * P + 4 * P) * 4;
    return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int prd_pair_head(float* pair, const float* static_pair, const float* z, const float* mask, const float* centers,
                             const float* w_dist, const float* ebeta, int dist_dim, const float* ab, const float* w_out,
                             const float* b_out, int C, int apply_mask, float* bias_a, const float* gamma_a, const float* beta_a,
                             const float* w_a, const float* bvec_a, int Ha, float* bias_b, const float* gamma_b, const float* beta_b,
                             const float* w_b, const float* bvec_b, int Hb, int b, int N, int P, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!pair || !static_pair || !z || !mask || !centers || !w_dist || !ebeta || !ab || !w_out || !b_out || !bias_a || !bias_b ||
        !w_a || !w_b || b <= 0 || N <= 0 || Ha <= 0 || Ha > 8 || Hb <= 0 || Hb > 8 || (gamma_a && !beta_a) || (gamma_b && !beta_b))
        return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    if (!prd_pair_head_supported(P, dist_dim, C, arith)) return PRD_ERR_UNSUPPORTED;
    constexpr int NWH = 8;
    const size_t lds = (size_t)4 * P * dist_dim + (size_t)4 * P * C + ((size_t)dist_dim + P + 16 * P + 4 * P) * 4;
    const long ntask = (long)b * N * prd_ceil_div(N, 32);
    const int grid = grid_for(ntask, NWH, 256);
    if (P == 64) {
        PRD_SET_LDS((pair_head_h2_kernel<64, NWH>), lds);
        hipLaunchKernelGGL((pair_head_h2_kernel<64, NWH>), dim3(grid), dim3(NWH * 64), lds, stream, pair, static_pair, z, mask, centers, w_dist,
                           ebeta, dist_dim, ab, w_out, b_out, C, apply_mask, bias_a, gamma_a, beta_a, w_a, bvec_a, Ha, bias_b, gamma_b, beta_b,
                           w_b, bvec_b, Hb, b, N);
    } else {
        PRD_SET_LDS((pair_head_h2_kernel<32, NWH>), lds);
        hipLaunchKernelGGL((pair_head_h2_kernel<32, NWH>), dim3(grid), dim3(NWH * 64), lds, stream, pair, static_pair, z, mask, centers, w_dist,
                           ebeta, dist_dim, ab, w_out, b_out, C, apply_mask, bias_a, gamma_a, beta_a, w_a, bvec_a, Ha, bias_b, gamma_b, beta_b,
                           w_b, bvec_b, Hb, b, N);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_opm_pair(float* out, const float* pair, const float* ab, const float* mask, const float* w_out,
                            const float* b_out, int flags, int b, int N, int P, int C, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!out || !pair || !ab || !mask || !w_out || !b_out || b <= 0 || N <= 0) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    if (C <= 0 || (C & 7)) return PRD_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)P * (C + 4) + P) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const long ntask = (long)b * N * prd_ceil_div(N, 32);
    const int grid = grid_for(ntask, 4, 1024);
    if (arith == PRD_ARITH_SPLIT16 && (C % 128) == 0) {               // fp16 x 2 split operands
        const size_t lds2 = (size_t)4 * P * C + (size_t)P * 4;
        if (P == 64) {
            PRD_SET_LDS(opm_pair_h2_kernel<64>, lds2);
            hipLaunchKernelGGL(opm_pair_h2_kernel<64>, dim3(grid), dim3(WG), lds2, stream, out, pair, ab, mask, w_out, b_out, b, N, C, flags);
        } else {
            PRD_SET_LDS(opm_pair_h2_kernel<32>, lds2);
            hipLaunchKernelGGL(opm_pair_h2_kernel<32>, dim3(grid), dim3(WG), lds2, stream, out, pair, ab, mask, w_out, b_out, b, N, C, flags);
        }
        return (int)hipGetLastError();
    }
    if (P == 64) {
        PRD_SET_LDS(opm_pair_kernel<64>, lds);
        hipLaunchKernelGGL(opm_pair_kernel<64>, dim3(grid), dim3(WG), lds, stream, out, pair, ab, mask, w_out, b_out, b, N, C, flags);
    } else {
        PRD_SET_LDS(opm_pair_kernel<32>, lds);
        hipLaunchKernelGGL(opm_pair_kernel<32>, dim3(grid), dim3(WG), lds, stream, out, pair, ab, mask, w_out, b_out, b, N, C, flags);
    }
    return (int)hipGetLastError();
}

namespace {
// OuterLinear with the K axis (single_dim) split over the eight waves of a workgroup (modules.py:283-287, split form
// out[i,j,:] = W1 (x_i * x_j) + u_i - u_j + bias; symmetric half: the product term of (i, j) and (j, i) is the same).
//
// Why: the round-2 kernel (outer_linear_res_h2_kernel) gives one wave a whole (i, 32 j) tile and walks the 512 channels in 32
// dependent K steps.  Its time is neither MFMA (4.4 us of work) nor VALU: every operand / pair-row access is "one row per lane"
// (16 bytes from each of 32-64 cache lines per instruction), and the texture-address path of a CU serialises a wave instruction
// by cache line: ~10k cycles of address processing per tile (tools/phase_timing.py), 33 us per launch.  Here
//   * a TILE belongs to a workgroup; wave w takes channels [w S/8, (w+1) S/8); its slice of W1 lives in REGISTERS as MFMA A
//     operands (fp16 hi | lo, 64 VGPRs at S = 512): no LDS image, no staging barrier;
//   * a task = 8 rows i of one (i-block, j-block) pair: the x_j slice is fetched ONCE per task, COALESCED (16 lanes x 16 B = the
//     256 contiguous bytes a row contributes to the wave's slice, 4 rows per instruction: 8 lines instead of 64), turned into
//     the row-per-lane operand layout through a wave-private LDS tile (row pitch 272 B: conflict-free b128 reads) and kept in
//     registers for the 8 rows -- the operand traffic per output drops 8-fold; x_i and the pair rows of row i+1 are requested
//     while row i is reduced;
//   * the eight partial accumulators meet in LDS; wave g then finishes ROWS 4g .. 4g+3 of the tile, 16 lanes per row, so that the
//     pair rows are read and written as whole 256-byte rows (both the (i, j) rows and the mirrored (j, i) rows).
template <int P, int KS>          // KS = K steps (16 channels each) per wave: S = 128 KS
__global__ __launch_bounds__(512) void outer_linear_ks_kernel(float* out, const float* pair, const float* __restrict__ x,
                                                              const float* __restrict__ u, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int b, int N, int residual, int ldu) {
    constexpr int NB = P / 32, NW = 8, S = 128 * KS, NG = NB * 4;      // NG = 4-register groups of the accumulator
    constexpr int CW = 16 * KS;                                         // channels per wave
    constexpr int XP = CW * 4 + 16;                                     // staging row pitch in bytes (+16: bank spread)
    constexpr int LR = CW / 4;                                          // lanes per row of a coalesced slice load (16-byte pieces)
    constexpr int RPI = 64 / LR;                                        // rows per load instruction
    constexpr int NLD = 32 / RPI;                                       // load instructions per 32-row slice
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ks[];
    float4* part = reinterpret_cast<float4*>(smem_ks);                  // [2 buffers][NW waves][NG groups][64 lanes]
    float4* stage = part;                                               // [NW waves][32 rows][XP bytes]: task start only, aliased
    constexpr int PSZ = NW * NG * 64;                                   // float4 per partial buffer
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PairPhaseTimer pt;
    const int k0 = wave * CW;                                           // first channel of this wave's slice
    float4* mystage = stage + (size_t)wave * 32 * (XP / 16);             // 16-byte units
    // ---- this wave's slice of W1 as A operands: lane (row, hi) holds W1[32 nb + row][k0 + 16 s + 8 hi .. + 8], x 16 ----
    u32x4 wh[NB][KS], wl[NB][KS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const float* wp = w + (size_t)(32 * nb + r) * (2 * S) + k0 + 16 * st + 8 * hi;
            const float4 g0 = *reinterpret_cast<const float4*>(wp), g1 = *reinterpret_cast<const float4*>(wp + 4);
            const float v[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned a_, b_;
                split2h(H2_WSCALE * v[2 * q], H2_WSCALE * v[2 * q + 1], a_, b_);
                wh[nb][st][q] = a_;
                wl[nb][st][q] = b_;
            }
        }
    const int nvb = (N + 31) / 32;
    const int npairs = nvb * (nvb + 1) / 2;                 // block pairs (ib <= jb)
    constexpr int RG = 8;                                   // rows i per task: the x_j slice is fetched once for RG rows
    const int ntask = b * npairs * (32 / RG);
    struct Task { int bb, i0, jb; };                        // wave-uniform
    auto decode = [&](int t) {
        Task T;
        const int grp = t % (32 / RG), t2 = t / (32 / RG);
        T.bb = t2 / npairs;
        int pidx = t2 - T.bb * npairs, ib = 0;
        while (pidx >= nvb - ib) { pidx -= nvb - ib; ++ib; }
        T.jb = ib + pidx;
        T.i0 = ib * 32 + grp * RG;
        return T;
    };
    const int lane_w = lane;
    struct Epi { float4 ui, pu, pm; };
    pt.mark(6);                                             // 6: prologue (W1 slice -> registers)
    for (int t = blockIdx.x; t < ntask; t += gridDim.x) {
        const Task T = decode(t);
        // Lane coordinates opaque per task: the lane-dependent offsets below are a dozen VALU instructions per task; hoisted out of
        // the task loop they sat in registers next to the W1 slices (256 VGPRs) and eight of them went to scratch (32 B / lane).
        int lane = lane_w;
        asm volatile("" : "+v"(lane));
        const int r = lane & 31, hi = lane >> 5;
        const int lrow = lane / LR, lpiece = lane % LR;
        // epilogue: wave g owns rows 4g .. 4g+3 of the tile; lane = (row 4g + lane / 16, 16-byte piece lane % 16 of the P channels).
        // For P = 32 (8 pieces per row) the upper half of every 16-lane group idles.
        const int erow = 4 * wave + (lane >> 4), epiece = lane & 15;
        const bool elive = epiece < P / 4;
        const int ech = 4 * (elive ? epiece : 0);
        // channel ech .. ech+3 sits in accumulator group (ech >> 5) * 4 + ((ech >> 3) & 3) of the lane of row erow with hi = (ech >> 2) & 1
        const int egrp = (ech >> 5) * 4 + ((ech >> 3) & 3), esrc = erow + 32 * ((ech >> 2) & 1);
        // The partial tile of (wave, group g) is stored lane-linear EXCEPT that the low four lane bits are XORed with (2 g + hi): the
        // sixteen lanes of a ds_read_b128 group of the reduction (two tile rows x 8 channel pieces: same low lane bits, all eight
        // (group, hi) combinations) then read sixteen different 16-byte bank slots.  Unswizzled they hit two slots, 8-way: 67 % of the
        // kernel's LDS cycles were bank conflicts (profiles/r03_roofline.txt).
        const int eoff = egrp * 64 + ((esrc & 48) | ((esrc & 15) ^ ((2 * egrp + (esrc >> 5)) & 15)));
        const float4 bs = *reinterpret_cast<const float4*>(bias + ech);
        // ---- x_j slice of the block, once per task: coalesced -> wave-private LDS tile -> row-per-lane operands ----
        {
            float4 xj[NLD];
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int j = T.jb * 32 + RPI * q + lrow;
                const int jc = j < N ? j : N - 1;
                xj[q] = *reinterpret_cast<const float4*>(x + ((size_t)T.bb * N + jc) * S + k0 + 4 * lpiece);
            }
#pragma unroll
            for (int q = 0; q < NLD; ++q)      // (a whole-struct copy of the HIP float4 keeps the array in scratch)
                mystage[(RPI * q + lrow) * (XP / 16) + lpiece] = make_float4(xj[q].x, xj[q].y, xj[q].z, xj[q].w);
        }
        wave_lds_fence();
        float4 xb[KS][2];
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            xb[st][0] = mystage[r * (XP / 16) + 4 * st + 2 * hi];
            xb[st][1] = mystage[r * (XP / 16) + 4 * st + 2 * hi + 1];
        }
        const int je = T.jb * 32 + erow;                    // this lane's row j in the epilogue
        const bool jvalid = elive && je < N;
        const int jj = jvalid ? je : 0;
        const float4 uj = *reinterpret_cast<const float4*>(u + ((size_t)T.bb * N + jj) * ldu + ech);
        auto row_ok = [&](int i) { return i < N; };
        auto fetch_x = [&](int i, float4 (&xa)[KS][2]) {
            const float* xi = x + ((size_t)T.bb * N + (row_ok(i) ? i : 0)) * S + k0 + 8 * hi;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                xa[st][0] = *reinterpret_cast<const float4*>(xi + 16 * st);
                xa[st][1] = *reinterpret_cast<const float4*>(xi + 16 * st + 4);
            }
        };
        auto fetch_epi = [&](int i, Epi& e) {
            const bool ok = row_ok(i);
            const int ic = ok ? i : 0;
            const bool upper = ok && jvalid && je >= i, mirror = ok && jvalid && je > i;
            e.ui = *reinterpret_cast<const float4*>(u + ((size_t)T.bb * N + ic) * ldu + ech);
            e.pu = make_float4(0.f, 0.f, 0.f, 0.f);
            e.pm = e.pu;
            if (upper && residual) e.pu = *reinterpret_cast<const float4*>(pair + (((size_t)T.bb * N + ic) * N + jj) * P + ech);
            if (mirror && residual) e.pm = *reinterpret_cast<const float4*>(pair + (((size_t)T.bb * N + jj) * N + ic) * P + ech);
        };
        float4 xa[KS][2];
        Epi ecur, enext;
        fetch_x(T.i0, xa);
        __syncthreads();                                    // every wave has its x_j slice: the staging tiles become partial buffers
        pt.mark(4);                                         // 4: task decode, x_j slice through LDS, first row's requests
        // rows are software-pipelined: the MFMAs of row ii run while the partials of row ii-1 are reduced and stored; the
        // partial buffers alternate, so one LDS barrier per row suffices
        auto finish = [&](int i, const float4* pb, const Epi& e) {
            if (!row_ok(i)) return;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) {
                const float4 v = pb[(size_t)w2 * NG * 64 + eoff];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const bool upper = jvalid && je >= i, mirror = jvalid && je > i;
            const float4 ui = e.ui, pu = e.pu, pm = e.pm;
            if (upper)
                *reinterpret_cast<float4*>(out + (((size_t)T.bb * N + i) * N + je) * P + ech) =
                    make_float4(pu.x + (((sum.x * H2_INV_WSCALE + ui.x) - uj.x) + bs.x), pu.y + (((sum.y * H2_INV_WSCALE + ui.y) - uj.y) + bs.y),
                                pu.z + (((sum.z * H2_INV_WSCALE + ui.z) - uj.z) + bs.z), pu.w + (((sum.w * H2_INV_WSCALE + ui.w) - uj.w) + bs.w));
            if (mirror)
                *reinterpret_cast<float4*>(out + (((size_t)T.bb * N + je) * N + i) * P + ech) =
                    make_float4(pm.x + (((sum.x * H2_INV_WSCALE + uj.x) - ui.x) + bs.x), pm.y + (((sum.y * H2_INV_WSCALE + uj.y) - ui.y) + bs.y),
                                pm.z + (((sum.z * H2_INV_WSCALE + uj.z) - ui.z) + bs.z), pm.w + (((sum.w * H2_INV_WSCALE + uj.w) - ui.w) + bs.w));
        };
        for (int ii = 0; ii < RG; ++ii) {
            const int i = T.i0 + ii;
            fetch_epi(i, enext);                            // needed one iteration later
            // ---- partial product over this wave's channels ----
            f32x16 acc[NB];
            zero_acc(acc);
            if (row_ok(i)) {
#pragma unroll
                for (int st = 0; st < KS; ++st) {
                    const float f[8] = {xa[st][0].x * xb[st][0].x, xa[st][0].y * xb[st][0].y, xa[st][0].z * xb[st][0].z, xa[st][0].w * xb[st][0].w,
                                        xa[st][1].x * xb[st][1].x, xa[st][1].y * xb[st][1].y, xa[st][1].z * xb[st][1].z, xa[st][1].w * xb[st][1].w};
                    u32x4 ph, pl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned a_, b_;
                        split2h(f[2 * q], f[2 * q + 1], a_, b_);
                        ph[q] = a_;
                        pl[q] = b_;
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh[nb][st]), __builtin_bit_cast(f16x8_t, ph), acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wh[nb][st]), __builtin_bit_cast(f16x8_t, pl), acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wl[nb][st]), __builtin_bit_cast(f16x8_t, ph), acc[nb], 0, 0, 0);
                }
            }
            if (ii + 1 < RG) fetch_x(i + 1, xa);
            pt.mark(0);                                     // 0: products, MFMA issue, next row's requests
            if (ii > 0) finish(i - 1, part + (size_t)((ii - 1) & 1) * PSZ, ecur);      // while the MFMAs drain
            ecur = enext;
            pt.mark(3);                                     // 3: reduction + epilogue of the previous row
            {
                float4* pw = part + (size_t)(ii & 1) * PSZ + ((size_t)wave * NG) * 64;
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    pw[g * 64 + ((lane & 48) | ((lane & 15) ^ ((2 * g + hi) & 15)))] = make_float4(acc[g >> 2][4 * (g & 3)], acc[g >> 2][4 * (g & 3) + 1], acc[g >> 2][4 * (g & 3) + 2],
                                                     acc[g >> 2][4 * (g & 3) + 3]);
            }
            pt.mark(1);                                     // 1: partial stores
            // LDS-only barrier: __syncthreads() would also drain the global loads just requested and the previous row's stores
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            pt.mark(2);                                     // 2: barrier
        }
        finish(T.i0 + RG - 1, part + (size_t)((RG - 1) & 1) * PSZ, ecur);
        __syncthreads();                                    // partials consumed before the next task's staging overwrites them
    }
    pt.mark(7);
    pt.flush();
}
}  // namespace

extern "C" int prd_outer_linear(float* out, const float* pair, const float* x, const float* u, int ldu, const float* w,
                                const float* bias, int residual, int b, int N, int P, int S, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!out || !pair || !x || !u || !w || !bias || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (ldu < P || (ldu & 3)) return PRD_ERR_ALIGN;           // u rows are read in 16-byte pieces
    PRD_CHECK_P(P);
    if (S <= 0 || (S & 7)) return PRD_ERR_UNSUPPORTED;
    const long ntask = (long)b * N * prd_ceil_div(N, 32);
    const size_t lds = ((size_t)P * (S + 4) + P) * sizeof(float);
    const bool ol_v1 = PRD_TGET_OL_GEN2(tune);          // A/B switch: the round-2 kernel
    if (arith == PRD_ARITH_SPLIT16 && !ol_v1 && (S == 128 || S == 256 || S == 512) && (long)b * N * N < (1L << 30)) {
        // K split over the waves of a workgroup, W1 slices in registers (outer_linear_ks_kernel)
        const int nvb = prd_ceil_div(N, 32);
        const long ntile = (long)b * (nvb * (nvb + 1) / 2) * 4;       // (block pair, 8 rows i) tasks
        const int grid = (int)(ntile < 256 ? ntile : 256);
        const size_t ldsp = (size_t)2 * 8 * (P / 32 * 4) * 64 * 16, ldss = (size_t)8 * 32 * ((S / 8) * 4 + 16);
        const size_t ldsk = ldsp > ldss ? ldsp : ldss;             // two partial buffers; the staging tiles alias them
#define PRD_OLKS(PP, KK)                                                                                               \
        do {                                                                                                           \
            PRD_SET_LDS((outer_linear_ks_kernel<PP, KK>), ldsk);                                                       \
            hipLaunchKernelGGL((outer_linear_ks_kernel<PP, KK>), dim3(grid), dim3(512), ldsk, stream, out, pair, x, u, w, bias, b, N, residual, ldu); \
        } while (0)
        if (P == 64) { if (S == 512) PRD_OLKS(64, 4); else if (S == 256) PRD_OLKS(64, 2); else PRD_OLKS(64, 1); }
        else { if (S == 512) PRD_OLKS(32, 4); else if (S == 256) PRD_OLKS(32, 2); else PRD_OLKS(32, 1); }
#undef PRD_OLKS
        return (int)hipGetLastError();
    }
    if (arith == PRD_ARITH_SPLIT16 && (S % 128) == 0 && (size_t)4 * P * S + 4 * P <= 160 * 1024) {   // fp16 x 2 split operands
        constexpr int NWL = 8;
        const int nvb = prd_ceil_div(N, 32);
        const long nsym = (long)b * (nvb * (nvb + 1) / 2) * 32;
        const int grid = grid_for(nsym, 4, 256);
        const size_t lds2 = (size_t)4 * P * S + 4 * P;
        if (P == 64) {
            PRD_SET_LDS((outer_linear_res_h2_kernel<64, NWL>), lds2);
            hipLaunchKernelGGL((outer_linear_res_h2_kernel<64, NWL>), dim3(grid), dim3(NWL * 64), lds2, stream, out, pair, x, u, w, bias, b, N, S, residual, ldu);
        } else {
            PRD_SET_LDS((outer_linear_res_h2_kernel<32, NWL>), lds2);
            hipLaunchKernelGGL((outer_linear_res_h2_kernel<32, NWL>), dim3(grid), dim3(NWL * 64), lds2, stream, out, pair, x, u, w, bias, b, N, S, residual, ldu);
        }
        return (int)hipGetLastError();
    }
    if (lds <= 150 * 1024 && (S % 64) == 0) {   // W1 resident in LDS: persistent 8-wave workgroups, queue-fed
        constexpr int NWL = 8;
        const int nvb = prd_ceil_div(N, 32);
        const long nsym = (long)b * (nvb * (nvb + 1) / 2) * 32;      // symmetric half: (i, j-block >= i-block) tasks
        const int grid = grid_for(nsym, 4, 256);
        // fewer tasks than resident waves (symmetric half): the static assignment beats the queue (56 vs 74 us at N = 320)
        int* oq = (nsym > (long)grid * NWL) ? queue : nullptr;
        if (P == 64) {
            PRD_SET_LDS((outer_linear_res_kernel<64, NWL>), lds);
            hipLaunchKernelGGL((outer_linear_res_kernel<64, NWL>), dim3(grid), dim3(NWL * 64), lds, stream, oq, out, pair, x, u, w, bias, b, N, S, residual, ldu);
        } else {
            PRD_SET_LDS((outer_linear_res_kernel<32, NWL>), lds);
            hipLaunchKernelGGL((outer_linear_res_kernel<32, NWL>), dim3(grid), dim3(NWL * 64), lds, stream, oq, out, pair, x, u, w, bias, b, N, S, residual, ldu);
        }
        return (int)hipGetLastError();
    }
    const int grid = grid_for(ntask, 4, 1024);       // very wide single track: stream W1 through LDS in K chunks
    if (P == 64) hipLaunchKernelGGL(outer_linear_kernel<64>, dim3(grid), dim3(WG), 0, stream, out, pair, x, u, w, bias, b, N, S, residual, ldu);
    else hipLaunchKernelGGL(outer_linear_kernel<32>, dim3(grid), dim3(WG), 0, stream, out, pair, x, u, w, bias, b, N, S, residual, ldu);
    return (int)hipGetLastError();
}

namespace {
template <int P>
int launch_pair_tail_h2(float* out, const float* pair, const float* og, const float* wo, const float* bo, const float* w1,
                        const float* b1, const float* w2, const float* b2, const float* bias_w, const float* bias_b,
                        float* bias_out, int H, long rows, long nn, int residual, hipStream_t stream) {
    constexpr int NWH = 12;
    const size_t lds = ((size_t)2 * 4 * P * (P / 8) + (size_t)2 * P * (4 * P / 8) + (size_t)2 * P * 8) * 16 + (size_t)(4 * P + 2 * P + 8 * P) * 4;
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const int grid = grid_for((rows + 31) / 32, 4, 256);
    PRD_SET_LDS((pair_tail_h2_kernel<P, NWH>), lds);
    hipLaunchKernelGGL((pair_tail_h2_kernel<P, NWH>), dim3(grid), dim3(NWH * 64), lds, stream, out, pair, og, wo, bo, w1, b1, w2, b2,
                       bias_w, bias_b, bias_out, H, rows, nn, residual);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int prd_pair_transition(float* out, const float* pair, const float* w1, const float* b1, const float* w2,
                                   const float* b2, int residual, int b, int N, int P, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!out || !pair || !w1 || !b1 || !w2 || !b2 || b <= 0 || N <= 0) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    if (arith == PRD_ARITH_SPLIT16) {             // split 16-bit operands (fp16 x 2), see pair_tail_h2_kernel
        const long rows_ = (long)b * N * N;
        return P == 64 ? launch_pair_tail_h2<64>(out, pair, nullptr, nullptr, nullptr, w1, b1, w2, b2, nullptr, nullptr, nullptr, 0, rows_, (long)N * N, residual, stream)
                       : launch_pair_tail_h2<32>(out, pair, nullptr, nullptr, nullptr, w1, b1, w2, b2, nullptr, nullptr, nullptr, 0, rows_, (long)N * N, residual, stream);
    }
    constexpr int NWT = 12;                    // one persistent 12-wave workgroup per CU (weights: 137 KB of LDS at P=64)
    const long rows = (long)b * N * N;
    const size_t lds = ((size_t)4 * P * (P + 4) + (size_t)P * (4 * P + 4) + 5 * P) * sizeof(float);
    const int grid = grid_for((rows + 31) / 32, 4, 256);
    if (P == 64) {
        PRD_SET_LDS((pair_transition_kernel<64, NWT>), lds);
        hipLaunchKernelGGL((pair_transition_kernel<64, NWT>), dim3(grid), dim3(NWT * 64), lds, stream, queue, out, pair, w1, b1, w2, b2, rows, residual);
    } else {
        PRD_SET_LDS((pair_transition_kernel<32, NWT>), lds);
        hipLaunchKernelGGL((pair_transition_kernel<32, NWT>), dim3(grid), dim3(NWT * 64), lds, stream, queue, out, pair, w1, b1, w2, b2, rows, residual);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_block_tail(float* pair, const float* og, const float* wo, const float* bo, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* bias_w, const float* bias_b, float* bias_out,
                              int b, int N, int P, int H, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!pair || !og || !wo || !bo || !w1 || !b1 || !w2 || !b2 || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (bias_out && (!bias_w || H <= 0 || H > 8)) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    if (arith == PRD_ARITH_SPLIT16) {             // split 16-bit operands (fp16 x 2), see pair_tail_h2_kernel
        const long rows_ = (long)b * N * N;
        return P == 64 ? launch_pair_tail_h2<64>(pair, pair, og, wo, bo, w1, b1, w2, b2, bias_w, bias_b, bias_out, H, rows_, (long)N * N, 1, stream)
                       : launch_pair_tail_h2<32>(pair, pair, og, wo, bo, w1, b1, w2, b2, bias_w, bias_b, bias_out, H, rows_, (long)N * N, 1, stream);
    }
    constexpr int NWT = 8;
    const long rows = (long)b * N * N;
    const size_t lds = ((size_t)4 * P * (P + 4) + (size_t)P * (4 * P + 4) + (size_t)P * 68 + 6 * P + 8 * P) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const int grid = grid_for((rows + 31) / 32, 4, 256);
    // measured at N = 320: static round-robin 87 us, queue 93 us (8-wave workgroups, < 2 tasks per wave); queue beyond that
    int* bq = ((rows + 31) / 32 > (long)2 * grid * NWT) ? queue : nullptr;
    if (P == 64) {
        PRD_SET_LDS((block_tail_kernel<64, NWT>), lds);
        hipLaunchKernelGGL((block_tail_kernel<64, NWT>), dim3(grid), dim3(NWT * 64), lds, stream, bq, pair, og, wo, bo, w1, b1, w2, b2, bias_w, bias_b, bias_out, H, rows, (long)N * N);
    } else {
        PRD_SET_LDS((block_tail_kernel<32, NWT>), lds);
        hipLaunchKernelGGL((block_tail_kernel<32, NWT>), dim3(grid), dim3(NWT * 64), lds, stream, bq, pair, og, wo, bo, w1, b1, w2, b2, bias_w, bias_b, bias_out, H, rows, (long)N * N);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_coord_head(float* eps_raw, const float* pair, const float* z, const float* mask,
                              const float* w1, const float* b1, const float* w2, int b, int N, int P, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!eps_raw || !pair || !z || !mask || !w1 || !b1 || !w2 || b <= 0 || N <= 0) return PRD_ERR_ARG;
    PRD_CHECK_P(P);
    const int grid = grid_for((long)b * N, 1, 2048);
    const bool b3 = arith == PRD_ARITH_SPLIT16;
#define PRD_CH(PP, BB) hipLaunchKernelGGL((coord_head_kernel<PP, BB>), dim3(grid), dim3(WG), 0, stream, eps_raw, pair, z, mask, w1, b1, w2, b, N)
    if (P == 64) { if (b3) PRD_CH(64, true); else PRD_CH(64, false); }
    else { if (b3) PRD_CH(32, true); else PRD_CH(32, false); }
#undef PRD_CH
    return (int)hipGetLastError();
}

extern "C" int prd_remove_mean(float* out, const float* x, const float* mask, int b, int N, int D, hipStream_t stream) {
    if (!out || !x || !mask || b <= 0 || N <= 0 || D <= 0 || D > 64) return PRD_ERR_ARG;
    hipLaunchKernelGGL(remove_mean_kernel, dim3(b), dim3(256), 0, stream, out, x, mask, N, D);
    return (int)hipGetLastError();
}

extern "C" int prd_reverse_update(float* z, float* seq_t, int64_t* t, const float* noise_pred, const float* seq_pred,
                                  const float* noise, const float* mask, const float* coef,
                                  int b, int N, int n_cls, int num_steps, hipStream_t stream) {
    if (!z || !seq_t || !t || !noise_pred || !seq_pred || !noise || !mask || !coef || b <= 0 || N <= 0 || n_cls <= 0) return PRD_ERR_ARG;
    hipLaunchKernelGGL(reverse_update_kernel, dim3(b), dim3(256), 0, stream, z, seq_t, t, noise_pred, seq_pred, noise, mask, coef, N, n_cls, num_steps);
    return (int)hipGetLastError();
}

extern "C" int prd_step_boundary(float* z, float* seq_t, int64_t* t, const float* eps_raw, float* seq_pred,
                                 const float* noise, const float* mask, const float* coef,
                                 float* single_next, const float* static_single, const float* residue_mask, const float* w_rt,
                                 float* ebeta_next, const float* freqs, const float* w_beta, int* sync,
                                 int b, int N, int n_cls, int num_steps, int S, int P, int time_dim,
                                 const float* seq_h, int ldh, const float* w_seq, int S_h, hipStream_t stream) {
    if (!z || !seq_t || !t || !eps_raw || !seq_pred || !noise || !mask || !coef || !single_next || !static_single || !residue_mask ||
        !w_rt || !ebeta_next || !freqs || !w_beta || !sync || b <= 0 || N <= 0 || S <= 0 || P <= 0) return PRD_ERR_ARG;
    if (n_cls != 21 || time_dim <= 0 || time_dim > 512 || (time_dim & 1)) return PRD_ERR_UNSUPPORTED;    // 20 residue types + 'X'
    if (seq_h && (!w_seq || S_h <= 0)) return PRD_ERR_ARG;
    if (seq_h && ((S_h & 3) || (ldh & 3) || ldh < S_h)) return PRD_ERR_ALIGN;
    const int nblk = prd_ceil_div(N, SB_NODES), neb = prd_ceil_div(P, 4);
    hipLaunchKernelGGL(step_boundary_kernel<21>, dim3(b * (nblk + neb)), dim3(256), 0, stream, z, seq_t, t, eps_raw, seq_pred, noise, mask, coef,
                       single_next, static_single, residue_mask, w_rt, ebeta_next, freqs, w_beta, sync, b, N, num_steps, S, P,
                       time_dim, nblk, neb, seq_h, ldh, w_seq, S_h);
    return (int)hipGetLastError();
}
