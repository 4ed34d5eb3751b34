// Pair-track kernels, part 2: triangle multiplication and triangle attention.
//
// Triangle multiplication (reference modules.py:262-274) = three launches:
//   tri_mul_proj     : p = LN(pair); ab = m2 * sigmoid(Wg p + bg) * (Wp p + bp), stored CHANNEL-MAJOR AB[b][2P][N][ldn]
//                      (ldn = round_up(N,32), zero padded) so that the contraction is P independent, K-contiguous N x N x N
//                      GEMMs.  "incoming" writes the transposed operand (a[k,i] -> A[i][k]): both modes share one contraction.
//                      Tasks are (32-row block, 32-output block) pairs, rows prefetched one task ahead.
//   tri_mul_contract : O[b][d][i][j] = sum_k A[d][i][k] B[d][j][k]; persistent 64x64 tiles, double-buffered LDS, buffer loads,
//                      the tiles of one channel on one XCD.
//   tri_mul_out      : pair += sigmoid(Wog LN(pair) + bog) * (Wo LN_d(O) + bo); persistent waves, leftover tasks computed
//                      cooperatively by the four SIMDs of a workgroup.
// Triangle attention (modules.py:236-243 -> 185-225) = two launches:
//   tri_attn_core    : one PERSISTENT workgroup per CU serves one head for a strided set of pair rows.  Phase 1 projects
//                      k | v | q | gate of the LayerNorm-ed row into LDS; phase 2 streams the keys in blocks of 64 on the
//                      16x16x4 fp32 MFMA in the swapped form (S^T = K Q^T, O^T = V^T P^T: every softmax quantity of a query is
//                      lane-local), softmax in the exp2 domain with a frozen reference maximum after the first block.  The
//                      N x N logits never exist in memory.  Rows too long for Q / gate tiles in LDS use
//                      tri_attn_core_long (queries re-projected per 32-query block, operands moved by wave shuffles).
//   tri_attn_out     : pair += Wo og + bo (starting mode; the ending mode's projection is fused into block_tail, prd_pair.hip).
// single_attn_core (single-track attention with pair bias, heads of width 16) lives here too: it shares the key-loop scheme.
#include "prd_common.h"
#include "../../include/prd_hip.h"
#include <atomic>
#include <cstdlib>
#include <mutex>

#ifdef PRD_TIMING     // diagnostic builds only (tools/ta_timing.py, tools/phase_timing.py): in-kernel cycle stamps
__device__ unsigned long long prd_dbg[256 * 16 * 8 * 4];
__device__ int prd_dbg_sel;        // which kernel's PhaseTimer is recorded (0: every one, the last launch wins)
extern "C" int prd_debug_read(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(prd_dbg), sizeof(prd_dbg)); }
extern "C" int prd_debug_select(int id) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(prd_dbg_sel), &id, sizeof(int)); }
// tri_attn_core: [wg][12 waves][8 iterations][4 stamps]
#define PRD_STAMP(k) do { if (lane == 0 && it < 8) prd_dbg[((blockIdx.x * 12 + wave) * 8 + it) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
// row kernels: cycles per phase summed over the tasks of a wave, [wg][16 waves][8 phases]
struct PhaseTimer {
    unsigned long long t, acc[8];
    __device__ PhaseTimer() { for (int k = 0; k < 8; ++k) acc[k] = 0; t = __builtin_readcyclecounter(); }
    __device__ void mark(int k) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long n = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        acc[k] += n - t;
        t = n;
    }
    __device__ void flush(int id = 0) {
        if (prd_dbg_sel != 0 && prd_dbg_sel != id) return;
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) prd_dbg[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + k] = acc[k];
    }
};
#else
#define PRD_STAMP(k)
struct PhaseTimer {
    __device__ void mark(int) {}
    __device__ void flush(int = 0) {}
};
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// position of a tri_mul_proj task; every field is wave-uniform (scalar registers, scalar index arithmetic; 32-bit on
// purpose: a 64-bit division expands to ~130 instructions)
struct ProjTask {
    int ob, vb, bb, u, bu;
    bool ok;
};
template <int OB>
PRD_DEV ProjTask proj_decode(long task, int nrb, int nvb, int N) {
    ProjTask t;
    const int ti = task < 0 ? 0 : (int)task;
    const int grp = ti / (8 * OB), within = ti % (8 * OB);
    t.ob = within >> 3;
    int rb = grp * 8 + (within & 7);
    t.ok = task >= 0 && rb < nrb;               // false: past the end, or padding of the last group of 8 row blocks
    if (!t.ok) rb = 0;                          // such tasks still prefetch (a valid row) but do not compute
    t.vb = rb % nvb;
    t.bu = rb / nvb;                            // bb*N + u
    t.bb = t.bu / N;
    t.u = t.bu - t.bb * N;
    return t;
}
// row + mask loads of a task: issued one task ahead, so that they are OLDER than the stores of the task computed
// meanwhile -- vmcnt retires in order, a load issued after those stores would wait for all of them (measured: 57 % of
// the wave's time in tri_mul_proj before this prefetch)
template <int P>
PRD_DEV void proj_fetch(const ProjTask& t, const float* __restrict__ pair, const float* __restrict__ mask, int N, int incoming,
                        int r, int hi, float (&x)[P / 2], float& mu, float& mv) {
    const int v = t.vb * 32 + r;
    const bool valid = v < N;
    const int vv = valid ? v : 0;
    mu = mask[t.bu];
    mv = mask[t.bb * N + vv];
    // outgoing: operand row u, contraction index v <-> pair[u, v]; incoming: pair[v, u].  Descriptor base = pair of batch
    // element bb (uniform); lane offset = the row's position inside it
    const prd_rsrc rs = make_rsrc(pair + (long)t.bb * N * N * P);
    const unsigned rowi = incoming ? (unsigned)vv * N + t.u : (unsigned)t.u * N + vv;
    load_row_cll_buf<P>(rs, valid ? (rowi * P + 4 * hi) * 4u : BUF_OOB, x);
}

template <int P, bool B3>
PRD_DEV void proj_compute(const ProjTask& t, float (&x)[P / 2], float mu, float mv, float* __restrict__ AB,
                          const float* Wpl, const float* Wgl, const float* bpl, const float* bgl,
                          int N, int ldn, long cstride, unsigned lane_off, int r, int hi, PhaseTimer& pt) {
    constexpr int KH = P / 2, OUT = 2 * P;
    const bool valid = t.vb * 32 + r < N;
    pt.mark(0);                                     // 0: task fetch / decode, prefetch issue
    const float m2 = valid ? mu * mv : 0.f;
    const bool plain = __all(m2 == 1.0f);           // every row of the block valid and unmasked (the common case)
    pt.mark(1);                                     // 1: wait for this task's row + masks
    ln_cll<KH>(x);
    pt.mark(2);                                     // 2: LayerNorm
    f32x16 ap[1], ag[1];
    bias_acc(ap, bpl + hi * P + 16 * t.ob);         // biases ride in the accumulators
    bias_acc(ag, bgl + hi * P + 16 * t.ob);
    if (B3) {                                       // fp16 x 2 form (prd_common.h: rowgemm_h2); weights and biases staged x 16
        u32x4 xs[2][P / 16];
        split2h_cll<KH>(x, xs);
        rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wpl), OUT, t.ob * 32, xs, ap, r, hi);
        rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wgl), OUT, t.ob * 32, xs, ag, r, hi);
#pragma unroll
        for (int q = 0; q < 16; ++q) { ap[0][q] *= H2_INV_WSCALE; ag[0][q] *= H2_INV_WSCALE; }
    } else {
        rowgemm<P, 1>(Wpl + t.ob * 32 * (P + 4), x, ap, r, hi);
        rowgemm<P, 1>(Wgl + t.ob * 32 * (P + 4), x, ag, r, hi);
    }
    pt.mark(3);                                     // 3: MFMAs
    // output channel of register q: 32*ob + (q&3) + 8*(q>>2) + 4*hi; the hi part sits in lane_off
    const prd_rsrc cb = make_rsrc(AB + ((((long)t.bb * 2 * P) + 32 * t.ob) * N + t.u) * ldn + t.vb * 32);
    const unsigned cbytes = (unsigned)cstride * 4u;
    if (plain) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const f32x2 den = f32x2{__builtin_amdgcn_exp2f(ag[0][q]), __builtin_amdgcn_exp2f(ag[0][q + 1])} + 1.0f;
            const f32x2 val = f32x2{ap[0][q], ap[0][q + 1]} * f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            buf_store(val.x, cb, lane_off, ((q & 3) + 8 * (q >> 2)) * cbytes);
            buf_store(val.y, cb, lane_off, (((q + 1) & 3) + 8 * ((q + 1) >> 2)) * cbytes);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            buf_store(m2 * (gate_from_scaled(ag[0][q]) * ap[0][q]), cb, lane_off, ((q & 3) + 8 * (q >> 2)) * cbytes);
    }
    pt.mark(4);                                     // 4: epilogue + store issue
}

template <int P, int NW, bool B3>
__global__ __launch_bounds__(NW * 64) void tri_mul_proj_kernel(int* queue, float* __restrict__ AB, const float* __restrict__ pair,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ wp, const float* __restrict__ bp,
                                                          const float* __restrict__ wg, const float* __restrict__ bg,
                                                          int b, int N, int ldn, int incoming) {
    constexpr int KH = P / 2, OUT = 2 * P, OB = OUT / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    PhaseTimer pt;
    constexpr int WSZ = B3 ? OUT * P : OUT * (P + 4);    // floats per staged weight matrix (fp16 hi | lo planes, or fp32 rows)
    float* Wpl = smem;                       // fp32: [2P][P+4]; bf16 x 3: 3 planes of [2P] rows (prd_common.h)
    float* Wgl = Wpl + WSZ;
    float* bpl = Wgl + WSZ;                  // [2P] CLL
    float* bgl = bpl + OUT;
    if (B3) {
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wpl), wp, OUT, P, threadIdx.x, NW * 64, H2_WSCALE);
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wgl), wg, OUT, P, threadIdx.x, NW * 64, NEG_LOG2E * H2_WSCALE);
    } else {
        stage_weight_cll<P>(Wpl, wp, OUT, P, threadIdx.x, NW * 64);
        stage_weight_cll<P>(Wgl, wg, OUT, P, threadIdx.x, NW * 64, NEG_LOG2E);
    }
    stage_vec_cll(bpl, bp, OUT, threadIdx.x, NW * 64, B3 ? H2_WSCALE : 1.0f);                 // biases ride in the accumulators
    stage_vec_cll(bgl, bg, OUT, threadIdx.x, NW * 64, NEG_LOG2E * (B3 ? H2_WSCALE : 1.0f));
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const int nvb = ldn / 32;
    const int nrb = b * N * nvb;                        // 32-row blocks
    // A task is one 32-output block `ob` of one 32-row block: at N = 320 there are only 3.1 row blocks per SIMD, and
    // whole-block tasks leave the SIMDs that draw a 4th one running alone (28 % of the launch).  The row is re-read
    // (L2) and re-normalised per task; the four tasks of a row block sit 8 task ids apart = same XCD, different CUs.
    const long ntask = (long)((nrb + 7) / 8 * 8) * OB;
    const long cstride = (long)N * ldn;                 // channel stride of AB
    const unsigned lane_off = (hi * 4 * (unsigned)cstride + r) * 4u;   // lane part (bytes) of every store address: channel 4*hi, column r
    WaveTasks tasks(queue, ntask, NW);
    // two row buffers, used alternately: the row of the NEXT task is in flight while the current one is computed
    float xa[KH], xb[KH], mua, mva, mub, mvb;
    long t = tasks.next();
    ProjTask ta = proj_decode<OB>(t, nrb, nvb, N), tb;
    proj_fetch<P>(ta, pair, mask, N, incoming, r, hi, xa, mua, mva);        // overlaps the weight staging
    __syncthreads();
    pt.mark(6);                                     // 6: prologue (weight staging, first fetch, barrier)
    while (t >= 0) {
        const long tn = tasks.next();
        tb = proj_decode<OB>(tn, nrb, nvb, N);
        proj_fetch<P>(tb, pair, mask, N, incoming, r, hi, xb, mub, mvb);
        if (ta.ok) proj_compute<P, B3>(ta, xa, mua, mva, AB, Wpl, Wgl, bpl, bgl, N, ldn, cstride, lane_off, r, hi, pt);
        if (tn < 0) break;
        t = tasks.next();
        ta = proj_decode<OB>(t, nrb, nvb, N);
        proj_fetch<P>(ta, pair, mask, N, incoming, r, hi, xa, mua, mva);
        if (tb.ok) proj_compute<P, B3>(tb, xb, mub, mvb, AB, Wpl, Wgl, bpl, bgl, N, ldn, cstride, lane_off, r, hi, pt);
    }
    pt.mark(5);
    pt.flush(1);
}

// Output stage of the OUTGOING triangle multiplication fused with the projection stage of the INCOMING one that follows it
// in the folding block (modules.py:336-337; gemm mode 1): both are row-local, so the updated pair row never leaves the
// registers between them -- one launch and one pass over the pair tensor less per block (tri_mul_out 19 us + tri_mul_proj
// 24 us, most of the latter launch / prologue / imbalance rather than arithmetic).  The incoming projection reads pair[v, u]
// for 32 consecutive v (its operand layout is [channel][u][v]), so the tasks run DOWN the columns: (u, 32 rows v).  For
// those tasks the contraction output must be contiguous in v, which the outgoing contraction provides by swapping its
// operands (O^T = B A^T).  Per task: Ot[:, u, v-block] and pair[v-block, u] -> gate, projection of LN(O), residual ->
// pair[v-block, u] (written back) -> LN -> a | b projections and gates of the incoming module, masked -> AB[:, u, v-block].
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_mul_out_proj_kernel(float* pair, const float* __restrict__ Ot, const float* __restrict__ mask,
                                                                   const float* __restrict__ wo, const float* __restrict__ bo,
                                                                   const float* __restrict__ wog, const float* __restrict__ bog,
                                                                   const float* __restrict__ wp, const float* __restrict__ bp,
                                                                   const float* __restrict__ wg, const float* __restrict__ bg,
                                                                   float* __restrict__ AB, int b, int N, int ldn) {
    constexpr int KH = P / 2, NB = P / 32, OUT = 2 * P, OB = OUT / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_op[];
    float* Wol = smem_op;                    // fp16 hi | lo planes (prd_common.h: stage_weight_h2), x 16
    float* Wgol = Wol + P * P;
    float* Wpl = Wgol + P * P;
    float* Wgl = Wpl + OUT * P;
    float* bol = Wgl + OUT * P;              // [P] CLL
    float* bgol = bol + P;                   // [P] CLL
    float* bpl = bgol + P;                   // [2P] CLL, x 16 (rides in the accumulators)
    float* bgl = bpl + OUT;                  // [2P] CLL, x -log2(e) x 16
    stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wol), wo, P, P, threadIdx.x, NW * 64, H2_WSCALE);
    stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wgol), wog, P, P, threadIdx.x, NW * 64, H2_WSCALE);
    stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wpl), wp, OUT, P, threadIdx.x, NW * 64, H2_WSCALE);
    stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wgl), wg, OUT, P, threadIdx.x, NW * 64, NEG_LOG2E * H2_WSCALE);
    stage_vec_cll(bol, bo, P, threadIdx.x, NW * 64);
    stage_vec_cll(bgol, bog, P, threadIdx.x, NW * 64);
    stage_vec_cll(bpl, bp, OUT, threadIdx.x, NW * 64, H2_WSCALE);
    stage_vec_cll(bgl, bg, OUT, threadIdx.x, NW * 64, NEG_LOG2E * H2_WSCALE);
    __syncthreads();
    const int lane = threadIdx.x & 63, r_w = lane & 31, hi_w = lane >> 5;
    const int nvb = ldn / 32;
    const long ntask = (long)b * N * nvb;
    const unsigned cbytes = (unsigned)N * (unsigned)ldn * 4u;              // channel stride of Ot and AB
    WaveTasks tasks(nullptr, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        // lane coordinates opaque per task (as in pair_tail_h2_kernel): lane-dependent addresses hoisted out of the task loop were
        // parked in scratch at the 168-register limit of 12 waves (12 B / lane)
        int r = r_w, hi = hi_w;
        asm volatile("" : "+v"(r), "+v"(hi));
        const unsigned lane_off = (unsigned)(4 * hi) * cbytes + (unsigned)r * 4u;
        const int ti = (int)task;
        const int vb = ti % nvb, bu = ti / nvb;            // bu = bb * N + u
        const int bb = bu / N, u = bu - bb * N;
        const int v = vb * 32 + r;
        const bool valid = v < N;
        const int vv = valid ? v : 0;
        float x[KH], xr[KH];
        {
            const prd_rsrc ro = make_rsrc(Ot + (((long)bb * P) * N + u) * ldn + vb * 32);
#pragma unroll
            for (int s = 0; s < KH; ++s) x[s] = buf_load(ro, valid ? lane_off : BUF_OOB, (unsigned)(8 * (s >> 2) + (s & 3)) * cbytes);
            const prd_rsrc rp = make_rsrc(pair + (long)bb * N * N * P);
            load_row_cll_buf<P>(rp, valid ? (((unsigned)vv * N + u) * P + 4 * hi) * 4u : BUF_OOB, xr);
        }
        const float mu = mask[bu], mv = mask[bb * N + vv];
        // ---- output stage of the outgoing module ----
        float gate[KH];
        {
            float xn[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) xn[s] = xr[s];
            ln_cll<KH>(xn);
            f32x16 ag[NB];
            zero_acc(ag);
            u32x4 xs[2][P / 16];
            split2h_cll<KH>(xn, xs);
            rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wgol), P, 0, xs, ag, r, hi);
#pragma unroll
            for (int s = 0; s < KH; ++s) gate[s] = sigmoid_fast(ag[s >> 4][s & 15] * H2_INV_WSCALE + bgol[hi * KH + s]);
        }
        ln_cll<KH>(x);
        {
            f32x16 ao[NB];
            zero_acc(ao);
            u32x4 xs[2][P / 16];
            split2h_cll<KH>(x, xs);
            rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wol), P, 0, xs, ao, r, hi);
#pragma unroll
            for (int s = 0; s < KH; ++s) x[s] = xr[s] + gate[s] * (ao[s >> 4][s & 15] * H2_INV_WSCALE + bol[hi * KH + s]);
        }
        store_row_cll<P>(pair + (((long)bb * N + vv) * N + u) * P, hi, valid, x);          // the updated pair row
        // ---- projection stage of the incoming module on the row still in registers ----
        ln_cll<KH>(x);
        u32x4 xs[2][P / 16];
        split2h_cll<KH>(x, xs);
        const float m2 = valid ? mu * mv : 0.f;
        const bool plain = __all(m2 == 1.0f);
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            f32x16 ap[1], ag[1];
            bias_acc(ap, bpl + hi * P + 16 * ob);
            bias_acc(ag, bgl + hi * P + 16 * ob);
            rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wpl), OUT, ob * 32, xs, ap, r, hi);
            rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wgl), OUT, ob * 32, xs, ag, r, hi);
            // output channel of register q: 32 ob + (q & 3) + 8 (q >> 2) + 4 hi; the hi part sits in lane_off
            const prd_rsrc cb = make_rsrc(AB + ((((long)bb * 2 * P) + 32 * ob) * N + u) * ldn + vb * 32);
            if (plain) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float val = (ap[0][q] * H2_INV_WSCALE) * gate_from_scaled(ag[0][q] * H2_INV_WSCALE);
                    buf_store(val, cb, lane_off, (unsigned)((q & 3) + 8 * (q >> 2)) * cbytes);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    buf_store(m2 * ((ap[0][q] * H2_INV_WSCALE) * gate_from_scaled(ag[0][q] * H2_INV_WSCALE)), cb, lane_off,
                              (unsigned)((q & 3) + 8 * (q >> 2)) * cbytes);
            }
        }
    }
}

// Triangle-multiplication contraction (reference modules.py:272, "ikd,jkd->ijd" / "kid,kjd->ijd" after the
// operand transposition done by tri_mul_proj): nch = b*P independent K-contiguous GEMMs
//   O[ch][i][j] = sum_k A[ch][i][k] * B[ch][j][k],   A/B/O rows of pitch ldn (zero padded to a multiple of 32).
// 64x64 output tile per workgroup (exact for N = 320), 2x2 waves of 32x32 on v_mfma_f32_32x32x2_f32, K in
// chunks of 32 through DOUBLE-BUFFERED LDS: the global loads of chunk c+1 are in flight while chunk c is
// multiplied, one barrier per chunk.  Workgroups of one channel are placed on one XCD (they share A/B in L2).
__global__ __launch_bounds__(256) void tri_mul_contract_kernel(float* __restrict__ O, const float* __restrict__ AB,
                                                               int N, int ldn, int P, int nbatch, int tiles) {
    // Measured alternatives at N = 320 (b = 1): this form (32-wide K chunks, two LDS buffers, 4 workgroups / CU, global
    // loads two chunks ahead) 51 us; loads one chunk ahead 54 us; one LDS buffer + register prefetch 83 us; 64-wide chunks
    // (2 workgroups / CU) 119 us; generic prd_gemm 55 us.  In-kernel stamps: the first 1024 workgroups keep the matrix pipe
    // ~100 % busy for 17 us; the launch as a whole is cold-start + a 1.56-round tail (1600 tiles on 1024 slots).
    constexpr int KCH = 32, LDP = KCH + 4;
    __shared__ __attribute__((aligned(16))) float As[2][64 * LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    // Persistent workgroups (4 per CU): virtual block vb = blockIdx.x, blockIdx.x + gridDim.x, ...  One workgroup per tile
    // leaves the dispatcher to refill whichever CUs finish first (1600 tiles on 1024 slots at N = 320: a second full round
    // on a subset of the CUs, 47 us); consecutive workgroup ids sit on different CUs, so the strided loop gives every CU
    // 6 or 7 tiles.
    const int nch = nbatch * P;
    const int t2 = tiles * tiles;
    for (int vblk = blockIdx.x; vblk < nch * t2; vblk += gridDim.x) {
    // (channel, tile) of this virtual block: the tiles of one channel stay on one XCD (they share A / B in L2)
    int ch, tile;
    if ((nch & 7) == 0 && (gridDim.x & 7) == 0) {
        const int xcd = vblk & 7, k = vblk >> 3;
        ch = xcd + 8 * (k / t2);
        tile = k % t2;
    } else {
        ch = vblk / t2;
        tile = vblk % t2;
    }
    const int bb = ch / P, d = ch - bb * P;
    const int m0 = (tile / tiles) * 64, n0 = (tile % tiles) * 64;
    const float* __restrict__ A = AB + ((size_t)bb * 2 * P + d) * N * ldn;
    const float* __restrict__ B = AB + ((size_t)bb * 2 * P + P + d) * N * ldn;
    // staging assignment: thread -> (row = tid>>3 (+32), 16-byte group f = tid&7); explicit scalars, no arrays
    const int srow = tid >> 3, sf = tid & 7;
    const bool a0 = (m0 + srow) < N, a1 = (m0 + srow + 32) < N, b0 = (n0 + srow) < N, b1 = (n0 + srow + 32) < N;
    // buffer addressing: descriptor base = the tile's first operand row (uniform), lane offset = (row, 16-byte group) fixed
    // for the whole kernel (BUF_OOB for rows past the edge: they load zeros), per-chunk offset in an SGPR -- the chunk loop
    // carries no VALU address arithmetic and no edge selects (fp32 MFMA and VALU share the SIMD's issue time)
    const prd_rsrc ra = make_rsrc(A + (size_t)m0 * ldn), rb = make_rsrc(B + (size_t)n0 * ldn);
    const unsigned oa0 = a0 ? ((unsigned)srow * ldn + 4 * sf) * 4u : BUF_OOB, oa1 = a1 ? ((unsigned)(srow + 32) * ldn + 4 * sf) * 4u : BUF_OOB;
    const unsigned ob0 = b0 ? ((unsigned)srow * ldn + 4 * sf) * 4u : BUF_OOB, ob1 = b1 ? ((unsigned)(srow + 32) * ldn + 4 * sf) * 4u : BUF_OOB;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int nchunk = ldn / KCH;                        // ldn is a multiple of 32; columns >= N hold zeros
#define PRD_TMC_LD1(RS, OFF, KO) [&] { const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(RS, OFF, (KO) * 4, 0);      \
        return make_float4(__uint_as_float(v_[0]), __uint_as_float(v_[1]), __uint_as_float(v_[2]), __uint_as_float(v_[3])); }()
#define PRD_TMC_LOAD(R, KO)                                                       \
    R##a0 = PRD_TMC_LD1(ra, oa0, KO);                                             \
    R##a1 = PRD_TMC_LD1(ra, oa1, KO);                                             \
    R##b0 = PRD_TMC_LD1(rb, ob0, KO);                                             \
    R##b1 = PRD_TMC_LD1(rb, ob1, KO);
#define PRD_TMC_STAGE(R, BUF)                                                     \
    *reinterpret_cast<float4*>(&As[BUF][srow * LDP + 4 * sf]) = R##a0;            \
    *reinterpret_cast<float4*>(&As[BUF][(srow + 32) * LDP + 4 * sf]) = R##a1;     \
    *reinterpret_cast<float4*>(&Bs[BUF][srow * LDP + 4 * sf]) = R##b0;            \
    *reinterpret_cast<float4*>(&Bs[BUF][(srow + 32) * LDP + 4 * sf]) = R##b1;
    // One chunk: the global loads of chunk c+2 are issued into register set N (measured: one chunk of distance leaves the
    // wave waiting ~2000 cycles per chunk for them), chunk c is multiplied out of LDS buffer CUR, then register set R
    // (chunk c+1, loaded during chunk c-1) goes to the other buffer.  Loads are UNCONDITIONAL with a clamped chunk index
    // (loads under an `if` make hipcc emit `s_waitcnt vmcnt(0)` in front of the MFMAs).
#define PRD_TMC_CHUNK(C, CUR, R, N)                                               \
    {                                                                             \
        const int c2 = (C) + 2 < nchunk ? (C) + 2 : nchunk - 1;                   \
        PRD_TMC_LOAD(N, c2 * KCH)                                                 \
        const float* as = &As[CUR][(wm0 + r) * LDP + hi * 16];                    \
        const float* bs = &Bs[CUR][(wn0 + r) * LDP + hi * 16];                    \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                           \
            const float4 a = *reinterpret_cast<const float4*>(as + 4 * t);        \
            const float4 bq = *reinterpret_cast<const float4*>(bs + 4 * t);       \
            acc = mfma32(a.x, bq.x, acc);                                         \
            acc = mfma32(a.y, bq.y, acc);                                         \
            acc = mfma32(a.z, bq.z, acc);                                         \
            acc = mfma32(a.w, bq.w, acc);                                         \
        }                                                                         \
        if ((C) + 1 < nchunk) { PRD_TMC_STAGE(R, (CUR) ^ 1) }                     \
        __syncthreads();                                                          \
    }
    float4 ua0, ua1, ub0, ub1, va0, va1, vb0, vb1;
    PRD_TMC_LOAD(u, 0)
    PRD_TMC_STAGE(u, 0)
    {
        const int c1 = 1 < nchunk ? 1 : 0;
        PRD_TMC_LOAD(u, c1 * KCH)                       // chunk 1 -> set u (staged at the end of chunk 0)
    }
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {
        PRD_TMC_CHUNK(c, 0, u, v)
        if (c + 1 < nchunk) PRD_TMC_CHUNK(c + 1, 1, v, u)
    }
#undef PRD_TMC_LOAD
#undef PRD_TMC_LD1
#undef PRD_TMC_STAGE
#undef PRD_TMC_CHUNK
    float* __restrict__ Oc = O + (size_t)ch * N * ldn;
    const int n = n0 + wn0 + r;
    if (n < N) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm0 + drow32(q, hi);
            if (m < N) Oc[(size_t)m * ldn + n] = acc[q];
        }
    }
    }   // virtual blocks
}

// The same contraction on the fp16 matrix pipe (gemm mode 1).  The fp32 operands are split into fp16 hi + lo (RTZ: 22 bits, see
// prd_common.h; no scaling: the operands are gated projections of LayerNorm-ed rows, O(1), and what falls below the normal range
// of the lo part is < 3e-8 absolute) while they are staged into LDS (each element is split N / 160 times in total); the three
// products hi*hi, hi*lo, lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation run at 16/3 of the fp32 matrix rate.  At that
// rate a 64x64 tile is bound by the CU's 64 B/clk memory path, so the tile is 160 x 160: 5 x 5 sub-tiles of 32 x 32 dealt to
// 8 waves (N = 320: exactly 2 x 2 tiles per channel, one tile per CU), K in chunks of 32 through double-buffered LDS.
// LDS rows are 64 B (32 fp16) per plane without padding; the 16-byte slot j of row r sits at j ^ ((r >> 2) & 3), so the
// sixteen lanes of a ds_read_b128 group (rows distinct mod 16, same logical slot) hit sixteen different 4-bank groups.
constexpr int TMS_T = 160, TMS_PLANE = TMS_T * 64, TMS_OPER = 2 * TMS_PLANE;       // bytes
// waves per workgroup of the split contraction: 8 by default; 16 (PRD_TMS_NW=16) makes the kernel itself 1 us faster (18.8 vs
// 19.8 us) but the whole step 15 us slower in the same run (1.932 vs 1.918 ms, twice); 12 waves: 22.5 vs 22.7 us -- A/B switch
// PRD_TUNE_TMS_NW in the upper bits of `arith`
template <int NWV, int DEPTH = 2>                   // 8 or 16 waves: 25 sub-tiles dealt round-robin, 4 or 2 accumulators per wave; DEPTH: chunks of operands in flight
__global__ __launch_bounds__(NWV * 64) void tri_mul_contract_split_kernel(float* __restrict__ O, const float* __restrict__ AB,
                                                                          int N, int ldn, int P, int nbatch, int tiles, int swap) {
    constexpr int NT = NWV * 64, NPT = (2560 + NT - 1) / NT, NSUB = (25 + NWV - 1) / NWV;
    extern __shared__ __attribute__((aligned(16))) unsigned char tms[];          // [2 buffers][A | B][2 planes][160 rows][64 B]
    PhaseTimer pt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int nch = nbatch * P;
    const int t2 = tiles * tiles;
    // staging: 2 x 160 rows x 8 pieces of 4 floats per chunk = 2560 pieces, NPT per thread.  Piece p = tid + NT i; pieces
    // below 1280 are A rows, the others B rows: 1280 is a multiple of 64, so the operand (and whether the piece exists) is
    // wave-uniform.
    unsigned srow[NPT], sdst[NPT], ssrc[NPT];
    bool sisb[NPT], sok[NPT];
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);     // scalar: operand / existence of a piece are decided per wave
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int pw = wbase + NT * i;
        sok[i] = pw < 2560;
        const int oper = (sok[i] && pw >= 1280) ? 1 : 0;
        const int pc = sok[i] ? tid + NT * i : 0;
        const int q = pc - 1280 * oper;
        const int row = q >> 3, f = q & 7;
        sisb[i] = oper == 1;
        srow[i] = row;
        sdst[i] = oper * TMS_OPER + row * 64 + (((f >> 1) ^ ((row >> 2) & 3)) << 4) + (f & 1) * 8;
        ssrc[i] = ((unsigned)row * ldn + 4 * f) * 4u;
    }
    const unsigned swz = (unsigned)((r >> 2) & 3);
    for (int vblk = blockIdx.x; vblk < nch * t2; vblk += gridDim.x) {
        int ch, tile;
        if ((nch & 7) == 0 && (gridDim.x & 7) == 0) {     // the tiles of one channel stay on one XCD (they share A / B in L2)
            const int xcd = vblk & 7, k = vblk >> 3;
            ch = xcd + 8 * (k / t2);
            tile = k % t2;
        } else {
            ch = vblk / t2;
            tile = vblk % t2;
        }
        const int bb = ch / P, d = ch - bb * P;
        const int m0 = (tile / tiles) * TMS_T, n0 = (tile % tiles) * TMS_T;
        // swap: the operands change roles, i.e. the output is the TRANSPOSE O^T[j][i] (what the column-wise tasks of
        // tri_mul_out_proj_kernel read contiguously)
        const float* __restrict__ A = AB + ((size_t)bb * 2 * P + (swap ? P : 0) + d) * N * ldn;
        const float* __restrict__ B = AB + ((size_t)bb * 2 * P + (swap ? 0 : P) + d) * N * ldn;
        const prd_rsrc ra = make_rsrc(A + (size_t)m0 * ldn), rb = make_rsrc(B + (size_t)n0 * ldn);
        unsigned off[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            off[i] = (sok[i] && (sisb[i] ? n0 : m0) + (int)srow[i] < N) ? ssrc[i] : BUF_OOB;      // rows past the edge load zeros
        // sub-tiles of this wave: s = wave + NWV k (k < NSUB) of the 5 x 5 grid, skipped when outside the matrix
        int si[NSUB], sj[NSUB];
        bool sv[NSUB];
#pragma unroll
        for (int k = 0; k < NSUB; ++k) {
            const int s_ = wave + NWV * k;
            si[k] = s_ / 5;
            sj[k] = s_ - 5 * si[k];
            sv[k] = s_ < 25 && m0 + 32 * si[k] < N && n0 + 32 * sj[k] < N;
        }
        f32x16 acc[NSUB];
        zero_acc(acc);
        const int nchunk = ldn / 32;
        u32x4 u[NPT], v[NPT];
#define PRD_TMS_LOAD(R, C)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NPT; ++i)                                                                 \
        R[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sisb[i] ? rb : ra, off[i], (C) * 128, 0));
        // four fp32 values -> two planes of four fp16 (8 bytes each)
#define PRD_TMS_STAGE(R, BUF)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NPT; ++i) {                                                               \
        if (!sok[i]) continue;                                                                                      \
        unsigned h0, l0, h1, l1;                                                                                    \
        split2h(__uint_as_float(R[i][0]), __uint_as_float(R[i][1]), h0, l0);                                        \
        split2h(__uint_as_float(R[i][2]), __uint_as_float(R[i][3]), h1, l1);                                        \
        unsigned char* dst = tms + (BUF) * 2 * TMS_OPER + sdst[i];                                                  \
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};                                                             \
        *reinterpret_cast<u32x2*>(dst + TMS_PLANE) = u32x2{l0, l1};                                                 \
    }
#define PRD_TMS_CHUNK(C, CUR, R, NX)                                                                                \
    {                                                                                                               \
        const int c2 = (C) + 2 < nchunk ? (C) + 2 : nchunk - 1;                                                     \
        PRD_TMS_LOAD(NX, c2)                                                                                        \
        const unsigned char* base = tms + (CUR) * 2 * TMS_OPER;                                                     \
        _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                          \
            const unsigned col = (((unsigned)(2 * st + hi)) ^ swz) << 4;                                            \
            _Pragma("unroll") for (int k = 0; k < NSUB; ++k) {                                                      \
                if (sv[k]) {                                                                                        \
                    const unsigned char* ap = base + (32 * si[k] + r) * 64 + col;                                   \
                    const unsigned char* bp = base + TMS_OPER + (32 * sj[k] + r) * 64 + col;                        \
                    u32x4 a[2], bq[2];                                                                              \
                    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                              \
                        a[pl] = *reinterpret_cast<const u32x4*>(ap + pl * TMS_PLANE);                               \
                        bq[pl] = *reinterpret_cast<const u32x4*>(bp + pl * TMS_PLANE);                              \
                    }                                                                                               \
                    const int pa[3] = {0, 0, 1}, pb[3] = {0, 1, 0};                                                 \
                    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                   \
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[pa[t]]),      \
                                                                        __builtin_bit_cast(f16x8_t, bq[pb[t]]), acc[k], 0, 0, 0); \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        pt.mark(0);                                     /* 0: load issue + LDS reads + MFMAs */                     \
        if ((C) + 1 < nchunk) { PRD_TMS_STAGE(R, (CUR) ^ 1) }                                                       \
        pt.mark(1);                                     /* 1: wait for the next chunk + split + LDS writes */       \
        __syncthreads();                                                                                            \
        pt.mark(2);                                     /* 2: barrier */                                            \
    }
        pt.mark(6);                                     // 6: tile decode (and, for the first tile, kernel entry)
        if constexpr (DEPTH == 3) {
            // three register sets: while chunk c is multiplied from LDS, chunk c + 1 is in registers (staged at the end of c), c + 2
            // and c + 3 are in flight.  What a CU ingests is set by the bytes it has in flight (gemm_h2): 123 KB here
            // against 82 KB with two sets
            u32x4 w[NPT];
#define PRD_TMS_CHUNK3(C, R, NX)                                                                                      \
    {                                                                                                               \
        const int c3 = (C) + 3 < nchunk ? (C) + 3 : nchunk - 1;                                                     \
        PRD_TMS_LOAD(NX, c3)                                                                                        \
        const int cur_ = (C) & 1;                                                                                   \
        const unsigned char* base = tms + cur_ * 2 * TMS_OPER;                                                      \
        _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                          \
            const unsigned col = (((unsigned)(2 * st + hi)) ^ swz) << 4;                                            \
            _Pragma("unroll") for (int k = 0; k < NSUB; ++k) {                                                      \
                if (sv[k]) {                                                                                        \
                    const unsigned char* ap = base + (32 * si[k] + r) * 64 + col;                                   \
                    const unsigned char* bp = base + TMS_OPER + (32 * sj[k] + r) * 64 + col;                        \
                    u32x4 a[2], bq[2];                                                                              \
                    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                              \
                        a[pl] = *reinterpret_cast<const u32x4*>(ap + pl * TMS_PLANE);                               \
                        bq[pl] = *reinterpret_cast<const u32x4*>(bp + pl * TMS_PLANE);                              \
                    }                                                                                               \
                    const int pa[3] = {0, 0, 1}, pb[3] = {0, 1, 0};                                                 \
                    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                   \
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[pa[t]]),      \
                                                                        __builtin_bit_cast(f16x8_t, bq[pb[t]]), acc[k], 0, 0, 0); \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        pt.mark(0);                                                                                                 \
        if ((C) + 1 < nchunk) { PRD_TMS_STAGE(R, cur_ ^ 1) }                                                        \
        pt.mark(1);                                                                                                 \
        __syncthreads();                                                                                            \
        pt.mark(2);                                                                                                 \
    }
            PRD_TMS_LOAD(u, 0)
            PRD_TMS_STAGE(u, 0)
            PRD_TMS_LOAD(u, (1 < nchunk ? 1 : 0))
            PRD_TMS_LOAD(v, (2 < nchunk ? 2 : nchunk - 1))
            __syncthreads();
            pt.mark(3);
            for (int c = 0; c < nchunk; c += 3) {
                PRD_TMS_CHUNK3(c, u, w)
                if (c + 1 < nchunk) PRD_TMS_CHUNK3(c + 1, v, u)
                if (c + 2 < nchunk) PRD_TMS_CHUNK3(c + 2, w, v)
            }
#undef PRD_TMS_CHUNK3
        } else {
        PRD_TMS_LOAD(u, 0)
        PRD_TMS_STAGE(u, 0)
        {
            const int c1 = 1 < nchunk ? 1 : 0;
            PRD_TMS_LOAD(u, c1)
        }
        __syncthreads();
        pt.mark(3);                                     // 3: first chunk: exposed load latency + staging
        for (int c = 0; c < nchunk; c += 2) {
            PRD_TMS_CHUNK(c, 0, u, v)
            if (c + 1 < nchunk) PRD_TMS_CHUNK(c + 1, 1, v, u)
        }
        }
#undef PRD_TMS_LOAD
#undef PRD_TMS_STAGE
#undef PRD_TMS_CHUNK
        float* __restrict__ Oc = O + (size_t)ch * N * ldn;
#pragma unroll
        for (int k = 0; k < NSUB; ++k) {
            if (!sv[k]) continue;
            const int n = n0 + 32 * sj[k] + r;
            if (n < N) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = m0 + 32 * si[k] + drow32(q, hi);
                    if (m < N) Oc[(size_t)m * ldn + n] = acc[k][q];
                }
            }
        }
        pt.mark(4);                                     // 4: output stores
    }   // virtual blocks
    pt.flush(3);
}

template <int P, int NW, bool B3>
__global__ __launch_bounds__(NW * 64) void tri_mul_out_kernel(int* queue, float* out, const float* pair, const float* __restrict__ O,
                                                              const float* __restrict__ wo, const float* __restrict__ bo,
                                                              const float* __restrict__ wog, const float* __restrict__ bog,
                                                              int b, int N, int ldn, int residual) {
    constexpr int KH = P / 2, NB = P / 32;
    constexpr int WSZ = B3 ? P * P : P * (P + 4);        // B3: fp16 x 2 row GEMMs (prd_common.h: rowgemm_h2), weights x 16
    __shared__ __attribute__((aligned(16))) float Wol[WSZ];
    __shared__ __attribute__((aligned(16))) float Wgl[WSZ];
    __shared__ __attribute__((aligned(16))) float bol[P];
    __shared__ __attribute__((aligned(16))) float bgl[P];
    constexpr float ASC = B3 ? H2_INV_WSCALE : 1.0f;     // accumulator scale of the split form
    PhaseTimer pt;
    if (B3) {
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wol), wo, P, P, threadIdx.x, NW * 64, H2_WSCALE);
        stage_weight_h2<P>(reinterpret_cast<u32x4*>(Wgl), wog, P, P, threadIdx.x, NW * 64, H2_WSCALE);
    } else {
        stage_weight_cll<P>(Wol, wo, P, P, threadIdx.x, NW * 64);
        stage_weight_cll<P>(Wgl, wog, P, P, threadIdx.x, NW * 64);
    }
    stage_vec_cll(bol, bo, P, threadIdx.x, NW * 64);
    stage_vec_cll(bgl, bog, P, threadIdx.x, NW * 64);
    __syncthreads();
    pt.mark(6);                                         // 6: prologue (weight staging, barrier)
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
    const int nvb = (N + 31) / 32;
    const long ntask = (long)b * N * nvb;
    // What is left after the whole rounds of 4 * gridDim.x tasks (one per SIMD) is computed by the four SIMDs of a
    // workgroup together, like in block_tail: waves 0 / 1 the gate for output channels 0-31 / 32-63, waves 2 / 3 the
    // projection of LN(O) for the same halves (32 MFMAs each instead of 128 in one wave); the halves meet in LDS.
    const long slots = (long)gridDim.x * 4;
    const long left = ntask % slots;
    // (fp32 form only: the split form is bound by operand latency, not by the matrix pipe -- co-resident waves overlap, and a
    // partial extra round on the second wave of a SIMD costs less than the cooperative pass)
    const bool coop = !B3 && queue == nullptr && NB == 2 && NW >= 4 && left > 0 && left <= 2 * (long)gridDim.x;
    const long nwhole = coop ? ntask - left : ntask;
    WaveTasks tasks(queue, nwhole, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const int vb = (int)(task % nvb);
        const long bi = task / nvb;
        const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
        const int j = vb * 32 + r;
        const bool valid = j < N;
        const int jj = valid ? j : 0;
        const long off = (bi * N + jj) * P;
        // Both operands of the task are requested up front: the contraction output of this (i,j) for the lane's channels
        // (coalesced over j per channel; lanes past N read column 0 and are never stored) and the pair row -- one exposed
        // memory latency per task instead of three (row, O, row again for the residual).
        // Buffer addressing: one descriptor per operand and task, the lane part of the address computed once, the channel
        // stride of O as a scalar offset -- 32 scattered 64-bit address computations per task were a third of a wave's cycles.
        float x[KH], xr[KH];
        {
            const prd_rsrc ro = make_rsrc(O + (((long)bb * P) * N + i) * ldn + vb * 32);
            const unsigned cbytes = (unsigned)N * (unsigned)ldn * 4u;              // channel stride of the contraction output
            const unsigned lo = valid ? (unsigned)r * 4u + (unsigned)(4 * hi) * cbytes : BUF_OOB;
#pragma unroll
            for (int s = 0; s < KH; ++s) x[s] = buf_load(ro, lo, (unsigned)(8 * (s >> 2) + (s & 3)) * cbytes);
            const prd_rsrc rp = make_rsrc(pair + (bi * N + vb * 32) * P);
            load_row_cll_buf<P>(rp, valid ? ((unsigned)r * P + 4 * hi) * 4u : BUF_OOB, xr);
        }
        pt.mark(0);                                     // 0: task decode, load issue
        float gate[KH];
        {
            float xn[KH];
#pragma unroll
            for (int s = 0; s < KH; ++s) xn[s] = xr[s];
            ln_cll<KH>(xn);
            pt.mark(1);                                 // 1: wait for the pair row, LayerNorm
            f32x16 ag[NB];
            zero_acc(ag);
            if (B3) {
                u32x4 xs[2][P / 16];
                split2h_cll<KH>(xn, xs);
                rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wgl), P, 0, xs, ag, r, hi);
            } else {
                rowgemm<P, NB>(Wgl, xn, ag, r, hi);
            }
#pragma unroll
            for (int s = 0; s < KH; ++s) gate[s] = sigmoid_fast(ag[s >> 4][s & 15] * ASC + bgl[hi * KH + s]);
        }
        pt.mark(2);                                     // 2: gate GEMM + sigmoid
        ln_cll<KH>(x);
        pt.mark(3);                                     // 3: wait for O, LayerNorm
        f32x16 ao[NB];
        zero_acc(ao);
        if (B3) {
            u32x4 xs[2][P / 16];
            split2h_cll<KH>(x, xs);
            rowgemm_h2<P, NB>(reinterpret_cast<const u32x4*>(Wol), P, 0, xs, ao, r, hi);
        } else {
            rowgemm<P, NB>(Wol, x, ao, r, hi);
        }
        pt.mark(4);                                     // 4: projection GEMM
#pragma unroll
        for (int s = 0; s < KH; ++s) x[s] = (residual ? xr[s] : 0.f) + gate[s] * (ao[s >> 4][s & 15] * ASC + bol[hi * KH + s]);
        store_row_cll<P>(out + off, hi, valid, x);
        pt.mark(5);                                     // 5: epilogue + store issue
    }
    pt.mark(7);                                         // 7: leaving the task loop (queue exhausted)
    pt.flush(2);
    if (coop) {
        __shared__ float part[2][64][17];                     // LN(O) projections of waves 2 / 3, per lane 16 values
        for (long task = nwhole + blockIdx.x; task < ntask; task += gridDim.x) {     // uniform over the workgroup
            const int vb = (int)(task % nvb);
            const long bi = task / nvb;
            const int bb = (int)(bi / N), i = (int)(bi - (long)bb * N);
            const int j = vb * 32 + r;
            const bool valid = j < N;
            const int jj = valid ? j : 0;
            const long off = (bi * N + jj) * P;
            const int nb = wave & 1;                           // output channels [32 nb, 32 nb + 32)
            f32x16 acc[1];
            zero_acc(acc);
            float x[KH];
            if (wave < 2) {
                load_row_cll<P>(pair + off, hi, valid, x);
                ln_cll<KH>(x);
                if (B3) {
                    u32x4 xs[2][P / 16];
                    split2h_cll<KH>(x, xs);
                    rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wgl), P, nb * 32, xs, acc, r, hi);
                } else {
                    rowgemm<P, 1>(Wgl + nb * 32 * (P + 4), x, acc, r, hi);
                }
            } else if (wave < 4) {
#pragma unroll
                for (int s = 0; s < KH; ++s)
                    x[s] = valid ? O[(((long)bb * P + cll_ch(s, hi)) * N + i) * ldn + jj] : 0.f;
                ln_cll<KH>(x);
                if (B3) {
                    u32x4 xs[2][P / 16];
                    split2h_cll<KH>(x, xs);
                    rowgemm_h2<P, 1>(reinterpret_cast<const u32x4*>(Wol), P, nb * 32, xs, acc, r, hi);
                } else {
                    rowgemm<P, 1>(Wol + nb * 32 * (P + 4), x, acc, r, hi);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) part[nb][lane][q] = acc[0][q] * ASC;
            }
            __syncthreads();
            if (wave < 2) {
                // CLL elements 16 nb .. 16 nb + 15 = channels 32 nb + 8 g + 4 hi + e: four 16-byte groups of the row
                float* orow = out + off + 32 * nb + 4 * hi;
                const float* prow = pair + off + 32 * nb + 4 * hi;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid && residual) pv = *reinterpret_cast<const float4*>(prow + 8 * gq);
                    float o4[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = 4 * gq + e, sidx = 16 * nb + q;
                        const float gt = sigmoid_fast(acc[0][q] * ASC + bgl[hi * KH + sidx]);
                        o4[e] = o4[e] + gt * (part[nb][lane][q] + bol[hi * KH + sidx]);
                    }
                    if (valid) *reinterpret_cast<float4*>(orow + 8 * gq) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                }
            }
            __syncthreads();                                   // part[] consumed before the next cooperative task
        }
    }
}

// ---------------------------------------------------------------------------------------------
// triangle attention core.  HC = H*c = 64, c = 16.  One workgroup per (b, row u, head h), NW waves.
//   phase 1: every wave projects K_h / V_h for 32-position blocks of the row into LDS
//   phase 2: every wave projects q_h / gate_h for its 32 queries, then streams the keys in blocks of
//            64: S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_16x16x4_f32 for BOTH 16-query tiles at once
//            (two independent dependency chains), online softmax in the exp2 domain (log2(e)/sqrt(c)
//            folded into q), one cross-lane max (permlane swaps) per 64 keys.
// Key padding / masking is an fma with per-key (mul, add): valid (1, 0); masked (0, -2^15 log2 e),
// i.e. masked_fill(-2**15) of modules.py:220; beyond N (0, -inf) = excluded.
// ---------------------------------------------------------------------------------------------
constexpr int KP = 20;          // LDS pitch (floats) of the [*, 16] K / Q / G tiles
constexpr float LOG2E = 1.4426950408889634f;

// The key loop of one wave for NTQ (1 or 2) 16-query tiles: S^T = K Q^T, online softmax in the exp2
// domain over blocks of 16*JT keys, O^T += V^T P^T.  Returns O^T[c = 4*g4 + e][q = ql] and l per tile.
//
// fp32 MFMA and VALU instructions share the SIMD's issue time on gfx950 (tools/ubench/coissue_bench.hip: the cycles
// add, for any number of waves), so the softmax arithmetic is kept as short as it gets: 64 keys per running-max update,
// v_max3 without canonicalisation, packed fp32 subtract / add (two logits per instruction); what is left is the one
// v_exp_f32 per logit.
PRD_DEV void ta_prio(int rem, int npad) {
    // Priority = fraction of the wave's own key loop still to do.  The waves of a SIMD are arbitrated strictly
    // oldest-first: without this the youngest wave is starved until the others are done and then runs alone,
    // latency-bound (measured at N = 320: phase 2 of a row 42.4k -> 39.4k cycles, all waves end together).
    if (4 * rem > 3 * npad) __builtin_amdgcn_s_setprio(3);
    else if (2 * rem > npad) __builtin_amdgcn_s_setprio(2);
    else if (4 * rem > npad) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// One 64-key block (JT = 4 tiles of 16 keys) of the key loop.
//   ONLINE: running maximum, rescale of o / l (the classic online softmax).
//   !ONLINE: the reference maximum m_run stays frozen; the logits come out of the MFMA already relative to it (the
//            accumulator is preloaded with -m_run), so a block costs one v_exp_f32 per logit and the sum -- no max, no
//            cross-lane reduction, no subtract, no rescale (-45 % VALU instructions, which on gfx950 are matrix-pipe time).
template <int NTQ, bool MASKED, bool ONLINE>
PRD_DEV void ta_block(const float* __restrict__ Kl, const float* __restrict__ Vt, const float* __restrict__ kadd,
                      const float4 (&qf)[NTQ], int npad, int key0, int ql, int g4,
                      float (&m_run)[NTQ], float (&l_run)[NTQ], f32x4 (&o)[NTQ]) {
    constexpr int JT = 4;                      // 16-key tiles per block (64 keys; npad is a multiple of 64)
    ta_prio(npad - key0, npad);
    float4 kf[JT], ma[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        kf[j] = *reinterpret_cast<const float4*>(Kl + (key0 + 16 * j + ql) * KP + 4 * g4);
        if (MASKED) ma[j] = *reinterpret_cast<const float4*>(kadd + key0 + 16 * j + 4 * g4);
    }
    f32x4 s[NTQ][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const float c0 = ONLINE ? 0.f : -m_run[t];
            f32x4 z4 = {c0, c0, c0, c0};
            z4 = mfma16(kf[j].x, qf[t].x, z4);          // S^T[key = key0+16j+4*g4+e][q]  (- m_run when !ONLINE)
            z4 = mfma16(kf[j].y, qf[t].y, z4);
            z4 = mfma16(kf[j].z, qf[t].z, z4);
            z4 = mfma16(kf[j].w, qf[t].w, z4);
            s[t][j] = z4;
        }
    __builtin_amdgcn_sched_barrier(0);                  // V^T is fetched behind the QK^T MFMAs, not before them
    float4 vf[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
        vf[j] = *reinterpret_cast<const float4*>(Vt + ql * (npad + 4) + key0 + 16 * j + 4 * g4);
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        if (MASKED) {
            const float mr = ONLINE ? 0.f : m_run[t];   // override values are absolute logits
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                s[t][j][0] = (ma[j].x == 0.f) ? s[t][j][0] : ma[j].x - mr;
                s[t][j][1] = (ma[j].y == 0.f) ? s[t][j][1] : ma[j].y - mr;
                s[t][j][2] = (ma[j].z == 0.f) ? s[t][j][2] : ma[j].z - mr;
                s[t][j][3] = (ma[j].w == 0.f) ? s[t][j][3] : ma[j].w - mr;
            }
        }
        f32x2 ps = {0.f, 0.f};
        if (ONLINE) {
            float tmax = max3f(s[t][0][0], s[t][0][1], s[t][0][2]);
            tmax = max3f(tmax, s[t][0][3], s[t][1][0]);
#pragma unroll
            for (int j = 1; j < JT; ++j) {
                tmax = max3f(tmax, s[t][j][1], s[t][j][2]);
                if (j + 1 < JT) tmax = max3f(tmax, s[t][j][3], s[t][j + 1][0]);
                else tmax = max2f(tmax, s[t][j][3]);
            }
            tmax = rows4_max(tmax);
            const float m_new = max2f(m_run[t], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
            m_run[t] = m_new;
            const f32x2 mm = {m_new, m_new};
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const f32x2 d0 = f32x2{s[t][j][0], s[t][j][1]} - mm, d1 = f32x2{s[t][j][2], s[t][j][3]} - mm;
                const f32x2 e0 = {__builtin_amdgcn_exp2f(d0.x), __builtin_amdgcn_exp2f(d0.y)};
                const f32x2 e1 = {__builtin_amdgcn_exp2f(d1.x), __builtin_amdgcn_exp2f(d1.y)};
                s[t][j][0] = e0.x; s[t][j][1] = e0.y; s[t][j][2] = e1.x; s[t][j][3] = e1.y;
                ps += e0;
                ps += e1;
            }
            l_run[t] = l_run[t] * alpha + (ps.x + ps.y);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[t][e] *= alpha;
        } else {
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const f32x2 e0 = {__builtin_amdgcn_exp2f(s[t][j][0]), __builtin_amdgcn_exp2f(s[t][j][1])};
                const f32x2 e1 = {__builtin_amdgcn_exp2f(s[t][j][2]), __builtin_amdgcn_exp2f(s[t][j][3])};
                s[t][j][0] = e0.x; s[t][j][1] = e0.y; s[t][j][2] = e1.x; s[t][j][3] = e1.y;
                ps += e0;
                ps += e1;
            }
            l_run[t] += ps.x + ps.y;
        }
    }
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            o[t] = mfma16(vf[j].x, s[t][j][0], o[t]);    // O^T += V^T[c = ql][key] * P^T[key][q]
            o[t] = mfma16(vf[j].y, s[t][j][1], o[t]);
            o[t] = mfma16(vf[j].z, s[t][j][2], o[t]);
            o[t] = mfma16(vf[j].w, s[t][j][3], o[t]);
        }
}

// The key loop of one wave for NTQ (1 or 2) 16-query tiles: S^T = K Q^T, softmax in the exp2 domain over blocks of 64
// keys, O^T += V^T P^T.  Returns O^T[c = 4*g4 + e][q = ql] and l per tile.
//
// fp32 MFMA and VALU instructions share the SIMD's issue time on gfx950 (tools/ubench/coissue_bench.hip: the cycles
// add, for any number of waves), so the softmax arithmetic is kept as short as it gets: the first block runs the online
// update and fixes the reference maximum, the others use it unchanged (softmax is shift invariant; a later logit above the
// reference only makes p > 1).  Should a logit exceed the reference by more than the fp32 exponent range the sum
// overflows to inf -- then, and only then, the wave redoes its tiles with the online update in every block.
template <int NTQ, bool MASKED>
PRD_DEV void ta_keyloop(const float* __restrict__ Kl, const float* __restrict__ Vt, const float* __restrict__ kadd,
                        const float4 (&qf)[NTQ], int npad, int ql, int g4, f32x4 (&o)[NTQ], float (&l_tot)[NTQ],
                        float* m_out = nullptr) {
    float m_run[NTQ], l_run[NTQ];
    bool online_all = false;                     // second pass only: online update in every block
    while (true) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            m_run[t] = -1e30f;
            l_run[t] = 0.f;
            o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int key0 = 0; key0 < npad; key0 += 64) {
            if (key0 == 0 || online_all) ta_block<NTQ, MASKED, true>(Kl, Vt, kadd, qf, npad, key0, ql, g4, m_run, l_run, o);
            else ta_block<NTQ, MASKED, false>(Kl, Vt, kadd, qf, npad, key0, ql, g4, m_run, l_run, o);
        }
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            l_tot[t] = rows4_sum(l_run[t]);
            bad |= !(l_tot[t] < 3.0e38f);        // inf or NaN
        }
        if (online_all || !__any(bad)) break;    // the second pass is never needed for logit spreads below 2^127
        online_all = true;
    }
    if (m_out) {                                 // the reference the sums are relative to (log2 domain), per query: chunk merges
#pragma unroll
        for (int t = 0; t < NTQ; ++t) m_out[t] = m_run[t];
    }
}

// Single-track gated attention core (reference modules.py:216-223 with the pair bias of :300-304), heads of width 16:
// o[b,q,h*16+c] = gate * softmax_k(q.k + bias[b,h,q,k], keys with mask < 0.5 filled with -2^15) v.  Replaces three launches
// (logits GEMM, row softmax, P*V GEMM) and the [b,H,N,N] logits round trip.
// The kernel is pure latency (0.2 GF, 1.3 MB at N = 320), so it is laid out WIDE: one workgroup per (b, h, 16 queries), the
// keys split in contiguous quarters over the four waves, every operand of a wave's first three 32-key blocks (K rows, V
// columns, pair bias, key mask -- straight from global memory in the lane layout of the swapped 16x16x4 MFMA scheme of the
// triangle attention) requested before the first MFMA; the four partial (max, sum, o) triples are merged through LDS in
// wave order.  (The first form -- 64 queries per workgroup, K / V^T of all nodes staged in LDS, ten dependent key blocks
// per wave -- took 14.9 us on 20 workgroups.)
struct SaBlock {                      // one 32-key block in the lane layout (lane = (query ql | key row, quad g4))
    float4 k[2];                      // K[key0 + 16 j + ql][4 g4 ..]
    float v[2][4], bi[2][4], mk[2][4];   // V[key][ql], bias[q][key], mask[key] for key = key0 + 16 j + 4 g4 + e
};

PRD_DEV void sa_load(SaBlock& s, const float* __restrict__ base, const float* __restrict__ brow, const float* __restrict__ mrow,
                     int N, int key0, int h, int ql, int g4, int L) {
    constexpr int C = 16, HC = 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kr = key0 + 16 * j + ql;
        s.k[j] = *reinterpret_cast<const float4*>(base + (size_t)(kr < N ? kr : N - 1) * L + HC + h * C + 4 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = key0 + 16 * j + 4 * g4 + e, kc = key < N ? key : N - 1;
            s.v[j][e] = base[(size_t)kc * L + 2 * HC + h * C + ql];
            s.bi[j][e] = brow[kc];
            s.mk[j][e] = mrow ? mrow[kc] : 1.0f;
        }
    }
}

__global__ __launch_bounds__(256) void single_attn_core_kernel(float* __restrict__ o_out, const float* __restrict__ qkvg,
                                                               const float* __restrict__ bias, const float* __restrict__ mask,
                                                               int b, int N, int H, int L) {
    constexpr int C = 16, HC = 64, RD = 3;                 // L: row pitch of qkvg in floats (>= 4 HC: the projection may sit inside a wider GEMM output)
    __shared__ float part[4][6][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, g4 = lane >> 4;
    const int qtiles = (N + 15) / 16;
    const int qt = blockIdx.x % qtiles;
    const int bh = blockIdx.x / qtiles;
    const int h = bh % H, bb = bh / H;
    const float* base = qkvg + (size_t)bb * N * L;
    const int q = qt * 16 + ql;
    const bool qok = q < N;
    const int qq = qok ? q : 0;
    const float* brow = bias + (((size_t)bb * H + h) * N + qq) * N;
    const float* mrow = mask ? mask + (size_t)bb * N : nullptr;
    const int nblk = (N + 31) / 32, per = (nblk + 3) / 4;
    const int blk0 = wave * per;
    const int nb = nblk - blk0 < per ? (nblk - blk0 > 0 ? nblk - blk0 : 0) : per;      // blocks of this wave (wave-uniform)
    SaBlock ring[RD];
    if (nb > 0) {
#pragma unroll
        for (int d = 0; d < RD; ++d) sa_load(ring[d], base, brow, mrow, N, (blk0 + (d < nb ? d : nb - 1)) * 32, h, ql, g4, L);
    }
    const float4 qf = *reinterpret_cast<const float4*>(base + (size_t)qq * L + h * C + 4 * g4);   // already scaled by 1/sqrt(c)
    float m_run = -1e30f, l_run = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < nb; i0 += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            const int i = i0 + d;
            if (i < nb) {                                          // wave-uniform
                const int key0 = (blk0 + i) * 32;
                const SaBlock& blk = ring[d];
                f32x4 s[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    z4 = mfma16(blk.k[j].x, qf.x, z4);
                    z4 = mfma16(blk.k[j].y, qf.y, z4);
                    z4 = mfma16(blk.k[j].z, qf.z, z4);
                    z4 = mfma16(blk.k[j].w, qf.w, z4);
                    s[j] = z4;
                }
                float tmax = -INFINITY;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = key0 + 16 * j + 4 * g4 + e;
                        const float v = (s[j][e] + blk.bi[j][e]) * LOG2E;         // logits + bias, exp2 domain
                        s[j][e] = key >= N ? -INFINITY : (blk.mk[j][e] >= 0.5f ? v : -32768.0f * LOG2E);
                        tmax = fmaxf(tmax, s[j][e]);
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
                        const float pe = __builtin_amdgcn_exp2f(s[j][e] - m_new);
                        s[j][e] = pe;
                        psum += pe;
                    }
                l_run = l_run * alpha + psum;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool vin = key0 + 16 * j + 4 * g4 < N;                 // clamped V rows of keys >= N meet p = 0, but guard NaN payloads
                    o = mfma16(vin ? blk.v[j][0] : 0.f, s[j][0], o);
                    o = mfma16(key0 + 16 * j + 4 * g4 + 1 < N ? blk.v[j][1] : 0.f, s[j][1], o);
                    o = mfma16(key0 + 16 * j + 4 * g4 + 2 < N ? blk.v[j][2] : 0.f, s[j][2], o);
                    o = mfma16(key0 + 16 * j + 4 * g4 + 3 < N ? blk.v[j][3] : 0.f, s[j][3], o);
                }
                const int nx = i + RD;                              // refill the slot (clamped: the tail re-reads the last block)
                sa_load(ring[d], base, brow, mrow, N, (blk0 + (nx < nb ? nx : nb - 1)) * 32, h, ql, g4, L);
            }
        }
    }
    const float l_w = rows4_sum(l_run);
    part[wave][0][lane] = m_run;
    part[wave][1][lane] = l_w;
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][2 + e][lane] = o[e];
    __syncthreads();
    if (wave == 0 && qok) {
        float m_all = part[0][0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) m_all = fmaxf(m_all, part[w][0][lane]);
        float l_tot = 0.f, ot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {                              // fixed order; a wave without keys has l = 0, o = 0
            const float sc = __builtin_amdgcn_exp2f(part[w][0][lane] - m_all);
            l_tot += part[w][1][lane] * sc;
#pragma unroll
            for (int e = 0; e < 4; ++e) ot[e] += part[w][2 + e][lane] * sc;
        }
        const float4 gf = *reinterpret_cast<const float4*>(base + (size_t)q * L + 3 * HC + h * C + 4 * g4);
        *reinterpret_cast<float4*>(o_out + ((size_t)bb * N + q) * HC + h * C + 4 * g4) =
            make_float4(gf.x * (ot[0] / l_tot), gf.y * (ot[1] / l_tot), gf.z * (ot[2] / l_tot), gf.w * (ot[3] / l_tot));
    }
}

// One workgroup (NW waves, persistent) serves head h = blockIdx % H for a strided set of pair rows.
//   phase 1: each wave LayerNorms 32-position blocks of the row and projects [k_h; v_h; q_h; g_h] (64 outputs)
//            on v_mfma_f32_32x32x2_f32; K, V^T, Q (pre-scaled by log2(e)/sqrt(c)) and the gate go to LDS in the
//            operand layouts of the 16x16x4 MFMAs.  The next row's block is already in registers (prefetch).
//   phase 2: the ceil(N/16) query tiles are dealt round-robin to the waves (SIMD-balanced: waves w and w+4
//            share a SIMD), two tiles at a time through ta_keyloop.
template <int P, int NW, bool PREFETCH, bool B3>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int npad, int H, int ending) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WSZ = B3 ? 3 * 64 * (2 * (P / 16) + 1) * 4 : 64 * (P + 4);   // B3: opt-in bf16 x 3 projections (prd_common.h)
    float* Wl = smem;                          // [64][P+4]: rows 0-15 k_h, 16-31 v_h, 32-47 q_h, 48-63 g_h
    float* Kl = Wl + WSZ;                      // [npad][KP]           (npad = round_up(N, 64))
    float* Vt = Kl + npad * KP;                // [16][npad+4]
    float* kadd = Vt + C * (npad + 4);         // [npad]: 0 = keep the logit, else the value that replaces it
    float* Ql = kadd + npad;                   // [npad][KP]
    float* Gl = Ql + npad * KP;                // [npad][KP]
    float* bqg = Gl + npad * KP;               // [2][16]: accumulator preload of the [q; g] half (0 | -log2e * gate bias)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ql = lane & 15, g4 = lane >> 4;
    const int nqb = (N + 31) / 32;
    const int ntile = (N + 15) / 16;
    // workgroup -> (head, row slot).  Workgroup w is observed to run on XCD w % 8; the H heads of one row
    // are given to workgroups of the SAME XCD so that the row is fetched into one L2 only (speed only).
    const int rstride = gridDim.x / H;          // row slots
    int h, slot;
    if ((rstride & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        h = idx % H;
        slot = (idx / H) * 8 + xcd;
    } else {
        h = blockIdx.x % H;
        slot = blockIdx.x / H;
    }
    const float sc = 0.25f * LOG2E;            // 1/sqrt(c) (modules.py:176,216) in the exp2 domain, folded into Wq
    if (B3) {       // one image of 64 rows; stage_weight_b3 takes (first row pointer, rows staged, plane stride in rows)
        u32x4* Wb = reinterpret_cast<u32x4*>(Wl);
        constexpr int PITCH = 2 * (P / 16) + 1;
        stage_weight_b3_rows<P>(Wb, 64, 0, wk + (long)h * C * P, C, P, tid, NT, 1.0f);
        stage_weight_b3_rows<P>(Wb, 64, C, wv + (long)h * C * P, C, P, tid, NT, 1.0f);
        stage_weight_b3_rows<P>(Wb, 64, 2 * C, wq + (long)h * C * P, C, P, tid, NT, sc);
        stage_weight_b3_rows<P>(Wb, 64, 3 * C, wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E);
        (void)PITCH;
    } else {
        stage_weight_cll<P>(Wl, wk + (long)h * C * P, C, P, tid, NT);
        stage_weight_cll<P>(Wl + C * (P + 4), wv + (long)h * C * P, C, P, tid, NT);
        stage_weight_cll<P>(Wl + 2 * C * (P + 4), wq + (long)h * C * P, C, P, tid, NT, sc);
        stage_weight_cll<P>(Wl + 3 * C * (P + 4), wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E);   // gate_from_scaled
    }
    // accumulator preload of the [q; g] half: lane half hi owns D rows q{4hi+e}, q{8+4hi+e}, g{4hi+e}, g{8+4hi+e}
    if (tid < 32) {
        const int hh = tid >> 4, e = tid & 15;
        bqg[tid] = e < 8 ? 0.f : NEG_LOG2E * bg[h * C + (e < 12 ? 4 * hh + (e - 8) : 8 + 4 * hh + (e - 12))];
    }
    const long nrows = (long)b * N;

    auto row_pos = [&](long bu, int v) -> long {          // pair position of sequence element v of row bu
        const long bb = bu / N;
        const long u = bu - bb * N;
        return ending ? ((bb * N + v) * N + u) : (bu * N + v);
    };
    // ---- phase-1 work of this wave: (block, halves) units.  Half 0 = [k; v], half 1 = [q; gate] (32 MFMAs each).
    // Whole rounds of NW blocks go one block per wave; of the R blocks left, as many as needed are split in halves
    // over two waves so that the SIMDs (waves w, w+4, w+8 share one) end together: N = 320 -> waves 0-7 a block each,
    // waves 8-11 half a block each = 5 half units per SIMD instead of 6 / 6 / 4 / 4.
    const int full_rounds = nqb / NW, R = nqb - full_rounds * NW;
    const int S = 2 * R <= NW ? R : NW - R;               // blocks of the last round that are split
    const int F = R - S;                                   // blocks of the last round done whole by waves 0..F-1
    int last_blk = -1, last_halves = 0;
    if (wave < F) { last_blk = full_rounds * NW + wave; last_halves = 3; }
    else if (wave - F < 2 * S) { last_blk = full_rounds * NW + F + (wave - F) % S; last_halves = 1 << ((wave - F) / S); }
    const int nunits = full_rounds + (last_blk >= 0 ? 1 : 0);
    const int first_blk = full_rounds > 0 ? wave : last_blk;
    // positions past the last real block never change: K = V = Q = 0, logit override -inf
    for (int v = nqb * 32 + tid; v < npad; v += NT) {
#pragma unroll
        for (int e = 0; e < C; ++e) { Kl[v * KP + e] = 0.f; Ql[v * KP + e] = 0.f; Gl[v * KP + e] = 0.f; Vt[e * (npad + 4) + v] = 0.f; }
        kadd[v] = -INFINITY;
    }
    // prefetch of this wave's first block of the first row
    float xnext[KH];
    if (PREFETCH) {
        const long bu0 = slot;
        const int v = first_blk * 32 + r;
        const bool ok = bu0 < nrows && first_blk >= 0 && v < N;
        load_row_cll<P>(pair + row_pos(ok ? bu0 : 0, ok ? v : 0) * P, hi, ok, xnext);
    }
    int it = 0;
    for (long bu = slot; bu < nrows; bu += rstride, ++it) {
        const int bb = (int)(bu / N);
        __syncthreads();                        // previous row's LDS fully consumed (and weights staged)
        PRD_STAMP(0);
        const float mu = mask[bu];
        // ---- phase 1 ----
        for (int un = 0; un < nunits; ++un) {
            const int vb = un < full_rounds ? un * NW + wave : last_blk;
            const int halves = un < full_rounds ? 3 : last_halves;
            const int v = vb * 32 + r;
            const bool valid = v < N;
            float x[KH];
            if (PREFETCH && un == 0) {
#pragma unroll
                for (int s = 0; s < KH; ++s) x[s] = xnext[s];
            } else {
                load_row_cll<P>(pair + row_pos(bu, valid ? v : 0) * P, hi, valid, x);
            }
            ln_cll<KH>(x);
            u32x4 xs[3][P / 16];
            if (B3) split3_cll<P>(x, xs);
            if (halves & 1) {
                if (hi == 0) {                                // per-key logit override of this row (see header comment)
                    const bool keep = valid && (mu * mask[(long)bb * N + (valid ? v : 0)] >= 0.5f);
                    kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E : -INFINITY);
                }
                f32x16 acc[1];
                zero_acc(acc);
                if (B3) rowgemm_b3<P, 1>(reinterpret_cast<const u32x4*>(Wl), 64, 0, xs, acc, r, hi);
                else rowgemm<P, 1>(Wl, x, acc, r, hi);
                // D rows of a block: channels {4hi+e} in registers 0-3 and {8+4hi+e} in 4-7 of each 16-row group
                *reinterpret_cast<float4*>(Kl + v * KP + 4 * hi) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4*>(Kl + v * KP + 8 + 4 * hi) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Vt[(4 * hi + e) * (npad + 4) + v] = acc[0][8 + e];
                    Vt[(8 + 4 * hi + e) * (npad + 4) + v] = acc[0][12 + e];
                }
            }
            if (halves & 2) {
                f32x16 acc[1];
                bias_acc(acc, bqg + 16 * hi);
                if (B3) rowgemm_b3<P, 1>(reinterpret_cast<const u32x4*>(Wl), 64, 32, xs, acc, r, hi);
                else rowgemm<P, 1>(Wl + 2 * C * (P + 4), x, acc, r, hi);
                *reinterpret_cast<float4*>(Ql + v * KP + 4 * hi) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4*>(Ql + v * KP + 8 + 4 * hi) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
                *reinterpret_cast<float4*>(Gl + v * KP + 4 * hi) = make_float4(gate_from_scaled(acc[0][8]), gate_from_scaled(acc[0][9]),
                                                                                gate_from_scaled(acc[0][10]), gate_from_scaled(acc[0][11]));
                *reinterpret_cast<float4*>(Gl + v * KP + 8 + 4 * hi) = make_float4(gate_from_scaled(acc[0][12]), gate_from_scaled(acc[0][13]),
                                                                                    gate_from_scaled(acc[0][14]), gate_from_scaled(acc[0][15]));
            }
        }
        PRD_STAMP(1);
        __syncthreads();
        PRD_STAMP(2);
        // next row's first block: in flight during the whole key loop
        if (PREFETCH) {
            const long bun = bu + rstride;
            const int v = first_blk * 32 + r;
            const bool ok = bun < nrows && first_blk >= 0 && v < N;
            load_row_cll<P>(pair + row_pos(ok ? bun : 0, ok ? v : 0) * P, hi, ok, xnext);
        }
        // ---- phase 2 ----
        // rows without masked or padded keys (the common case) take a key loop without the per-key override
        bool row_masked = false;
        for (int k = lane; k < npad; k += 64) row_masked |= (kadd[k] != 0.f);
        row_masked = __any(row_masked);
        for (int t0 = wave; t0 < ntile; t0 += 2 * NW) {
            const int t1 = t0 + NW;
            if (t1 < ntile) {
                float4 qf[2];
                qf[0] = *reinterpret_cast<const float4*>(Ql + (16 * t0 + ql) * KP + 4 * g4);
                qf[1] = *reinterpret_cast<const float4*>(Ql + (16 * t1 + ql) * KP + 4 * g4);
                f32x4 o[2];
                float l_tot[2];
                if (row_masked) ta_keyloop<2, true>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot);
                else ta_keyloop<2, false>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int v = 16 * (t == 0 ? t0 : t1) + ql;
                    if (v < N) {
                        const float4 gf = *reinterpret_cast<const float4*>(Gl + v * KP + 4 * g4);
                        *reinterpret_cast<float4*>(og + row_pos(bu, v) * HC + h * C + 4 * g4) =
                            make_float4(gf.x * (o[t][0] / l_tot[t]), gf.y * (o[t][1] / l_tot[t]),
                                        gf.z * (o[t][2] / l_tot[t]), gf.w * (o[t][3] / l_tot[t]));
                    }
                }
            } else {
                float4 qf[1];
                qf[0] = *reinterpret_cast<const float4*>(Ql + (16 * t0 + ql) * KP + 4 * g4);
                f32x4 o[1];
                float l_tot[1];
                if (row_masked) ta_keyloop<1, true>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot);
                else ta_keyloop<1, false>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot);
                const int v = 16 * t0 + ql;
                if (v < N) {
                    const float4 gf = *reinterpret_cast<const float4*>(Gl + v * KP + 4 * g4);
                    *reinterpret_cast<float4*>(og + row_pos(bu, v) * HC + h * C + 4 * g4) =
                        make_float4(gf.x * (o[0][0] / l_tot[0]), gf.y * (o[0][1] / l_tot[0]),
                                    gf.z * (o[0][2] / l_tot[0]), gf.w * (o[0][3] / l_tot[0]));
                }
            }
        }
        PRD_STAMP(3);
    }
}

// ---- FIRST-GENERATION split-16 attention cores (round 2): compiled only with -DPRD_AB (python -m protein_redesign_amd.build --ab ->
// libprd_hip_ab.so, for A/B measurements and their own parity tests).  The shipped library serves the split-16 arithmetic with the
// second generation (csrc/prd_tri2.hip) and, where that does not apply, with the fp32-MFMA kernels of this file. ----
#ifdef PRD_AB
// ---------------------------------------------------------------------------------------------------------------------
// Triangle attention core on the 16-bit matrix pipes (gemm mode 1), fp32-accurate by operand splitting:
//   * projections: bf16 x 3 row GEMM (rowgemm_b3), as in the kernel above;
//   * S^T = K Q^T: K and Q are split exactly into three bf16 parts ONCE per row in phase 1 (the key loop pays nothing for it).
//     The head width is 16 but v_mfma_f32_16x16x32_bf16 contracts 32: two 16-wide products ride in one instruction,
//       [kh | kh] x [qh | qm],  [km | kl] x [qh | qh],  [km | kh] x [qm | ql]
//     = all six products of the three-way split in 3 MFMAs of ~17 cycles instead of 4 fp32 MFMAs of 32 cycles;
//   * O^T = V^T P^T: V (x 16, a power of two) and the probabilities are split into fp16 hi + lo (RTZ hi, so hi + lo carries
//     22 bits; |p| <= 1 after the first block, |16 v| is far inside the fp16 range); hi*hi + hi*lo + lo*hi on
//     v_mfma_f32_16x16x32_f16: 6 MFMAs per 64 keys instead of 16 fp32 ones.  Splitting p costs VALU per logit
//     (cvt_pkrtz, cvt, sub, cvt_pkrtz), which is why the probabilities use the 2-part fp16 form and not the 3-part bf16 one.
//     A probability that would leave the fp16 range (only possible in the frozen-maximum blocks) sends the wave to the
//     online pass, where p <= 1.
// LDS per position: K / Q planes 3 x 32 B each (row pitch 32 B: the b128 operand reads of a 16-lane group are conflict free),
// V hi / lo transposed [16][npad + 8] fp16, gate fp32.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));      // what __builtin_amdgcn_cvt_pkrtz returns

PRD_DEV unsigned pk_f16_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
// fp16 hi / lo words of the pair (a, b): hi = RTZ(x), lo = RTZ(x - hi) (the subtraction is exact)
PRD_DEV void split2_f16(float a, float b, unsigned& hi, unsigned& lo) { split2h(a, b, hi, lo); }

struct SplitLds {               // byte offsets from the dynamic LDS base
    unsigned kp[3], qp[3], vh, vl, gl, kadd, bqg;
    unsigned vpitch;            // halfs per V^T row
};

// NM = QK^T MFMAs per 16-key tile: 3 = bf16 x 3 operands (short rows), 2 = fp16 x 2 operands (long rows, see the long kernel)
template <int NM>
PRD_DEV f32x4 qk_mfma(u32x4 a, u32x4 b, f32x4 c) {
    if (NM == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int NTQ, int NM, bool MASKED, bool ONLINE>
PRD_DEV void ta_block_split(const unsigned char* __restrict__ lds, const SplitLds& L, const unsigned (&kbase)[NM],
                            const u32x4 (&qb)[NTQ][NM], int npad, int key0, int ql, int g4,
                            float (&m_run)[NTQ], float (&l_run)[NTQ], f32x4 (&o)[NTQ][2], bool& big) {
    constexpr int JT = 4;
    ta_prio(npad - key0, npad);
    const float* kadd = reinterpret_cast<const float*>(lds + L.kadd);
    float4 ma[JT];
    f32x4 s[NTQ][JT];
    u32x4 ka[JT][NM];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const unsigned krow = (unsigned)(key0 + 16 * j + ql) * 32u + (unsigned)(g4 & 1) * 16u;
#pragma unroll
        for (int m = 0; m < NM; ++m) ka[j][m] = *reinterpret_cast<const u32x4*>(lds + kbase[m] + krow);
        if (MASKED) ma[j] = *reinterpret_cast<const float4*>(kadd + key0 + 16 * j + 4 * g4);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const float c0 = ONLINE ? 0.f : -m_run[t];
            s[t][j] = f32x4{c0, c0, c0, c0};
        }
    }
    // the three MFMAs of a logit tile are issued NTQ * JT independent accumulators apart (no back-to-back dependent pair)
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int t = 0; t < NTQ; ++t) s[t][j] = qk_mfma<NM>(ka[j][m], qb[t][m], s[t][j]);
    __builtin_amdgcn_sched_barrier(0);                  // V^T is fetched behind the QK^T MFMAs, not before them
    // A operands of P V: lane (c = ql, g4) holds V^T[c][keys 16 j0 + 4 g4 .. +3 | 16 (j0 + 1) + 4 g4 .. +3] for j0 = 0, 2
    u32x4 vh[2], vl[2];
    {
        const unsigned vrow = ((unsigned)ql * L.vpitch + (unsigned)key0 + 4u * g4) * 2u;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const u32x2 a = *reinterpret_cast<const u32x2*>(lds + L.vh + vrow + 64 * gi);
            const u32x2 b = *reinterpret_cast<const u32x2*>(lds + L.vh + vrow + 64 * gi + 32);
            const u32x2 c = *reinterpret_cast<const u32x2*>(lds + L.vl + vrow + 64 * gi);
            const u32x2 d = *reinterpret_cast<const u32x2*>(lds + L.vl + vrow + 64 * gi + 32);
            vh[gi] = u32x4{a[0], a[1], b[0], b[1]};
            vl[gi] = u32x4{c[0], c[1], d[0], d[1]};
        }
    }
    u32x4 php[NTQ][2], plp[NTQ][2];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        if (MASKED) {
            const float mr = ONLINE ? 0.f : m_run[t];   // override values are absolute logits
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                s[t][j][0] = (ma[j].x == 0.f) ? s[t][j][0] : ma[j].x - mr;
                s[t][j][1] = (ma[j].y == 0.f) ? s[t][j][1] : ma[j].y - mr;
                s[t][j][2] = (ma[j].z == 0.f) ? s[t][j][2] : ma[j].z - mr;
                s[t][j][3] = (ma[j].w == 0.f) ? s[t][j][3] : ma[j].w - mr;
            }
        }
        f32x2 ps = {0.f, 0.f};
        if (ONLINE) {
            float tmax = max3f(s[t][0][0], s[t][0][1], s[t][0][2]);
            tmax = max3f(tmax, s[t][0][3], s[t][1][0]);
#pragma unroll
            for (int j = 1; j < JT; ++j) {
                tmax = max3f(tmax, s[t][j][1], s[t][j][2]);
                if (j + 1 < JT) tmax = max3f(tmax, s[t][j][3], s[t][j + 1][0]);
                else tmax = max2f(tmax, s[t][j][3]);
            }
            tmax = rows4_max(tmax);
            const float m_new = max2f(m_run[t], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
            m_run[t] = m_new;
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[t][j][e] = __builtin_amdgcn_exp2f(s[t][j][e] - m_new);
                    ps[e & 1] += s[t][j][e];
                }
            l_run[t] = l_run[t] * alpha + (ps.x + ps.y);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[t][0][e] *= alpha; o[t][1][e] *= alpha; }
        } else {
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[t][j][e] = __builtin_amdgcn_exp2f(s[t][j][e]);
                    ps[e & 1] += s[t][j][e];
                }
            const float bsum = ps.x + ps.y;
            big |= !(bsum < 30000.0f);                  // a probability near the fp16 range (or inf / NaN): redo online
            l_run[t] += bsum;
        }
        // probabilities -> fp16 hi / lo B operands of the two 32-key groups
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
                unsigned h0, l0, h1, l1;
                split2_f16(s[t][2 * gi + hj][0], s[t][2 * gi + hj][1], h0, l0);
                split2_f16(s[t][2 * gi + hj][2], s[t][2 * gi + hj][3], h1, l1);
                php[t][gi][2 * hj] = h0; php[t][gi][2 * hj + 1] = h1;
                plp[t][gi][2 * hj] = l0; plp[t][gi][2 * hj + 1] = l1;
            }
        }
    }
    // O^T += V^T P^T: the three products of an accumulator are issued 2 NTQ independent accumulators apart
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
                o[t][gi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, pr == 2 ? vl[gi] : vh[gi]),
                                                                  __builtin_bit_cast(f16x8, pr == 1 ? plp[t][gi] : php[t][gi]), o[t][gi], 0, 0, 0);
}

template <int NTQ, int NM, bool MASKED>
PRD_DEV void ta_keyloop_split(const unsigned char* __restrict__ lds, const SplitLds& L, const unsigned (&kbase)[NM],
                              const u32x4 (&qb)[NTQ][NM], int npad, int ql, int g4, f32x4 (&o)[NTQ], float (&l_tot)[NTQ]) {
    float m_run[NTQ], l_run[NTQ];
    f32x4 o2[NTQ][2];                            // one accumulator per 32-key group: two independent MFMA chains per tile
    bool online_all = false;
    while (true) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            m_run[t] = -1e30f;
            l_run[t] = 0.f;
            o2[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            o2[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bool big = false;
#pragma unroll 1
        for (int key0 = 0; key0 < npad; key0 += 64) {
            if (key0 == 0 || online_all) ta_block_split<NTQ, NM, MASKED, true>(lds, L, kbase, qb, npad, key0, ql, g4, m_run, l_run, o2, big);
            else ta_block_split<NTQ, NM, MASKED, false>(lds, L, kbase, qb, npad, key0, ql, g4, m_run, l_run, o2, big);
        }
        bool bad = big;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            o[t] = o2[t][0] + o2[t][1];
            l_tot[t] = rows4_sum(l_run[t]);
            bad |= !(l_tot[t] < 3.0e38f);
        }
        if (online_all || !__any(bad)) break;
        online_all = true;
    }
}

// FUSE: the row this kernel attends over is not `pair` itself but pair + og_in W_o^T + b_o -- the residual update of the
// PREVIOUS triangle attention (its output projection, modules.py:339-340) applied while the row is loaded, instead of a
// separate tri_attn_out launch and pass over the pair tensor.  The four head-workgroups of a row all apply it (24 MFMAs per
// 32 positions, ~5 % of a row's cycles); the workgroup of head 0 writes the updated row to `pair_out`, which must be a
// DIFFERENT buffer than `pair` (the other heads of the row read `pair` concurrently).
template <int P, int NW, int NTQ, bool PREFETCH, bool FUSE = false>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_split_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int npad, int H, int ending,
    const float* __restrict__ og_in = nullptr, const float* __restrict__ wo_in = nullptr, const float* __restrict__ bo_in = nullptr,
    float* __restrict__ pair_out = nullptr) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;         // the projection weights are staged x 16 (rowgemm_h2): V stays x 16 (exact), so
                                                // that its small components keep a normal fp16 lo part; k, q, gate are scaled back
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr unsigned WBYTES = 2u * 64 * (P / 8) * 16;      // fp16 hi | lo planes of the 64 projection rows
    SplitLds L;
    {
        unsigned off = WBYTES;
        for (int pl = 0; pl < 3; ++pl) { L.kp[pl] = off; off += (unsigned)npad * 32u; }
        for (int pl = 0; pl < 3; ++pl) { L.qp[pl] = off; off += (unsigned)npad * 32u; }
        L.vpitch = (unsigned)npad + 8u;
        L.vh = off; off += 16u * L.vpitch * 2u;
        L.vl = off; off += 16u * L.vpitch * 2u;
        L.gl = off; off += (unsigned)npad * KP * 4u;
        L.kadd = off; off += (unsigned)npad * 4u;
        L.bqg = off;
    }
    u32x4* Wb = reinterpret_cast<u32x4*>(lds);
    float* Gl = reinterpret_cast<float*>(lds + L.gl);
    float* kadd = reinterpret_cast<float*>(lds + L.kadd);
    float* bqg = reinterpret_cast<float*>(lds + L.bqg);
    float* bol = bqg + 32;                                             // FUSE: [P] CLL, then the W_o image (fp16 hi | lo planes, x 16)
    u32x4* Wob = reinterpret_cast<u32x4*>(lds + L.bqg + 128 + P * 4);
    _Float16* Vh = reinterpret_cast<_Float16*>(lds + L.vh);
    _Float16* Vl = reinterpret_cast<_Float16*>(lds + L.vl);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ql = lane & 15, g4 = lane >> 4;
    const int nqb = (N + 31) / 32;
    const int ntile = (N + 15) / 16;
    const int rstride = gridDim.x / H;
    if (FUSE) {
        stage_weight_h2<HC>(Wob, wo_in, P, HC, tid, NT, H2_WSCALE);
        stage_vec_cll(bol, bo_in, P, tid, NT);
    }
    int h, slot;
    if ((rstride & 7) == 0) {                   // the H heads of one row on one XCD (see tri_attn_core_kernel)
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        h = idx % H;
        slot = (idx / H) * 8 + xcd;
    } else {
        h = blockIdx.x % H;
        slot = blockIdx.x / H;
    }
    const float sc = 0.25f * LOG2E;
    stage_weight_h2_rows<P>(Wb, 64, 0, wk + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, C, wv + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 2 * C, wq + (long)h * C * P, C, P, tid, NT, sc * H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 3 * C, wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E * H2_WSCALE);
    if (tid < 32) {
        const int hh = tid >> 4, e = tid & 15;
        bqg[tid] = e < 8 ? 0.f : H2_WSCALE * NEG_LOG2E * bg[h * C + (e < 12 ? 4 * hh + (e - 8) : 8 + 4 * hh + (e - 12))];
    }
    const long nrows = (long)b * N;
    auto row_pos = [&](long bu, int v) -> long {
        const long bb = bu / N;
        const long u = bu - bb * N;
        return ending ? ((bb * N + v) * N + u) : (bu * N + v);
    };
    const int full_rounds = nqb / NW, R = nqb - full_rounds * NW;
    const int S = 2 * R <= NW ? R : NW - R;
    const int F = R - S;
    int last_blk = -1, last_halves = 0;
    if (wave < F) { last_blk = full_rounds * NW + wave; last_halves = 3; }
    else if (wave - F < 2 * S) { last_blk = full_rounds * NW + F + (wave - F) % S; last_halves = 1 << ((wave - F) / S); }
    const int nunits = full_rounds + (last_blk >= 0 ? 1 : 0);
    const int first_blk = full_rounds > 0 ? wave : last_blk;
    // positions past the last real block never change: K = V = Q = 0, logit override -inf
    for (int v = nqb * 32 + tid; v < npad; v += NT) {
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<u32x4*>(lds + L.kp[pl] + v * 32) = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(lds + L.kp[pl] + v * 32 + 16) = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(lds + L.qp[pl] + v * 32) = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(lds + L.qp[pl] + v * 32 + 16) = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int e = 0; e < C; ++e) { Gl[v * KP + e] = 0.f; Vh[e * L.vpitch + v] = (_Float16)0.f; Vl[e * L.vpitch + v] = (_Float16)0.f; }
        kadd[v] = -INFINITY;
    }
    float xnext[PREFETCH ? KH : 1];
    if (PREFETCH) {
        const long bu0 = slot;
        const int v = first_blk * 32 + r;
        const bool ok = bu0 < nrows && first_blk >= 0 && v < N;
        load_row_cll<P>(pair + row_pos(ok ? bu0 : 0, ok ? v : 0) * P, hi, ok, reinterpret_cast<float (&)[KH]>(xnext));
    }
    // per-lane operand bases of the three QK^T MFMAs (see the header comment): A = K planes, B = Q planes
    const unsigned kbase[3] = {L.kp[0], g4 < 2 ? L.kp[1] : L.kp[2], g4 < 2 ? L.kp[1] : L.kp[0]};
    const unsigned qbase[3] = {g4 < 2 ? L.qp[0] : L.qp[1], L.qp[0], g4 < 2 ? L.qp[1] : L.qp[2]};
    int it = 0;
    for (long bu = slot; bu < nrows; bu += rstride, ++it) {
        const int bb = (int)(bu / N);
        __syncthreads();                        // previous row's LDS fully consumed (and weights staged)
        PRD_STAMP(0);
        const float mu = mask[bu];
        // ---- phase 1: project, split, store ----
        for (int un = 0; un < nunits; ++un) {
            const int vb = un < full_rounds ? un * NW + wave : last_blk;
            const int halves = un < full_rounds ? 3 : last_halves;
            const int v = vb * 32 + r;
            const bool valid = v < N;
            float x[KH];
            if (PREFETCH && un == 0) {
#pragma unroll
                for (int s_ = 0; s_ < KH; ++s_) x[s_] = xnext[PREFETCH ? s_ : 0];
            } else {
                load_row_cll<P>(pair + row_pos(bu, valid ? v : 0) * P, hi, valid, x);
            }
            if (FUSE) {                         // x += og_in W_o^T + b_o: the previous attention's residual update of this row
                const long pos = row_pos(bu, valid ? v : 0);
                float xo[HC / 2];
                load_row_cll<HC>(og_in + pos * HC, hi, valid, xo);
                u32x4 os[2][HC / 16];
                split2h_cll<HC / 2>(xo, os);
                f32x16 ao[P / 32];
                zero_acc(ao);
                rowgemm_h2<HC, P / 32>(Wob, P, 0, os, ao, r, hi);
#pragma unroll
                for (int s_ = 0; s_ < KH; ++s_) x[s_] = x[s_] + (ao[s_ >> 4][s_ & 15] * H2_INV_WSCALE + bol[hi * KH + s_]);
                if (h == 0 && (halves & 1)) store_row_cll<P>(pair_out + pos * P, hi, valid, x);
            }
            ln_cll<KH>(x);
            u32x4 xs[2][P / 16];
            split2h_cll<KH>(x, xs);
            if (halves & 1) {
                if (hi == 0) {
                    const bool keep = valid && (mu * mask[(long)bb * N + (valid ? v : 0)] >= 0.5f);
                    kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E : -INFINITY);
                }
                f32x16 acc[1];
                zero_acc(acc);
                rowgemm_h2<P, 1>(Wb, 64, 0, xs, acc, r, hi);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[0][e] *= H2_INV_WSCALE;
                // k channels {4hi+e} in registers 0-3 and {8+4hi+e} in 4-7: three bf16 planes, 8 bytes each
                unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                split3(acc[0][0], acc[0][1], h0, m0, l0);
                split3(acc[0][2], acc[0][3], h1, m1, l1);
                split3(acc[0][4], acc[0][5], h2, m2, l2);
                split3(acc[0][6], acc[0][7], h3, m3, l3);
                const unsigned ko = (unsigned)v * 32u + 8u * hi;
                *reinterpret_cast<u32x2*>(lds + L.kp[0] + ko) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(lds + L.kp[0] + ko + 16) = u32x2{h2, h3};
                *reinterpret_cast<u32x2*>(lds + L.kp[1] + ko) = u32x2{m0, m1};
                *reinterpret_cast<u32x2*>(lds + L.kp[1] + ko + 16) = u32x2{m2, m3};
                *reinterpret_cast<u32x2*>(lds + L.kp[2] + ko) = u32x2{l0, l1};
                *reinterpret_cast<u32x2*>(lds + L.kp[2] + ko + 16) = u32x2{l2, l3};
                // v channels likewise in registers 8-15 (already x 16): fp16 hi / lo, transposed
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const int c0 = (e < 4 ? 4 * hi : 8 + 4 * hi) + (e & 3);
                    unsigned hw, lw;
                    split2h(acc[0][8 + e], acc[0][9 + e], hw, lw);
                    const f16x2 hh2 = __builtin_bit_cast(f16x2, hw), ll2 = __builtin_bit_cast(f16x2, lw);
                    Vh[c0 * L.vpitch + v] = (_Float16)hh2[0];
                    Vh[(c0 + 1) * L.vpitch + v] = (_Float16)hh2[1];
                    Vl[c0 * L.vpitch + v] = (_Float16)ll2[0];
                    Vl[(c0 + 1) * L.vpitch + v] = (_Float16)ll2[1];
                }
            }
            if (halves & 2) {
                f32x16 acc[1];
                bias_acc(acc, bqg + 16 * hi);
                rowgemm_h2<P, 1>(Wb, 64, 32, xs, acc, r, hi);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[0][e] *= H2_INV_WSCALE;
                unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                split3(acc[0][0], acc[0][1], h0, m0, l0);
                split3(acc[0][2], acc[0][3], h1, m1, l1);
                split3(acc[0][4], acc[0][5], h2, m2, l2);
                split3(acc[0][6], acc[0][7], h3, m3, l3);
                const unsigned qo = (unsigned)v * 32u + 8u * hi;
                *reinterpret_cast<u32x2*>(lds + L.qp[0] + qo) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(lds + L.qp[0] + qo + 16) = u32x2{h2, h3};
                *reinterpret_cast<u32x2*>(lds + L.qp[1] + qo) = u32x2{m0, m1};
                *reinterpret_cast<u32x2*>(lds + L.qp[1] + qo + 16) = u32x2{m2, m3};
                *reinterpret_cast<u32x2*>(lds + L.qp[2] + qo) = u32x2{l0, l1};
                *reinterpret_cast<u32x2*>(lds + L.qp[2] + qo + 16) = u32x2{l2, l3};
                *reinterpret_cast<float4*>(Gl + v * KP + 4 * hi) = make_float4(gate_from_scaled(acc[0][8]), gate_from_scaled(acc[0][9]),
                                                                                gate_from_scaled(acc[0][10]), gate_from_scaled(acc[0][11]));
                *reinterpret_cast<float4*>(Gl + v * KP + 8 + 4 * hi) = make_float4(gate_from_scaled(acc[0][12]), gate_from_scaled(acc[0][13]),
                                                                                    gate_from_scaled(acc[0][14]), gate_from_scaled(acc[0][15]));
            }
        }
        PRD_STAMP(1);
        __syncthreads();
        PRD_STAMP(2);
        if (PREFETCH) {   // next row's first block: in flight during the whole key loop
            const long bun = bu + rstride;
            const int v = first_blk * 32 + r;
            const bool ok = bun < nrows && first_blk >= 0 && v < N;
            load_row_cll<P>(pair + row_pos(ok ? bun : 0, ok ? v : 0) * P, hi, ok, reinterpret_cast<float (&)[KH]>(xnext));
        }
        // ---- phase 2 ----
        bool row_masked = false;
        for (int k = lane; k < npad; k += 64) row_masked |= (kadd[k] != 0.f);
        row_masked = __any(row_masked);
        const float inv_vs = 1.0f / VSCALE;
        for (int t0 = wave; t0 < ntile; t0 += NTQ * NW) {
            const int t1 = t0 + NW;
            const unsigned half16 = (unsigned)(g4 & 1) * 16u;
            if (NTQ == 2 && t1 < ntile) {
                u32x4 qb[2][3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    qb[0][m] = *reinterpret_cast<const u32x4*>(lds + qbase[m] + (unsigned)(16 * t0 + ql) * 32u + half16);
                    qb[1][m] = *reinterpret_cast<const u32x4*>(lds + qbase[m] + (unsigned)(16 * t1 + ql) * 32u + half16);
                }
                f32x4 o[2];
                float l_tot[2];
                if (row_masked) ta_keyloop_split<2, 3, true>(lds, L, kbase, qb, npad, ql, g4, o, l_tot);
                else ta_keyloop_split<2, 3, false>(lds, L, kbase, qb, npad, ql, g4, o, l_tot);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int v = 16 * (t == 0 ? t0 : t1) + ql;
                    if (v < N) {
                        const float4 gf = *reinterpret_cast<const float4*>(Gl + v * KP + 4 * g4);
                        const float il = inv_vs / l_tot[t];
                        *reinterpret_cast<float4*>(og + row_pos(bu, v) * HC + h * C + 4 * g4) =
                            make_float4(gf.x * (o[t][0] * il), gf.y * (o[t][1] * il), gf.z * (o[t][2] * il), gf.w * (o[t][3] * il));
                    }
                }
            } else {
                u32x4 qb[1][3];
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    qb[0][m] = *reinterpret_cast<const u32x4*>(lds + qbase[m] + (unsigned)(16 * t0 + ql) * 32u + half16);
                f32x4 o[1];
                float l_tot[1];
                if (row_masked) ta_keyloop_split<1, 3, true>(lds, L, kbase, qb, npad, ql, g4, o, l_tot);
                else ta_keyloop_split<1, 3, false>(lds, L, kbase, qb, npad, ql, g4, o, l_tot);
                const int v = 16 * t0 + ql;
                if (v < N) {
                    const float4 gf = *reinterpret_cast<const float4*>(Gl + v * KP + 4 * g4);
                    const float il = inv_vs / l_tot[0];
                    *reinterpret_cast<float4*>(og + row_pos(bu, v) * HC + h * C + 4 * g4) =
                        make_float4(gf.x * (o[0][0] * il), gf.y * (o[0][1] * il), gf.z * (o[0][2] * il), gf.w * (o[0][3] * il));
                }
            }
        }
        PRD_STAMP(3);
    }
}

// Long rows on the 16-bit pipes (gemm mode 1, 512 < N <= 832): K / V of the whole row as fp16 hi | lo planes (132 B per position)
// fill the LDS, so the queries are re-projected per 32-query block in phase 2 into a small per-wave scratch (Q hi | lo planes and
// the gate), which the key loop then reads exactly like the short-row kernel reads its row-wide Q planes.  Q K^T uses the
// fp16 x 2 form here (2 MFMAs per 16-key tile: [kh|kh] x [qh|ql] and [kl|kl] x [qh|0]; 22-bit operands, |q|, |k| = O(1) projections
// of LayerNorm-ed rows) because three bf16 planes of K do not fit; P V as in the short-row kernel.  The H heads of a row run on
// one XCD (the fp32 long-row kernel fetched every row into four L2s: 1.19 GB of fabric reads per launch at N = 769).
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_split_long_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int npad, int H, int ending) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    constexpr float VSCALE = H2_WSCALE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr unsigned WBYTES = 2u * 64 * (P / 8) * 16;
    SplitLds L;
    unsigned scratch0;
    {
        unsigned off = WBYTES;
        L.kp[0] = off; off += (unsigned)npad * 32u;         // K hi
        L.kp[1] = off; off += (unsigned)npad * 32u;         // K lo
        L.kp[2] = 0;
        L.vpitch = (unsigned)npad + 8u;
        L.vh = off; off += 16u * L.vpitch * 2u;
        L.vl = off; off += 16u * L.vpitch * 2u;
        L.kadd = off; off += (unsigned)npad * 4u;
        L.bqg = off; off += 128u;
        scratch0 = off;                                      // per wave: Q hi [32][16] fp16 | Q lo | gate [32][16] fp32 = 4 KB
        L.qp[0] = L.qp[1] = L.qp[2] = L.gl = 0;
    }
    u32x4* Wb = reinterpret_cast<u32x4*>(lds);
    float* kadd = reinterpret_cast<float*>(lds + L.kadd);
    float* bqg = reinterpret_cast<float*>(lds + L.bqg);
    _Float16* Vh = reinterpret_cast<_Float16*>(lds + L.vh);
    _Float16* Vl = reinterpret_cast<_Float16*>(lds + L.vl);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ql = lane & 15, g4 = lane >> 4;
    const int nvb = npad / 32, nqb = (N + 31) / 32;
    const int rstride = gridDim.x / H;
    int h, slot;
    if ((rstride & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        h = idx % H;
        slot = (idx / H) * 8 + xcd;
    } else {
        h = blockIdx.x % H;
        slot = blockIdx.x / H;
    }
    const float sc = 0.25f * LOG2E;
    stage_weight_h2_rows<P>(Wb, 64, 0, wk + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, C, wv + (long)h * C * P, C, P, tid, NT, H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 2 * C, wq + (long)h * C * P, C, P, tid, NT, sc * H2_WSCALE);
    stage_weight_h2_rows<P>(Wb, 64, 3 * C, wg + (long)h * C * P, C, P, tid, NT, NEG_LOG2E * H2_WSCALE);
    if (tid < 32) {
        const int hh = tid >> 4, e = tid & 15;
        bqg[tid] = e < 8 ? 0.f : H2_WSCALE * NEG_LOG2E * bg[h * C + (e < 12 ? 4 * hh + (e - 8) : 8 + 4 * hh + (e - 12))];
    }
    const unsigned qsc = scratch0 + (unsigned)wave * 4096u;          // this wave's scratch: Q hi at +0, Q lo at +1024, gate at +2048
    float* Gs = reinterpret_cast<float*>(lds + qsc + 2048);
    // operand bases of the two QK^T MFMAs: A = [kh | kh], [kl | kl];  B = [qh | ql], [qh | 0]
    const unsigned kbase[2] = {L.kp[0], L.kp[1]};
    const long nrows = (long)b * N;
    auto row_pos = [&](long bu, int v) -> long {
        const long bb = bu / N;
        const long u = bu - bb * N;
        return ending ? ((bb * N + v) * N + u) : (bu * N + v);
    };
    for (long bu = slot; bu < nrows; bu += rstride) {
        const int bb = (int)(bu / N);
        __syncthreads();                        // previous row's K / V fully consumed (and weights staged)
        const float mu = mask[bu];
        // ---- phase 1: K, V of every position of the row ----
        for (int vb = wave; vb < nvb; vb += NW) {
            const int v = vb * 32 + r;
            const bool valid = v < N;
            f32x16 acc[1];
            zero_acc(acc);
            if (vb < nqb) {
                float x[KH];
                load_row_cll<P>(pair + row_pos(bu, valid ? v : 0) * P, hi, valid, x);
                ln_cll<KH>(x);
                u32x4 xs[2][P / 16];
                split2h_cll<KH>(x, xs);
                rowgemm_h2<P, 1>(Wb, 64, 0, xs, acc, r, hi);
            }
            if (hi == 0) {
                const bool keep = valid && (mu * mask[(long)bb * N + (valid ? v : 0)] >= 0.5f);
                kadd[v] = keep ? 0.f : (valid ? -32768.0f * LOG2E : -INFINITY);
            }
            unsigned kh0, kl0, kh1, kl1, kh2, kl2, kh3, kl3;
            split2h(acc[0][0] * H2_INV_WSCALE, acc[0][1] * H2_INV_WSCALE, kh0, kl0);
            split2h(acc[0][2] * H2_INV_WSCALE, acc[0][3] * H2_INV_WSCALE, kh1, kl1);
            split2h(acc[0][4] * H2_INV_WSCALE, acc[0][5] * H2_INV_WSCALE, kh2, kl2);
            split2h(acc[0][6] * H2_INV_WSCALE, acc[0][7] * H2_INV_WSCALE, kh3, kl3);
            const unsigned ko = (unsigned)v * 32u + 8u * hi;
            *reinterpret_cast<u32x2*>(lds + L.kp[0] + ko) = u32x2{kh0, kh1};
            *reinterpret_cast<u32x2*>(lds + L.kp[0] + ko + 16) = u32x2{kh2, kh3};
            *reinterpret_cast<u32x2*>(lds + L.kp[1] + ko) = u32x2{kl0, kl1};
            *reinterpret_cast<u32x2*>(lds + L.kp[1] + ko + 16) = u32x2{kl2, kl3};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const int c0 = (e < 4 ? 4 * hi : 8 + 4 * hi) + (e & 3);
                unsigned hw, lw;
                split2h(acc[0][8 + e], acc[0][9 + e], hw, lw);
                const f16x2 hh2 = __builtin_bit_cast(f16x2, hw), ll2 = __builtin_bit_cast(f16x2, lw);
                Vh[c0 * L.vpitch + v] = (_Float16)hh2[0];
                Vh[(c0 + 1) * L.vpitch + v] = (_Float16)hh2[1];
                Vl[c0 * L.vpitch + v] = (_Float16)ll2[0];
                Vl[(c0 + 1) * L.vpitch + v] = (_Float16)ll2[1];
            }
        }
        __syncthreads();
        // ---- phase 2: per 32-query block: project q | gate into the wave's scratch, then the key loop for its two tiles ----
        const float inv_vs = 1.0f / VSCALE;
        for (int qb_ = wave; qb_ < nqb; qb_ += NW) {
            {
                const int v = qb_ * 32 + r;
                const bool valid = v < N;
                float x[KH];
                load_row_cll<P>(pair + row_pos(bu, valid ? v : 0) * P, hi, valid, x);
                ln_cll<KH>(x);
                u32x4 xs[2][P / 16];
                split2h_cll<KH>(x, xs);
                f32x16 acc[1];
                bias_acc(acc, bqg + 16 * hi);
                rowgemm_h2<P, 1>(Wb, 64, 32, xs, acc, r, hi);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[0][e] *= H2_INV_WSCALE;
                unsigned qh0, ql0, qh1, ql1, qh2, ql2, qh3, ql3;
                split2h(acc[0][0], acc[0][1], qh0, ql0);
                split2h(acc[0][2], acc[0][3], qh1, ql1);
                split2h(acc[0][4], acc[0][5], qh2, ql2);
                split2h(acc[0][6], acc[0][7], qh3, ql3);
                const unsigned qo = qsc + (unsigned)r * 32u + 8u * hi;
                *reinterpret_cast<u32x2*>(lds + qo) = u32x2{qh0, qh1};
                *reinterpret_cast<u32x2*>(lds + qo + 16) = u32x2{qh2, qh3};
                *reinterpret_cast<u32x2*>(lds + qo + 1024) = u32x2{ql0, ql1};
                *reinterpret_cast<u32x2*>(lds + qo + 1024 + 16) = u32x2{ql2, ql3};
                *reinterpret_cast<float4*>(Gs + r * 16 + 4 * hi) = make_float4(gate_from_scaled(acc[0][8]), gate_from_scaled(acc[0][9]),
                                                                                gate_from_scaled(acc[0][10]), gate_from_scaled(acc[0][11]));
                *reinterpret_cast<float4*>(Gs + r * 16 + 8 + 4 * hi) = make_float4(gate_from_scaled(acc[0][12]), gate_from_scaled(acc[0][13]),
                                                                                    gate_from_scaled(acc[0][14]), gate_from_scaled(acc[0][15]));
            }
            wave_lds_fence();
            u32x4 qb[2][2];
            const unsigned half16 = (unsigned)(g4 & 1) * 16u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned qrow = qsc + (unsigned)(16 * t + ql) * 32u + half16;
                const u32x4 qh = *reinterpret_cast<const u32x4*>(lds + qrow), qlo = *reinterpret_cast<const u32x4*>(lds + qrow + 1024);
                qb[t][0] = g4 < 2 ? qh : qlo;                            // [qh | ql]
                qb[t][1] = g4 < 2 ? qh : u32x4{0, 0, 0, 0};              // [qh | 0]
            }
            f32x4 o[2];
            float l_tot[2];
            ta_keyloop_split<2, 2, true>(lds, L, kbase, qb, npad, ql, g4, o, l_tot);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int v = qb_ * 32 + 16 * t + ql;
                if (v < N) {
                    const float4 gf = *reinterpret_cast<const float4*>(Gs + (16 * t + ql) * 16 + 4 * g4);
                    const float il = inv_vs / l_tot[t];
                    *reinterpret_cast<float4*>(og + row_pos(bu, v) * HC + h * C + 4 * g4) =
                        make_float4(gf.x * (o[t][0] * il), gf.y * (o[t][1] * il), gf.z * (o[t][2] * il), gf.w * (o[t][3] * il));
                }
            }
            wave_lds_fence();                   // scratch consumed before the next query block overwrites it
        }
    }
}
#endif  // PRD_AB

// Long-row variant (K/V of the row fill the LDS, no room for Q / gate tiles): queries are re-projected per
// 32-query block in phase 2 and reach the MFMA operand layout through wave shuffles instead of LDS.
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_long_kernel(
    float* __restrict__ og, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int npad, int H, int ending) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                          // [64][P+4]: rows 0-15 k_h, 16-31 v_h, 32-47 q_h, 48-63 g_h
    float* Kl = Wl + 64 * (P + 4);             // [npad][KP]
    float* Vt = Kl + npad * KP;                // [16][npad+4]
    float* kadd = Vt + C * (npad + 4);         // [npad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ql = lane & 15, g4 = lane >> 4;
    const int nvb = npad / 32;
    const int nqb = (N + 31) / 32;
    const int h = blockIdx.x % H;
    const int rstride = gridDim.x / H;
    stage_weight_cll<P>(Wl, wk + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + C * (P + 4), wv + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + 2 * C * (P + 4), wq + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + 3 * C * (P + 4), wg + (long)h * C * P, C, P, tid, NT);
    const float sc = 0.25f * LOG2E;
    for (long bu = blockIdx.x / H; bu < (long)b * N; bu += rstride) {
        const int bb = (int)(bu / N), u = (int)(bu - (long)bb * N);
        __syncthreads();
        const float mu = mask[bu];
        for (int k = tid; k < npad; k += NT) {
            const bool inside = k < N;
            const bool keep = inside && (mu * mask[(long)bb * N + (inside ? k : 0)] >= 0.5f);
            kadd[k] = keep ? 0.f : (inside ? -32768.0f * LOG2E : -INFINITY);
        }
        for (int vb = wave; vb < nvb; vb += NW) {
            const int v = vb * 32 + r;
            const bool valid = v < N;
            const int vv = valid ? v : 0;
            const long pos = ending ? (((long)bb * N + vv) * N + u) : (bu * N + vv);
            f32x16 kv[1];
            zero_acc(kv);
            if (vb < nqb) {
                float x[KH];
                load_row_cll<P>(pair + pos * P, hi, valid, x);
                ln_cll<KH>(x);
                rowgemm<P, 1>(Wl, x, kv, r, hi);
            }
            *reinterpret_cast<float4*>(Kl + v * KP + 4 * hi) = make_float4(kv[0][0], kv[0][1], kv[0][2], kv[0][3]);
            *reinterpret_cast<float4*>(Kl + v * KP + 8 + 4 * hi) = make_float4(kv[0][4], kv[0][5], kv[0][6], kv[0][7]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(4 * hi + e) * (npad + 4) + v] = kv[0][8 + e];
                Vt[(8 + 4 * hi + e) * (npad + 4) + v] = kv[0][12 + e];
            }
        }
        __syncthreads();
        for (int qb = wave; qb < nqb; qb += NW) {
            float qg[16];
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
                rowgemm<P, 1>(Wl + 2 * C * (P + 4), x, acc, r, hi);
#pragma unroll
                for (int i = 0; i < 8; ++i) qg[i] = acc[0][i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qg[8 + e] = acc[0][8 + e] + bg[h * C + 4 * hi + e];
                    qg[12 + e] = acc[0][12 + e] + bg[h * C + 8 + 4 * hi + e];
                }
            }
            // lane (r, hi) holds q channels {4hi+e} in qg[0..3], {8+4hi+e} in qg[4..7], gate likewise in qg[8..15];
            // lane (ql, g4) of query tile t needs channels 4*g4+e of position 16t+ql = lane 16t+ql+32*(g4&1), half g4>>1
            float4 qf[2], gf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int src = 16 * t + ql + 32 * (g4 & 1);
                float lo[4], up[4], glo[4], gup[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __shfl(qg[e], src);
                    up[e] = __shfl(qg[4 + e], src);
                    glo[e] = __shfl(qg[8 + e], src);
                    gup[e] = __shfl(qg[12 + e], src);
                }
                const bool hiq = (g4 >> 1) != 0;
                qf[t] = make_float4(sc * (hiq ? up[0] : lo[0]), sc * (hiq ? up[1] : lo[1]), sc * (hiq ? up[2] : lo[2]), sc * (hiq ? up[3] : lo[3]));
                gf[t] = make_float4(sigmoid_fast(hiq ? gup[0] : glo[0]), sigmoid_fast(hiq ? gup[1] : glo[1]),
                                    sigmoid_fast(hiq ? gup[2] : glo[2]), sigmoid_fast(hiq ? gup[3] : glo[3]));
            }
            f32x4 o[2];
            float l_tot[2];
            ta_keyloop<2, true>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int v = qb * 32 + 16 * t + ql;
                if (v < N) {
                    const long pos = ending ? (((long)bb * N + v) * N + u) : (bu * N + v);
                    *reinterpret_cast<float4*>(og + pos * HC + h * C + 4 * g4) =
                        make_float4(gf[t].x * (o[t][0] / l_tot[t]), gf[t].y * (o[t][1] / l_tot[t]),
                                    gf[t].z * (o[t][2] / l_tot[t]), gf[t].w * (o[t][3] / l_tot[t]));
                }
            }
        }
    }
}

// Rows of any length (N > 960: K / V of a whole row no longer fit the LDS): the long-row kernel over ONE contiguous chunk of
// keys [key0, key0 + klen).  The chunks of a row are processed by consecutive launches; every launch attends all queries of
// the row to its chunk and merges with what the earlier chunks left: og holds the gated, normalised output so far, stats
// [b][N][N][H][2] the (reference m in the log2 domain, sum l relative to it) of the softmax so far.  With w_c = l_c 2^(m_c - M),
// M = max m_c:  og = sum_c w_c og_c / sum_c w_c -- the softmax over the union of the chunks (modules.py:216-223), exact up to
// fp32 rounding.  fp32 MFMA arithmetic in either mode.
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void tri_attn_core_chunk_kernel(
    float* __restrict__ og, float* __restrict__ stats, const float* __restrict__ pair, const float* __restrict__ mask,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ wg, const float* __restrict__ bg, int b, int N, int key0, int klen, int first, int H, int ending) {
    constexpr int C = 16, HC = 64, NT = NW * 64, KH = P / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npad = (klen + 63) / 64 * 64;    // keys of this chunk, padded
    float* Wl = smem;                          // [64][P+4]: rows 0-15 k_h, 16-31 v_h, 32-47 q_h, 48-63 g_h
    float* Kl = Wl + 64 * (P + 4);             // [npad][KP]
    float* Vt = Kl + npad * KP;                // [16][npad+4]
    float* kadd = Vt + C * (npad + 4);         // [npad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hi = lane >> 5;
    const int ql = lane & 15, g4 = lane >> 4;
    const int nvb = npad / 32;
    const int nqb = (N + 31) / 32;
    const int h = blockIdx.x % H;
    const int rstride = gridDim.x / H;
    stage_weight_cll<P>(Wl, wk + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + C * (P + 4), wv + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + 2 * C * (P + 4), wq + (long)h * C * P, C, P, tid, NT);
    stage_weight_cll<P>(Wl + 3 * C * (P + 4), wg + (long)h * C * P, C, P, tid, NT);
    const float sc = 0.25f * LOG2E;
    for (long bu = blockIdx.x / H; bu < (long)b * N; bu += rstride) {
        const int bb = (int)(bu / N), u = (int)(bu - (long)bb * N);
        __syncthreads();
        const float mu = mask[bu];
        for (int k = tid; k < npad; k += NT) {
            const bool inside = k < klen;
            const bool keep = inside && (mu * mask[(long)bb * N + (inside ? key0 + k : 0)] >= 0.5f);
            kadd[k] = keep ? 0.f : (inside ? -32768.0f * LOG2E : -INFINITY);
        }
        for (int vb = wave; vb < nvb; vb += NW) {
            const int kk = vb * 32 + r;                  // key index inside the chunk
            const bool valid = kk < klen;
            const int vv = valid ? key0 + kk : 0;
            const long pos = ending ? (((long)bb * N + vv) * N + u) : (bu * N + vv);
            f32x16 kv[1];
            zero_acc(kv);
            if (vb * 32 < klen) {
                float x[KH];
                load_row_cll<P>(pair + pos * P, hi, valid, x);
                ln_cll<KH>(x);
                rowgemm<P, 1>(Wl, x, kv, r, hi);
            }
            *reinterpret_cast<float4*>(Kl + kk * KP + 4 * hi) = make_float4(kv[0][0], kv[0][1], kv[0][2], kv[0][3]);
            *reinterpret_cast<float4*>(Kl + kk * KP + 8 + 4 * hi) = make_float4(kv[0][4], kv[0][5], kv[0][6], kv[0][7]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(4 * hi + e) * (npad + 4) + kk] = kv[0][8 + e];
                Vt[(8 + 4 * hi + e) * (npad + 4) + kk] = kv[0][12 + e];
            }
        }
        __syncthreads();
        for (int qb = wave; qb < nqb; qb += NW) {
            float qg[16];
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
                rowgemm<P, 1>(Wl + 2 * C * (P + 4), x, acc, r, hi);
#pragma unroll
                for (int i = 0; i < 8; ++i) qg[i] = acc[0][i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qg[8 + e] = acc[0][8 + e] + bg[h * C + 4 * hi + e];
                    qg[12 + e] = acc[0][12 + e] + bg[h * C + 8 + 4 * hi + e];
                }
            }
            float4 qf[2], gf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {              // same redistribution as tri_attn_core_long_kernel
                const int src = 16 * t + ql + 32 * (g4 & 1);
                float lo[4], up[4], glo[4], gup[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __shfl(qg[e], src);
                    up[e] = __shfl(qg[4 + e], src);
                    glo[e] = __shfl(qg[8 + e], src);
                    gup[e] = __shfl(qg[12 + e], src);
                }
                const bool hiq = (g4 >> 1) != 0;
                qf[t] = make_float4(sc * (hiq ? up[0] : lo[0]), sc * (hiq ? up[1] : lo[1]), sc * (hiq ? up[2] : lo[2]), sc * (hiq ? up[3] : lo[3]));
                gf[t] = make_float4(sigmoid_fast(hiq ? gup[0] : glo[0]), sigmoid_fast(hiq ? gup[1] : glo[1]),
                                    sigmoid_fast(hiq ? gup[2] : glo[2]), sigmoid_fast(hiq ? gup[3] : glo[3]));
            }
            f32x4 o[2];
            float l_tot[2], m_ref[2];
            ta_keyloop<2, true>(Kl, Vt, kadd, qf, npad, ql, g4, o, l_tot, m_ref);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int v = qb * 32 + 16 * t + ql;
                if (v < N) {
                    const long pos = ending ? (((long)bb * N + v) * N + u) : (bu * N + v);
                    float4* dst = reinterpret_cast<float4*>(og + pos * HC + h * C + 4 * g4);
                    float* st = stats + (pos * H + h) * 2;
                    const float il = 1.0f / l_tot[t];
                    float4 mine = make_float4(gf[t].x * (o[t][0] * il), gf[t].y * (o[t][1] * il), gf[t].z * (o[t][2] * il), gf[t].w * (o[t][3] * il));
                    float m_new = m_ref[t], l_new = l_tot[t];
                    if (!first) {
                        const float m0 = st[0], l0 = st[1];
                        const float4 prev = *dst;
                        const float M = fmaxf(m0, m_ref[t]);
                        const float w0 = l0 * exp2f(m0 - M), w1 = l_tot[t] * exp2f(m_ref[t] - M);
                        const float iw = 1.0f / (w0 + w1);
                        const float a0 = w0 * iw, a1 = w1 * iw;
                        mine = make_float4(a0 * prev.x + a1 * mine.x, a0 * prev.y + a1 * mine.y, a0 * prev.z + a1 * mine.z, a0 * prev.w + a1 * mine.w);
                        m_new = M;
                        l_new = w0 + w1;
                    }
                    *dst = mine;
                    // the four lanes (g4) of a query read the statistics above before lane g4 = 0 replaces them: one wave, program order
                    if (g4 == 0) { st[0] = m_new; st[1] = l_new; }
                }
            }
        }
    }
}

template <int P, int NW, bool B3>
__global__ __launch_bounds__(NW * 64) void tri_attn_out_kernel(int* queue, float* out, const float* pair, const float* __restrict__ og,
                                                               const float* __restrict__ wo, const float* __restrict__ bo, long rows, int residual) {
    constexpr int KH = P / 2, NB = P / 32, HC = 64;
    __shared__ __attribute__((aligned(16))) float Wl[B3 ? P * HC : P * (HC + 4)];      // B3: fp16 hi | lo planes (rowgemm_h2)
    __shared__ __attribute__((aligned(16))) float bl[P];
    if (B3) stage_weight_h2<HC>(reinterpret_cast<u32x4*>(Wl), wo, P, HC, threadIdx.x, NW * 64, H2_WSCALE);
    else stage_weight_cll<HC>(Wl, wo, P, HC, threadIdx.x, NW * 64);
    stage_vec_cll(bl, bo, P, threadIdx.x, NW * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    const long ntask = (rows + 31) / 32;
    WaveTasks tasks(queue, ntask, NW);
    for (long task = tasks.next(); task >= 0; task = tasks.next()) {
        const long pos = task * 32 + r;
        const bool valid = pos < rows;
        float x[HC / 2];
        load_row_cll<HC>(og + pos * HC, hi, valid, x);
        f32x16 acc[NB];
        zero_acc(acc);
        if (B3) {
            u32x4 xs[2][HC / 16];
            split2h_cll<HC / 2>(x, xs);
            rowgemm_h2<HC, NB>(reinterpret_cast<const u32x4*>(Wl), P, 0, xs, acc, r, hi);
        } else {
            rowgemm<HC, NB>(Wl, x, acc, r, hi);
        }
        float pr[KH];
        load_row_cll<P>(pair + pos * P, hi, valid && residual, pr);
#pragma unroll
        for (int s = 0; s < KH; ++s) pr[s] = pr[s] + (acc[s >> 4][s & 15] * (B3 ? H2_INV_WSCALE : 1.0f) + bl[hi * KH + s]);
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

// the split contraction by the A/B switches of the caller (waves per workgroup; chunks of operands in flight)
static void launch_contract_split(int tune, int grid, size_t lds3, hipStream_t stream, float* O, const float* AB, int N, int ldn, int P, int nbatch,
                                  int tiles, int swap) {
    const int nw = PRD_TGET_TMS_NW(tune);
#ifdef PRD_AB       // (libprd_hip_ab.so) three chunks in flight / 16 waves: measured in rounds 3 and 5, not faster
    if (nw == 8 && PRD_TGET_TMS_D3(tune)) {
        PRD_SET_LDS((tri_mul_contract_split_kernel<8, 3>), lds3);
        hipLaunchKernelGGL((tri_mul_contract_split_kernel<8, 3>), dim3(grid), dim3(512), lds3, stream, O, AB, N, ldn, P, nbatch, tiles, swap);
        return;
    }
    if (nw == 16) {
        PRD_SET_LDS((tri_mul_contract_split_kernel<16>), lds3);
        hipLaunchKernelGGL((tri_mul_contract_split_kernel<16>), dim3(grid), dim3(1024), lds3, stream, O, AB, N, ldn, P, nbatch, tiles, swap);
        return;
    }
#endif
    if (nw == 12) {     // (kept in the shipped library: the second arm of the direct parity test -- same results bit for bit)
        PRD_SET_LDS((tri_mul_contract_split_kernel<12>), lds3);
        hipLaunchKernelGGL((tri_mul_contract_split_kernel<12>), dim3(grid), dim3(768), lds3, stream, O, AB, N, ldn, P, nbatch, tiles, swap);
    } else {
        PRD_SET_LDS((tri_mul_contract_split_kernel<8>), lds3);
        hipLaunchKernelGGL((tri_mul_contract_split_kernel<8>), dim3(grid), dim3(512), lds3, stream, O, AB, N, ldn, P, nbatch, tiles, swap);
    }
}


#ifdef PRD_AB
constexpr bool PRD_FIRST_GEN = true;
#else
constexpr bool PRD_FIRST_GEN = false;       // first-generation split-16 cores compiled out: see the PRD_AB note above them
#endif
namespace {
// LDS bytes of the triangle-attention core for rows of N positions; long_row: the re-projecting variant is needed
size_t tri_attn_lds(int N, int P, bool b3, bool* long_row) {
    const int npad = prd_round_up(N, 64);
    const size_t wsz = b3 ? (size_t)3 * 64 * (2 * (P / 16) + 1) * 4 : (size_t)64 * (P + 4);
    size_t lds = (wsz + (size_t)3 * npad * KP + 16 * (npad + 4) + npad + 32) * sizeof(float);
    // split-operand kernel (gemm mode 1): K / Q planes 6 x 32 B, V hi / lo 2 x 16 x (npad + 8) fp16, gate, override, bias
    if (b3) lds = (size_t)64 * P * 4 + (size_t)npad * (192 + KP * 4 + 4) + (size_t)64 * (npad + 8) + 128;
    *long_row = lds > 160 * 1024;              // Q / gate tiles do not fit next to the row's K / V
    if (*long_row) {
        // split-operand long rows: weights + K hi|lo (64 B) + V hi|lo + override per position + bias + 8 waves x 4 KB of scratch
        const size_t lds_sl = (size_t)64 * P * 4 + (size_t)npad * 68 + (size_t)64 * (npad + 8) + 128 + 8 * 4096;
        if (b3 && lds_sl <= 160 * 1024) return lds_sl;
        lds = ((size_t)64 * (P + 4) + (size_t)npad * KP + 16 * (npad + 4) + npad) * sizeof(float);
    }
    return lds;
}
}  // namespace

extern "C" int prd_tri_attn_variant(int N, int P, int arith) {
    PRD_SPLIT_ARITH(arith);
    if (N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    bool long_row;
    const bool split = arith == PRD_ARITH_SPLIT16;
    const bool v2 = split && PRD_TGET_TA_VARIANT(tune) == 0 && prd_tri_attn_v2_supported(N, P, tune);   // what prd_tri_attn_core dispatches to first
    const bool b3 = split && (PRD_FIRST_GEN || v2);   // without the first generation, rows the second one does not serve run the fp32 kernels
    const size_t lds = tri_attn_lds(N, P, b3, &long_row);
    if (lds > 160 * 1024)                      // the round-3 core keeps K / V as fp16 planes: rows up to 1024; beyond: key-chunked
        return v2 ? 2 : 3;
    if (!long_row) return 0;
    if (v2) return 2;                          // long rows on the round-3 core (also where the fp32 long-row kernel would fit)
    const int npad = prd_round_up(N, 64);
    return (b3 && lds == (size_t)64 * P * 4 + (size_t)npad * 68 + (size_t)64 * (npad + 8) + 128 + 8 * 4096) ? 2 : 1;
}

namespace {
constexpr int TA_CHUNK_MAX = 960;              // keys per chunk: (64 (P+4) + 148 npad + 64) floats... <= 160 KB for P = 64
int ta_chunks(int N) { return prd_ceil_div(N, TA_CHUNK_MAX); }
}  // namespace

extern "C" size_t prd_tri_attn_stats_bytes(int b, int N, int P, int H, int arith) {
    if (b <= 0 || N <= 0 || H <= 0) return 0;
    return prd_tri_attn_variant(N, P, arith) == 3 ? (size_t)b * N * N * H * 2 * sizeof(float) : 0;
}

extern "C" int prd_tri_attn_core_chunked(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                                         const float* wv, const float* wg, const float* bg, int ending,
                                         int b, int N, int P, int H, int c, float* stats, size_t stats_bytes, hipStream_t stream) {
    if (!og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || !stats || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (stats_bytes < (size_t)b * N * N * H * 2 * sizeof(float)) return PRD_ERR_WORKSPACE;
    const int nchunk = ta_chunks(N);
    const int per = prd_round_up(prd_ceil_div(N, nchunk), 64);         // keys per chunk (the last one may be shorter)
    const size_t lds = ((size_t)64 * (P + 4) + (size_t)per * KP + 16 * (per + 4) + per) * sizeof(float);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const long rows_total = (long)b * N;
    const long cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    const int grid = (int)(per_head * H);
    for (int ck = 0; ck < nchunk; ++ck) {
        const int key0 = ck * per;
        const int klen = key0 + per <= N ? per : N - key0;
        if (klen <= 0) break;
        if (P == 64) {
            PRD_SET_LDS((tri_attn_core_chunk_kernel<64, 8>), lds);
            hipLaunchKernelGGL((tri_attn_core_chunk_kernel<64, 8>), dim3(grid), dim3(512), lds, stream, og, stats, pair, mask, wq, wk, wv, wg, bg,
                               b, N, key0, klen, ck == 0 ? 1 : 0, H, ending);
        } else {
            PRD_SET_LDS((tri_attn_core_chunk_kernel<32, 8>), lds);
            hipLaunchKernelGGL((tri_attn_core_chunk_kernel<32, 8>), dim3(grid), dim3(512), lds, stream, og, stats, pair, mask, wq, wk, wv, wg, bg,
                               b, N, key0, klen, ck == 0 ? 1 : 0, H, ending);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" size_t prd_workspace_bytes(const char* op, int b, int N, int S, int P) {
    (void)S;
    if (!op || b <= 0 || N <= 0) return 0;
    const size_t ldn = (size_t)prd_round_up(N, 32);
    if (op[0] == 't' && op[4] == 'm') return (size_t)3 * b * P * N * ldn * sizeof(float);   // "tri_mul": operands a | b + output
    if (op[0] == 't' && op[4] == 'a')        // "tri_attn": og, and for key-chunked rows the softmax statistics (either arithmetic)
        return (size_t)b * N * N * 64 * sizeof(float) + prd_tri_attn_stats_bytes(b, N, P, 4, PRD_ARITH_FP32);
    return 0;
}

extern "C" int prd_tri_mul(float* out, const float* pair, const float* mask, const float* w_proj, const float* b_proj,
                           const float* w_gate, const float* b_gate, const float* w_out, const float* b_out,
                           const float* w_ogate, const float* b_ogate, int incoming, int residual,
                           int b, int N, int P, float* ws, size_t ws_bytes, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!out || !pair || !mask || !w_proj || !b_proj || !w_gate || !b_gate || !w_out || !b_out || !w_ogate || !b_ogate || !ws ||
        b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    if (ws_bytes < prd_workspace_bytes("tri_mul", b, N, 0, P)) return PRD_ERR_WORKSPACE;
    const int ldn = prd_round_up(N, 32);
    float* AB = ws;                                   // [b][2P][N][ldn]
    float* O = ws + (size_t)2 * b * P * N * ldn;      // [b][P][N][ldn]
    const bool b3m = arith == PRD_ARITH_SPLIT16;
    {
        constexpr int NWP = 16;                      // one persistent 16-wave workgroup per CU (4 waves / SIMD)
        const bool b3 = b3m;                        // bf16 x 3 row GEMM (prd_set_gemm_mode)
        const size_t wsz = b3 ? (size_t)2 * P * P : (size_t)2 * P * (P + 4);
        const size_t lds = (2 * wsz + 4 * P) * sizeof(float);
        const long ntask = ((long)b * N * (ldn / 32) + 7) / 8 * 8 * (2 * P / 32);     // (row block, output block) tasks
        const int grid = grid_for(ntask, 4, 256);
#define PRD_PROJ_LAUNCH(PP, BB, NWV)                                                                                 \
        do {                                                                                                         \
            PRD_SET_LDS((tri_mul_proj_kernel<PP, NWV, BB>), lds);                                                    \
            hipLaunchKernelGGL((tri_mul_proj_kernel<PP, NWV, BB>), dim3(grid), dim3(NWV * 64), lds, stream, queue, AB, pair, mask, \
                               w_proj, b_proj, w_gate, b_gate, b, N, ldn, incoming);                                 \
        } while (0)
        // (the split operands of the bf16 x 3 form need the registers of a 12-wave workgroup)
        if (P == 64) { if (b3) { if (PRD_TGET_TMP_NW16(tune)) PRD_PROJ_LAUNCH(64, true, 16); else PRD_PROJ_LAUNCH(64, true, 12); } else PRD_PROJ_LAUNCH(64, false, NWP); }
        else { if (b3) PRD_PROJ_LAUNCH(32, true, 12); else PRD_PROJ_LAUNCH(32, false, NWP); }
#undef PRD_PROJ_LAUNCH
        int e = (int)hipGetLastError();
        if (e) return e;
    }
    {
        const int tiles = prd_ceil_div(N, 64);
        const int vblocks = b * P * tiles * tiles;
        if (b3m) {
            const int tl = prd_ceil_div(N, TMS_T);
            const int vb3 = b * P * tl * tl;
            const size_t lds3 = (size_t)4 * TMS_OPER;
        launch_contract_split(tune, vb3 < 256 ? vb3 : 256, lds3, stream, O, AB, N, ldn, P, b, tl, 0);
        }
        else
            hipLaunchKernelGGL(tri_mul_contract_kernel, dim3(vblocks < 1024 ? vblocks : 1024), dim3(256), 0, stream, O, AB, N, ldn, P, b, tiles);
        int e = (int)hipGetLastError();
        if (e) return e;
    }
    {
        constexpr int NWO = 8;
        const bool b3o = b3m;
        const long ntask = (long)b * N * prd_ceil_div(N, 32);
        const int grid = grid_for(ntask, 4, 256);
#define PRD_OUT_LAUNCH(PP, BB)                                                                                         \
        hipLaunchKernelGGL((tri_mul_out_kernel<PP, NWO, BB>), dim3(grid), dim3(NWO * 64), 0, stream, queue, out, pair, O, w_out, \
                           b_out, w_ogate, b_ogate, b, N, ldn, residual)
        if (P == 64) { if (b3o) PRD_OUT_LAUNCH(64, true); else PRD_OUT_LAUNCH(64, false); }
        else { if (b3o) PRD_OUT_LAUNCH(32, true); else PRD_OUT_LAUNCH(32, false); }
#undef PRD_OUT_LAUNCH
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_contract(float* O, const float* AB, int b, int N, int P, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!O || !AB || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64 && P != 128) return PRD_ERR_UNSUPPORTED;     // 2 x 64: the stacked gradient contractions of the backward
    const int ldn = prd_round_up(N, 32);
    if (arith == PRD_ARITH_SPLIT16) {
        const int tl = prd_ceil_div(N, TMS_T);
        const int vb3 = b * P * tl * tl;
        const size_t lds3 = (size_t)4 * TMS_OPER;
        launch_contract_split(tune, vb3 < 256 ? vb3 : 256, lds3, stream, O, AB, N, ldn, P, b, tl, 0);
    } else {
        const int tiles = prd_ceil_div(N, 64);
        const int vblocks = b * P * tiles * tiles;
        hipLaunchKernelGGL(tri_mul_contract_kernel, dim3(vblocks < 1024 ? vblocks : 1024), dim3(256), 0, stream, O, AB, N, ldn, P, b, tiles);
    }
    return (int)hipGetLastError();
}

extern "C" int prd_tri_mul_chain_supported(int N, int P, int arith) {
    return (N > 0 && (P == 32 || P == 64) && arith >= 0 && (arith & 0xff) == PRD_ARITH_SPLIT16) ? 1 : 0;
}

extern "C" int prd_tri_mul_chain(float* pair, const float* mask, const float* const* w_outgoing, const float* const* w_incoming,
                                 int b, int N, int P, float* ws, size_t ws_bytes, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!pair || !mask || !w_outgoing || !w_incoming || !ws || b <= 0 || N <= 0) return PRD_ERR_ARG;
    for (int k = 0; k < 8; ++k)
        if (!w_outgoing[k] || !w_incoming[k]) return PRD_ERR_ARG;
    if (!prd_tri_mul_chain_supported(N, P, arith)) return PRD_ERR_UNSUPPORTED;
    if (ws_bytes < prd_workspace_bytes("tri_mul", b, N, 0, P)) return PRD_ERR_WORKSPACE;
    const int ldn = prd_round_up(N, 32);
    float* AB = ws;                                   // [b][2P][N][ldn]
    float* O = ws + (size_t)2 * b * P * N * ldn;      // [b][P][N][ldn]
    const float* const* wa = w_outgoing;              // proj w, b | gate w, b | out w, b | out-gate w, b
    const float* const* wb = w_incoming;
    const size_t ldsp = ((size_t)2 * 2 * P * P + 4 * P) * sizeof(float);
    const long ptask = ((long)b * N * (ldn / 32) + 7) / 8 * 8 * (2 * P / 32);
    const int pgrid = grid_for(ptask, 4, 256);
    const int tl = prd_ceil_div(N, TMS_T);
    const int vb3 = b * P * tl * tl;
    const size_t lds3 = (size_t)4 * TMS_OPER;
    const long rtask = (long)b * N * (ldn / 32);
    const int rgrid = grid_for(rtask, 4, 256);
    const size_t ldsf = ((size_t)2 * P * P + (size_t)2 * 2 * P * P + 6 * P) * sizeof(float);
    // a failed launch must not let the later stages run over a half-written workspace: checked after every stage
#define PRD_CHAIN_STAGE_OK() do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
    // 1. a | b of the outgoing module
    if (P == 64 && PRD_TGET_TMP_NW16(tune)) {           // (A/B: PRD_TUNE_TMP_NW16)
        PRD_SET_LDS((tri_mul_proj_kernel<64, 16, true>), ldsp);
        hipLaunchKernelGGL((tri_mul_proj_kernel<64, 16, true>), dim3(pgrid), dim3(16 * 64), ldsp, stream, (int*)nullptr, AB, pair, mask,
                           wa[0], wa[1], wa[2], wa[3], b, N, ldn, 0);
    } else if (P == 64) {
        PRD_SET_LDS((tri_mul_proj_kernel<64, 12, true>), ldsp);
        hipLaunchKernelGGL((tri_mul_proj_kernel<64, 12, true>), dim3(pgrid), dim3(12 * 64), ldsp, stream, (int*)nullptr, AB, pair, mask,
                           wa[0], wa[1], wa[2], wa[3], b, N, ldn, 0);
    } else {
        PRD_SET_LDS((tri_mul_proj_kernel<32, 12, true>), ldsp);
        hipLaunchKernelGGL((tri_mul_proj_kernel<32, 12, true>), dim3(pgrid), dim3(12 * 64), ldsp, stream, (int*)nullptr, AB, pair, mask,
                           wa[0], wa[1], wa[2], wa[3], b, N, ldn, 0);
    }
    PRD_CHAIN_STAGE_OK();
    // 2. its contraction, transposed: O^T[c][j][i]
        launch_contract_split(tune, vb3 < 256 ? vb3 : 256, lds3, stream, O, AB, N, ldn, P, b, tl, 1);
    PRD_CHAIN_STAGE_OK();
    // 3. output stage of the outgoing module + a | b of the incoming one.  P = 64: 12 waves (168 VGPRs, three per SIMD) cover the
    // task's latency chain better than 8 (36.6 -> 31.7 us at N = 320; the plain output stage below needs 192 VGPRs and stays at 8:
    // 12 waves spill, 19.9 -> 28.7 us)
    if (P == 64) {
        PRD_SET_LDS((tri_mul_out_proj_kernel<64, 12>), ldsf);
        hipLaunchKernelGGL((tri_mul_out_proj_kernel<64, 12>), dim3(rgrid), dim3(12 * 64), ldsf, stream, pair, O, mask, wa[4], wa[5], wa[6], wa[7],
                           wb[0], wb[1], wb[2], wb[3], AB, b, N, ldn);
    } else {
        PRD_SET_LDS((tri_mul_out_proj_kernel<32, 8>), ldsf);
        hipLaunchKernelGGL((tri_mul_out_proj_kernel<32, 8>), dim3(rgrid), dim3(8 * 64), ldsf, stream, pair, O, mask, wa[4], wa[5], wa[6], wa[7],
                           wb[0], wb[1], wb[2], wb[3], AB, b, N, ldn);
    }
    PRD_CHAIN_STAGE_OK();
    // 4. contraction of the incoming module
        launch_contract_split(tune, vb3 < 256 ? vb3 : 256, lds3, stream, O, AB, N, ldn, P, b, tl, 0);
    PRD_CHAIN_STAGE_OK();
    // 5. its output stage
    {
        const long ntask = (long)b * N * prd_ceil_div(N, 32);
        const int grid = grid_for(ntask, 4, 256);
        if (P == 64) hipLaunchKernelGGL((tri_mul_out_kernel<64, 8, true>), dim3(grid), dim3(8 * 64), 0, stream, (int*)nullptr, pair, pair, O,
                                        wb[4], wb[5], wb[6], wb[7], b, N, ldn, 1);
        else hipLaunchKernelGGL((tri_mul_out_kernel<32, 8, true>), dim3(grid), dim3(8 * 64), 0, stream, (int*)nullptr, pair, pair, O,
                                wb[4], wb[5], wb[6], wb[7], b, N, ldn, 1);
    }
#undef PRD_CHAIN_STAGE_OK
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn_core(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                                 const float* wv, const float* wg, const float* bg, int ending,
                                 int b, int N, int P, int H, int c, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!og || !pair || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    const int npad = prd_round_up(N, 64);
    const int nqb = prd_ceil_div(N, 32);
    const bool split = arith == PRD_ARITH_SPLIT16;    // split 16-bit operands
    const int variant = PRD_FIRST_GEN ? PRD_TGET_TA_VARIANT(tune) : 0;    // A/B switch: first-generation kernels (-DPRD_AB builds only)
    // second generation (prd_tri2.hip): short rows, and long rows as far as K / V of a row fit the LDS as fp16 planes
    if (split && variant == 0 && prd_tri_attn_v2_supported(N, P, tune) && (long)b * N * N <= 0x7fffffffL / 2)
        return prd_tri_attn_core_v2(og, pair, mask, wq, wk, wv, wg, bg, ending, b, N, P, H, c, tune, stream);
    const bool b3 = split && PRD_FIRST_GEN;           // otherwise: the fp32-MFMA kernels below (more accurate, slower)
    bool long_row;
    const size_t lds = tri_attn_lds(N, P, b3, &long_row);
    if (lds > 160 * 1024) return PRD_ERR_UNSUPPORTED;
    const int nw = 8;                                   // 2 waves / SIMD: room for the next-row prefetch registers
    // persistent workgroups (weights staged once per workgroup), one per CU: per head the SMALLEST workgroup
    // count that reaches the minimum number of row rounds, so every workgroup walks the same number of rows
    const long rows_total = (long)b * N;
    const long cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head < 1) per_head = 1;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    const int grid = (int)(per_head * H);
#define PRD_TA_LAUNCH(KERNEL, NWV, ...)                                                                                \
    do {                                                                                                               \
        PRD_SET_LDS((KERNEL<__VA_ARGS__>), lds);                                                                       \
        hipLaunchKernelGGL((KERNEL<__VA_ARGS__>), dim3(grid), dim3(NWV * 64), lds, stream, og, pair, mask, wq, wk, wv, wg, bg, b, N, npad, H, ending); \
    } while (0)
    (void)nqb; (void)nw;
    // 12 waves (3 per SIMD) + next-row prefetch: the ceil(N/16) query tiles dealt in pairs land 5 per SIMD at N = 320
    // (measured: 12 waves + prefetch 142 us, 16 waves without prefetch 149 us, 8 waves + prefetch 146 us)
    const bool split_long = long_row && b3 &&
        lds == (size_t)64 * P * 4 + (size_t)npad * 68 + (size_t)64 * (npad + 8) + 128 + 8 * 4096;      // tri_attn_lds chose it
#ifdef PRD_AB
    if (split_long) { if (P == 64) PRD_TA_LAUNCH(tri_attn_core_split_long_kernel, 8, 64, 8); else PRD_TA_LAUNCH(tri_attn_core_split_long_kernel, 8, 32, 8); }
    else if (!long_row && b3) {
        if (P == 64) {
            if (variant == 1) PRD_TA_LAUNCH(tri_attn_core_split_kernel, 16, 64, 16, 1, false);
            else if (variant == 2) PRD_TA_LAUNCH(tri_attn_core_split_kernel, 12, 64, 12, 1, false);
            else if (variant == 3) PRD_TA_LAUNCH(tri_attn_core_split_kernel, 8, 64, 8, 1, true);
            else PRD_TA_LAUNCH(tri_attn_core_split_kernel, 8, 64, 8, 2, true);
        } else {
            PRD_TA_LAUNCH(tri_attn_core_split_kernel, 8, 32, 8, 2, true);
        }
    } else
#else
    (void)split_long; (void)variant;
#endif
    if (long_row) { if (P == 64) PRD_TA_LAUNCH(tri_attn_core_long_kernel, 8, 64, 8); else PRD_TA_LAUNCH(tri_attn_core_long_kernel, 8, 32, 8); }
    else { if (P == 64) PRD_TA_LAUNCH(tri_attn_core_kernel, 12, 64, 12, true, false); else PRD_TA_LAUNCH(tri_attn_core_kernel, 12, 32, 12, true, false); }
#undef PRD_TA_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int prd_single_attn_core(float* o, const float* qkvg, int ldq, const float* bias, const float* mask,
                                    int b, int N, int H, int c, hipStream_t stream) {
    if (!o || !qkvg || !bias || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (ldq < 4 * H * c || (ldq & 3)) return PRD_ERR_ALIGN;
    hipLaunchKernelGGL(single_attn_core_kernel, dim3(b * H * prd_ceil_div(N, 16)), dim3(256), 0, stream, o, qkvg, bias, mask, b, N, H, ldq);
    return (int)hipGetLastError();
}

// LDS of the fused form: the short-row split kernel's layout + the W_o image and bias of the previous attention
static size_t tri_attn_fused_lds(int N, int P) {
    bool long_row;
    const size_t lds = tri_attn_lds(N, P, true, &long_row);
    if (long_row) return (size_t)1 << 30;
    return lds + (size_t)P * 4 + (size_t)P * 256;
}

extern "C" int prd_tri_attn_core_fused_supported(int N, int P, int arith) {
    // (the fused form lives on the first-generation core: -DPRD_AB builds only; measured slower than two launches)
    return (PRD_FIRST_GEN && N > 0 && (P == 32 || P == 64) && arith >= 0 && (arith & 0xff) == PRD_ARITH_SPLIT16 &&
            tri_attn_fused_lds(N, P) <= 160 * 1024) ? 1 : 0;
}

extern "C" int prd_tri_attn_core_fused(float* og, float* pair_out, const float* pair, const float* og_in, const float* wo_in,
                                       const float* bo_in, const float* mask, const float* wq, const float* wk, const float* wv,
                                       const float* wg, const float* bg, int ending, int b, int N, int P, int H, int c,
                                       hipStream_t stream) {
    if (!og || !pair_out || !pair || !og_in || !wo_in || !bo_in || !mask || !wq || !wk || !wv || !wg || !bg || b <= 0 || N <= 0 ||
        pair_out == pair) return PRD_ERR_ARG;
    if ((P != 32 && P != 64) || c != 16 || H * c != 64) return PRD_ERR_UNSUPPORTED;
    if (!prd_tri_attn_core_fused_supported(N, P, PRD_ARITH_SPLIT16)) return PRD_ERR_UNSUPPORTED;
    const int npad = prd_round_up(N, 64);
    const size_t lds = tri_attn_fused_lds(N, P);
    const long rows_total = (long)b * N;
    const long cap = 256 / H;
    long per_head = cap < rows_total ? cap : rows_total;
    if (per_head < 1) per_head = 1;
    const long rounds = (rows_total + per_head - 1) / per_head;
    per_head = (rows_total + rounds - 1) / rounds;
    const int grid = (int)(per_head * H);
#ifndef PRD_AB
    (void)grid; (void)lds; (void)npad;
    return PRD_ERR_UNSUPPORTED;
#else
    if (P == 64) {
        PRD_SET_LDS((tri_attn_core_split_kernel<64, 8, 2, true, true>), lds);
        hipLaunchKernelGGL((tri_attn_core_split_kernel<64, 8, 2, true, true>), dim3(grid), dim3(512), lds, stream, og, pair, mask, wq, wk, wv, wg,
                           bg, b, N, npad, H, ending, og_in, wo_in, bo_in, pair_out);
    } else {
        PRD_SET_LDS((tri_attn_core_split_kernel<32, 8, 2, true, true>), lds);
        hipLaunchKernelGGL((tri_attn_core_split_kernel<32, 8, 2, true, true>), dim3(grid), dim3(512), lds, stream, og, pair, mask, wq, wk, wv, wg,
                           bg, b, N, npad, H, ending, og_in, wo_in, bo_in, pair_out);
    }
    return (int)hipGetLastError();
#endif
}

extern "C" int prd_tri_attn_out(float* out, const float* pair, const float* og, const float* wo, const float* bo,
                                int residual, int b, int N, int P, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!out || !pair || !og || !wo || !bo || b <= 0 || N <= 0) return PRD_ERR_ARG;
    if (P != 32 && P != 64) return PRD_ERR_UNSUPPORTED;
    constexpr int NWA = 12;
    const long rows = (long)b * N * N;
    const int grid2 = grid_for((rows + 31) / 32, 4, 256);
    const bool b3 = arith == PRD_ARITH_SPLIT16;
#define PRD_TAO(PP, BB) hipLaunchKernelGGL((tri_attn_out_kernel<PP, NWA, BB>), dim3(grid2), dim3(NWA * 64), 0, stream, queue, out, pair, og, wo, bo, rows, residual)
    if (P == 64) { if (b3) PRD_TAO(64, true); else PRD_TAO(64, false); }
    else { if (b3) PRD_TAO(32, true); else PRD_TAO(32, false); }
#undef PRD_TAO
    return (int)hipGetLastError();
}

extern "C" int prd_tri_attn(float* out, const float* pair, const float* mask, const float* wq, const float* wk, const float* wv,
                            const float* wg, const float* bg, const float* wo, const float* bo, int ending, int residual,
                            int b, int N, int P, int H, int c, float* ws, size_t ws_bytes, int* queue, int arith, hipStream_t stream) {
    PRD_SPLIT_ARITH(arith);
    if (!ws) return PRD_ERR_ARG;
    if (ws_bytes < prd_workspace_bytes("tri_attn", b, N, 0, P)) return PRD_ERR_WORKSPACE;
    int e;
    if (prd_tri_attn_variant(N, P, arith_full) == 3) {
        const size_t nog = (size_t)b * N * N * 64;
        e = prd_tri_attn_core_chunked(ws, pair, mask, wq, wk, wv, wg, bg, ending, b, N, P, H, c, ws + nog, ws_bytes - nog * sizeof(float), stream);
    } else {
        e = prd_tri_attn_core(ws, pair, mask, wq, wk, wv, wg, bg, ending, b, N, P, H, c, arith_full, stream);
    }
    if (e) return e;
    return prd_tri_attn_out(out, pair, ws, wo, bo, residual, b, N, P, queue, arith_full, stream);
}
