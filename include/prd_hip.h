/* libprd_hip.so -- C ABI of the MI355X-native ProteinReDiff denoiser hot path.
 *
 * The reference (HySonLab/Protein_Redesign) is pure eager PyTorch: it has no FFI of its own.  The
 * entry points below are therefore the operator boundary a maintainer would bind from Python
 * (ctypes): one function per nn.Module.forward on the hot path,
 * plus the building blocks they are composed of.  Each comment cites the reference code the
 * function replaces (paths relative to the reference repository).
 *
 * Rules of the boundary
 *   - extern "C", plain pointers / ints / floats only.  All pointers are DEVICE pointers to
 *     contiguous fp32 (or int64 where stated) buffers owned by the caller (PyTorch's allocator).
 *   - The library never allocates or frees device memory; scratch is passed in as `ws` with its
 *     size in bytes (query with prd_workspace_bytes).  It keeps NO process-wide state and reads no
 *     environment variable: the arithmetic and every kernel-selection switch are arguments (`arith`
 *     below), outputs and workspace are caller-provided, so all entry points are re-entrant across
 *     threads and streams (tests/native/host_abi_check.c calls one entry with both arithmetics
 *     from two threads).
 *   - Every call only enqueues kernels on `stream` (the caller's current HIP stream), never
 *     synchronises, and is therefore capturable into a hipGraph.
 *   - Return value: 0 on success, a positive hipError_t from the launch, or a negative PRD_ERR_*.
 *   - Layouts: single [b,N,S], pair [b,N,N,P] channel-last, masks [b,N] fp32 0/1, weights in
 *     nn.Linear layout [out,in] row-major.  P must be 32 or 64; H*c must be 64 (4 heads x 16);
 *     S, dist_dim and S/4 must be multiples of 8.
 */
#ifndef PRD_HIP_H
#define PRD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ABI version: bumped whenever the meaning or the SIZE of a caller-owned buffer changes, so that a binding built against an older
 * header can refuse to run (compare prd_version() with the PRD_VERSION it was compiled with).
 *   100  rounds 1-4;
 *   101  round 5: prd_step_boundary's `sync` buffer grew from ONE int32 to TWO (sync[1] = sticky non-finite flag): a caller that still
 *        allocates 4 bytes would take a 4-byte out-of-bounds device write the first time a step goes non-finite. */
#define PRD_VERSION 101
#define PRD_STEP_BOUNDARY_SYNC_INTS 2   /* int32 entries of prd_step_boundary's `sync` buffer */
#define PRD_ERR_ARG (-1)        /* null pointer / non-positive dimension */
#define PRD_ERR_ALIGN (-2)      /* leading dimension not a multiple of 4 floats */
#define PRD_ERR_UNSUPPORTED (-3)/* pair_dim / head layout outside the compiled set */
#define PRD_ERR_WORKSPACE (-4)  /* workspace too small */

int prd_version(void);

/* Arithmetic of the GEMMs inside the operators.  The library keeps NO state: every entry point whose kernels depend on the
 * arithmetic takes it as its `arith` argument (the last one before `stream`; PrdGemm carries it as a field), so calls with
 * different arithmetics may run concurrently on different streams / threads.  (The Python host side keeps the process default
 * and injects it per call: protein_redesign_amd._lib: `lib().prd_set_gemm_mode`, env PRD_GEMM_MODE.)
 * The low byte of `arith` is the arithmetic; the bits above it are PRD_TUNE_* kernel-selection switches for A/B measurements
 * (0 = default dispatch; defined after PRD_ARITH_* below).
 *   PRD_ARITH_FP32 (0)     fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4): plain fp32 FMA chains;
 *   PRD_ARITH_SPLIT16 (1)  fp32 operands are split into fp16 hi + lo -- hi = RN_fp16(x), lo = RN_fp16(x - hi): 24 bits while lo is
 *      a normal fp16 number -- and multiplied on the fp16 matrix pipe with fp32 accumulation: hi*hi + hi*lo + lo*hi, 16/3 of
 *      the fp32 rate.  Used by the row GEMMs (tri_mul projection / output, attention projections and output projection, pair
 *      transition / block tail, outer-linear, pair_init, OPM), the triangle-multiplication contraction, Q*K^T and P*V of the
 *      triangle attention and the node-row linears of the single track (prd_gemm with one batch, K a multiple of 64 or 32).
 *      Weight images stay the size of the fp32 ones.  The first-generation short-row attention core (PRD_TUNE_TA_VARIANT 10) uses
 *      bf16 x 3 by truncation (24 bits, 6 products) for Q*K^T.  The single-track attention core and pair_bias run fp32 MFMA / FMA in
 *      either mode.
 *      OPERAND RANGE: fp16 holds magnitudes up to 65504 (a larger operand becomes +-inf, the result NaN -- loudly wrong, never
 *      silently saturated) and keeps a normal lo part down to |x| ~ 0.25 (2^-12 |x| >= 6.1e-5); below that the lo part is
 *      subnormal and the operand carries an ABSOLUTE error of 2^-25 ~ 3e-8 instead of a relative 2^-24.  LayerNorm-ed rows
 *      (|x| <= sqrt(C)), their gated / ReLU-ed projections, probabilities (kept x 2^4) and weights (staged x 16) are well
 *      inside; tests/test_split16_range.py drives weights x 50 / x 1e-3 (triangle multiplication), x 30 / x 1e-3 (ReLU hidden
 *      units), logits x 30 / x 400 and inputs x 1e-4 / x 1e4 through both arithmetics.  Activations or weights beyond ~1e4 / below
 *      ~1e-4 in an un-normalised position are out of range for PRD_ARITH_SPLIT16: use PRD_ARITH_FP32.  The host side does
 *      that by itself: a sampling loop / optimisation step that ends non-finite under PRD_ARITH_SPLIT16 is run again under
 *      PRD_ARITH_FP32 (or raises; ProteinReDiffModel.nonfinite_policy), see prd_step_boundary's sync[1].
 * Both arithmetics meet every parity tolerance of tests/ (the GPU suite runs its operator / step / trajectory / gradient tests in both). */
#define PRD_ARITH_FP32 0
#define PRD_ARITH_SPLIT16 1
/* Kernel-selection switches (A/B measurements; never needed for correctness -- every selectable kernel meets the same
 * tolerances): arith = PRD_ARITH_* | PRD_TUNE(switches).  Entry points without an `arith` argument that dispatch between
 * kernel generations take the switch word itself as `tune`.  The Python binding fills them from the environment variables named
 * here (protein_redesign_amd/_lib.py), the library itself never reads the environment. */
#define PRD_TUNE(switches) ((switches) << 8)
#define PRD_TUNE_TA_VARIANT_MASK 15     /* bits 0-3 (PRD_TA_VARIANT): 10 = first-generation short-row attention core, 1/2/3 = its
                                          16- / 12- / single-buffered 8-wave forms; 0 = default dispatch (second generation) */
#define PRD_TUNE_TA2_NO_V3 (1 << 4)     /* PRD_TA2_V3=0: short rows on tri_attn_core_v2 (one barrier per phase) */
#define PRD_TUNE_TA2_NO_LONG (1 << 5)   /* PRD_TA2_LONG=0: rows of 385..1024 positions stay on the first-generation long-row kernels */
#define PRD_TUNE_TA2_FLAGS_SET (1 << 6) /* PRD_TA2_FLAGS=f: bits 7-11 = f replace the per-kernel default flags of the second-generation */
#define PRD_TUNE_TA2_FLAGS(f) (PRD_TUNE_TA2_FLAGS_SET | (((f) & 31) << 7))    /* core (bit 0 key-loop priorities, bit 3 next-row prefetch) */
#define PRD_TUNE_OL_GEN2 (1 << 12)      /* PRD_OL_VARIANT=1: round-2 outer-linear kernel instead of the K-split one */
#define PRD_TUNE_TMS_NW12 (1 << 13)     /* PRD_TMS_NW=12 / 16: waves per workgroup of the split contraction (default 8) */
#define PRD_TUNE_TMS_NW16 (2 << 13)
#define PRD_TUNE_TA2_NO_XCD8 (1 << 20)  /* PRD_TA2_XCD8=0: rows in flight per head not rounded to a multiple of 8 (heads of a row spread over XCDs) */
#define PRD_TUNE_TA2_NO_GV (1 << 21)            /* PRD_TA2_GV=0: short-row core with the round-3 phase 1 (a G row GEMM + a swapped V GEMM instead of one [G|V] GEMM + transposed store) */
#define PRD_TUNE_TMS_DEPTH3 (3 << 13)           /* PRD_TMS_DEPTH=3: the split contraction (8 waves) with three chunks of operands in flight (default 2) */
#define PRD_TUNE_TMP_NW16 (1 << 22)             /* PRD_TMP_NW=16: the projection stage of the triangle multiplication on 16 waves per workgroup (default 12) */
#define PRD_TUNE_TA2_NO_TAIL_SPLIT (1 << 19)    /* PRD_TA2_TAIL=0: long rows: a last round of few rows is not split by query blocks over the idle workgroups */
#define PRD_TUNE_GEMM_XCD_COLS (1 << 18)        /* PRD_GEMM_XCDCOLS=1: 64 x 64-tile node-row GEMMs with all row tiles of a column tile on one XCD (measured: no gain) */
#define PRD_TUNE_GEMM_NO_BATCHED_RING (1 << 17) /* PRD_GEMM_BRING=0: batched / [K][N]-operand GEMMs stay on the fp32 K-split kernel */
#define PRD_TUNE_GEMM_NO_SLAB (1 << 16) /* PRD_GEMM_SLAB=0: never split K across workgroups (round-3 kernels for the transition layers) */
#define PRD_TUNE_GEMM_NO_KG (1 << 15)   /* PRD_GEMM_KG=0: node-row GEMMs with few tiles keep one wave group per workgroup (round-3 dispatch) */

/* ---- generic batched GEMM:  C[g] = epilogue(A[g] * B[g]^T)  (b_kn = 1: A[g] * B[g]) -------------
 * Replaces aten::linear / bmm / matmul on the single track (modules.py:185-225, 306-311;
 * models/AF2_modules.py:251-293, 613-628) and the triangle-multiplication einsum (modules.py:272).
 * Batch index g = g1 * G2 + g2.  Epilogue, in this order:
 *   v = acc*alpha*colscale[n] + bias[n];  v += addmat[g][m][n];  if colmask[g1][n] < 0.5: v = fill;
 *   act (0 none, 1 relu, 2 sigmoid) applied to columns n >= act_from;  v *= rowmask[g1][m] (columns n < rowmask_cols if that is set);
 *   v *= mulmat[g][m][n];  v += resid[g][m][n] (* rscale[n]);  C[g][m][n] = v  (or C2[m][n - n_split] for n >= n_split).
 * lda/ldb must be multiples of 4 floats (rows 16-byte aligned). */
typedef struct PrdGemm {
    const float* A; const float* B; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int G1, G2;
    long long sa1, sa2, sb1, sb2, sc1, sc2;
    int b_kn;
    float alpha;
    const float* colscale;          /* optional per-column factor (e.g. the query scale of packed q|k|v|g weights) */
    const float* bias;
    int act, act_from;
    const float* addmat; long long sad1, sad2; int ldadd;
    const float* colmask; long long scm1; float fill;
    const float* rowmask; long long srm1;
    const float* mulmat; long long smu1, smu2; int ldmul;
    const float* resid; long long sr1, sr2; int ldr;
    int tile_hint;                  /* 0 = automatic; 32 / 64 / 128 force the workgroup tile */
    int a_ln;                       /* 2: the A rows go through a row SOFTMAX over K on the fly (split-16 arithmetic, b_kn, K % 64 == 0,
                                       K <= 512: SPAttention's softmax inside its P V product; PRD_ERR_UNSUPPORTED elsewhere).
                                       1: A rows are LayerNorm-ed over K on the fly (no affine, eps 1e-5, biased variance) --
                                       nn.LayerNorm(K, elementwise_affine=False) fused into the linear that follows it
                                       (reference modules.py:296,306).  Needs !b_kn, K % 4 == 0 and K <= 512 (32x32 K-split tile), or -- gemm mode 1,
                                       fewer than 512 tiles of 64x64, one batch -- K % 64 == 0 and K <= 1024 (operand-ring kernel);
                                       PRD_ERR_UNSUPPORTED otherwise. */
    float* ln_out; int ldlo;        /* optional, with a_ln and G1 = G2 = 1: the LayerNorm-ed A rows are also written here (row pitch
                                       ldlo floats) by the workgroups of the first column tile -- OuterLinear needs LN(single)
                                       itself next to the W2 projection of it (modules.py:283-287) */
    int arith;                      /* PRD_ARITH_FP32 / PRD_ARITH_SPLIT16 */
    /* Several consumers of the same (LayerNorm-ed) rows in ONE launch -- the head of the trunk projects OuterProductUpdate's a | b
     * and SPAttention's q | k | v | gate from the same normalised single representation (models/AF2_modules.py:421-545, affine
     * LayerNorm parameters folded into the weights: LN_affine(x) W^T + b = LN(x) (W diag(gamma))^T + (b + W beta)): */
    float* C2; int ldc2; int n_split;   /* optional: columns n >= n_split are stored to C2[m][n - n_split] (row pitch ldc2) instead of C */
    int rowmask_cols;               /* > 0: rowmask multiplies only columns n < rowmask_cols */
    const float* rscale;            /* optional per-column factor of resid: v += resid[m][n] * rscale[n] -- an AFFINE LayerNorm-ed
                                       residual (SPAttention adds its update to LN_affine(m): models/AF2_modules.py:465-472)
                                       given as the plain normalised rows x gamma (beta goes into bias) */
    /* K split ACROSS workgroups for large weights on few rows (the single-track transition 512 -> 2048 -> 512 at M = b N = a few
     * hundred: csrc/prd_gemm.hip gemm_slab_kernel).  Taken when `ws` is given and the shape qualifies (split-16 arithmetic, one
     * batch, K % 128 == 0, N % 4 == 0, M >= 96, the epilogue uses none of addmat / colmask / mulmat / C2 / ln_out): fp32 partial
     * tiles [K slabs][M][N] go to `ws`, a second launch sums them in slab order and applies the epilogue. */
    float* ws; size_t ws_bytes;     /* optional workspace; prd_gemm_slab_workspace(M, N, K) bytes suffice */
    const float* wsum;              /* with a_ln on this path: wsum[n] = sum_k B[n][k]; the GEMM runs on the RAW rows and the
                                       LayerNorm is applied by linearity, LN(x) W^T = rstd (x W^T - mean wsum) */
    float* out_ln; int ldol;        /* optional (this path only, N <= 512): nn.LayerNorm(N, elementwise_affine=False) of the OUTPUT rows
                                       is written here as well -- the next linear of the single track starts with it */
    float a_scale;                  /* 0 or an exact power of two: the A operand is multiplied by it while it is split into fp16 hi | lo
                                       (operand-ring kernel, split-16 arithmetic) and the accumulator divided by it -- for A operands
                                       far below 1 (softmax probabilities: P V of SPAttention), whose lo part would otherwise fall
                                       into the fp16 subnormal range (see OPERAND RANGE above).  Ignored by the fp32 kernels. */
    int mul_pos;                    /* with mulmat: v = mulmat[g][m][n] > 0 ? v : 0 instead of the product (the ReLU mask of a backward pass,
                                       read from the recomputed activations) */
} PrdGemm;
size_t prd_gemm_slab_workspace(int M, int N, int K);
int prd_gemm_slab_ok(int M, int N, int K, int arith);   /* 1 when a PrdGemm of this shape with `ws` set takes the K-slab path */
int prd_gemm(const PrdGemm* args, hipStream_t stream);

/* nn.LayerNorm over the last axis, eps 1e-5; gamma/beta may be NULL (elementwise_affine=False). */
int prd_ln_rows(const float* x, float* y, const float* gamma, const float* beta,
                int rows, int C, int ldx, int ldy, hipStream_t stream);
/* In-place softmax over the first n entries of every row; entries [n, ld) are set to 0. */
int prd_softmax_rows(float* x, int rows, int n, int ld, hipStream_t stream);

/* ---- input stage (model.py:342-361; modules.py:35-97) ------------------------------------------- */
/* step-invariant part of the pair input: atom-atom bond / bond-distance embeddings and
 * residue-residue relative-position embedding (model.py:348-358).  Index tensors are int64. */
int prd_static_pair(float* out, const float* atom_mask, const float* residue_mask, const float* bond_mask,
                    const int64_t* bond_feats, const int64_t* bond_distance,
                    const int64_t* residue_index, const int64_t* chain_index,
                    const float* tab_b0, const float* tab_b1, const float* tab_b2,
                    const float* tab_bdist, const float* tab_relpos,
                    int max_bond_distance, int max_relpos, int b, int N, int P, hipStream_t stream);
/* atom part of the single input: atom_mask * sum_f E_f[atom_feats_f] / 3 (model.py:342, modules.py:47-51).
 * tables = the nine tables concatenated row-wise [sum(card), S]; offsets[9] = first row of each. */
int prd_atom_embed(float* out, const int64_t* atom_feats, const float* atom_mask, const float* tables,
                   const int* offsets, int n_feats, int b, int N, int S, hipStream_t stream);
/* per-step single input: single = static_single + residue_mask * relu(W_rt * LN(seq_t)) (model.py:343-346) */
int prd_single_init(float* single, const float* static_single, const float* seq_t, const float* residue_mask,
                    const float* w_rt, int rows, int S, int n_cls, hipStream_t stream);
/* embed_beta: ebeta[b,P] = W_beta * [sin(w t/T), cos(w t/T)] (modules.py:85-97, model.py:341,360) */
int prd_time_embed(float* ebeta, const int64_t* t, const float* freqs, const float* w_beta,
                   int num_steps, int b, int P, int time_dim, hipStream_t stream);
/* pair = static_pair + m_i m_j (W_d rbf(|z_i - z_j|) + ebeta)  (model.py:339-340, 359-361; modules.py:73-82) */
int prd_pair_init(float* pair, const float* static_pair, const float* z, const float* mask,
                  const float* centers, const float* w_dist, const float* ebeta,
                  int b, int N, int P, int dist_dim, int arith, hipStream_t stream);

/* ---- trunk operators ------------------------------------------------------------------------------
 * `queue` (where present): device pointer to 256 int32 (one counter per XCD, 128 B apart), zero before the first use, owned by the
 * caller and shared only by stream-ordered launches; the persistent waves pull 32-row tasks from it and
 * the last fetch of each resets it to zero.  NULL selects the static wave-major task order instead (same results;
 * measured faster on MI355X for every kernel, the Python host passes NULL unless PRD_TASK_QUEUE=1). */
/* pair[b,N,N,P] -> bias[b,H,N,N] = Linear(LN(pair)) permuted (modules.py:300-304 with bias, no LN affine;
 * models/AF2_modules.py:406-411,454-459 with LN affine gamma/beta and no bias). */
int prd_pair_bias(float* bias_out, const float* pair, const float* gamma, const float* beta,
                  const float* w, const float* bvec, int b, int N, int P, int H, hipStream_t stream);
/* Two bias head sets from ONE pass over the pair tensor (same rows, same LayerNorm statistics): SPAttention's pair bias
 * (AF2_modules.py:454-459) and the first folding block's attention bias (modules.py:300-304) both read the pair tensor that the
 * outer-product update leaves.  Arguments as for prd_pair_bias, per set. */
int prd_pair_bias2(float* bias_a, const float* pair, const float* gamma_a, const float* beta_a, const float* w_a,
                   const float* bvec_a, int Ha, float* bias_b, const float* gamma_b, const float* beta_b, const float* w_b,
                   const float* bvec_b, int Hb, int b, int N, int P, hipStream_t stream);
/* OuterProductUpdate tail (models/AF2_modules.py:532-545 + modules.py:395-397):
 * out[i,j,:] = (flags&1 ? pair : 0) + (flags&2 ? m_i m_j : 1) * (W_o (a_i * b_j) + b_o) / (m_i m_j + 1e-3);
 * ab = [a | b] of shape [b,N,2C].  `out` may alias `pair` (in-place residual update), here and below. */
int prd_opm_pair(float* out, const float* pair, const float* ab, const float* mask, const float* w_out,
                 const float* b_out, int flags, int b, int N, int P, int C, int arith, hipStream_t stream);
/* Head of the pair track in one row pass (split-16 arithmetic): prd_pair_init, then prd_opm_pair with flags = 1 | (apply_mask ? 2 : 0)
 * in place, then prd_pair_bias2 on the result -- pair is written once instead of written, read + written and read again
 * (model.py:339-361; models/AF2_modules.py:532-545, 454-459; modules.py:300-304, 395-397).  Arguments as in those three. */
int prd_pair_head_supported(int P, int dist_dim, int C, int arith);
int prd_pair_head(float* pair, const float* static_pair, const float* z, const float* mask, const float* centers,
                  const float* w_dist, const float* ebeta, int dist_dim, const float* ab, const float* w_out,
                  const float* b_out, int C, int apply_mask, float* bias_a, const float* gamma_a, const float* beta_a,
                  const float* w_a, const float* bvec_a, int Ha, float* bias_b, const float* gamma_b, const float* beta_b,
                  const float* w_b, const float* bvec_b, int Hb, int b, int N, int P, int arith, hipStream_t stream);
/* OuterLinear (modules.py:283-287): out[i,j,:] = (residual ? pair : 0) + W1 (x_i * x_j) + u_i - u_j + bias,
 * x = LN(single), u = x W2^T [b,N,P] with row pitch ldu floats (computed by prd_gemm; ldu > P when u is a column block of a wider
 * GEMM output: the block's single track projects u and the NEXT block's attention q|k|v|gate from the same LN(single) in one
 * launch), w = [W1 | W2] of shape [P, 2S]. */
int prd_outer_linear(float* out, const float* pair, const float* x, const float* u, int ldu, const float* w,
                     const float* bias, int residual, int b, int N, int P, int S, int* queue, int arith, hipStream_t stream);
/* TriangleMultiplication (modules.py:262-274): out = (residual ? pair : 0) + update(pair).
 * ws: 3 * b * P * N * round_up(N,32) floats (query prd_workspace_bytes). */
int prd_tri_mul(float* out, const float* pair, const float* mask, const float* w_proj, const float* b_proj,
                const float* w_gate, const float* b_gate, const float* w_out, const float* b_out,
                const float* w_ogate, const float* b_ogate, int incoming, int residual,
                int b, int N, int P, float* ws, size_t ws_bytes, int* queue, int arith, hipStream_t stream);
/* ---- backward of TriangleMultiplication (autograd of modules.py:262-274; used by training.py) -------------------------------
 * The contraction of prd_tri_mul alone: O[b][d][i][j] = sum_k A[b][d][i][k] B[b][d][j][k], operands channel-major
 * AB[b][2P][N][ldn] (A = channels 0..P-1, B = channels P..2P-1, ldn = round_up(N,32), zero padded), O[b][P][N][ldn].
 * The backward calls it on transposed operands for dA and dB (P = 128 is accepted for its two contractions stacked). */
int prd_tri_mul_contract(float* O, const float* AB, int b, int N, int P, int arith, hipStream_t stream);
/* The operands of both gradient contractions stacked for ONE prd_tri_mul_contract call with 2P channel pairs:
 * ops[b][4P][N][ldn] = dO | dO^T | B^T | A^T by channel block (dO already in block 0: prd_tri_mul_out_bwd with
 * dO_batch_channels = 4P); A, B = the forward operands AB[b][2P][N][ldn].  prd_tri_mul_contract(dAB, ops, b, N, 2P) then yields
 * dAB[b][2P][N][ldn] = dA | dB.  Padding columns N..ldn of all four blocks are zeroed here. */
int prd_tri_mul_bwd_operands(float* ops, const float* AB, int b, int N, int P, hipStream_t stream);
/* TriangleMultiplication "outgoing" followed by "incoming", in place on `pair` (both residual updates of modules.py:336-337),
 * PRD_ARITH_SPLIT16 only (prd_tri_mul_chain_supported): five launches instead of six -- the output stage of the first module and the
 * projection stage of the second run as one row pass down the columns (the outgoing contraction stores its result transposed
 * for it).  w_outgoing / w_incoming: eight device pointers each, in prd_tri_mul's order (w_proj, b_proj, w_gate, b_gate, w_out,
 * b_out, w_ogate, b_ogate).  Workspace as for prd_tri_mul.  Results equal two prd_tri_mul calls up to fp32 rounding. */
int prd_tri_mul_chain_supported(int N, int P, int arith);
int prd_tri_mul_chain(float* pair, const float* mask, const float* const* w_outgoing, const float* const* w_incoming,
                      int b, int N, int P, float* ws, size_t ws_bytes, int arith, hipStream_t stream);
/* Output stage backward.  dy = gradient of the update [b,N,N,P]; O = contraction output (channel-major, as left in prd_tri_mul's
 * workspace); w_*_t = the transposed weights [in][out].  Writes dz = dy * gate and dgp = d(pre-activation of the output gate)
 * (row layout [b,N,N,P]; dW_out = dz^T LN(O), dW_ogate = dgp^T LN(pair) are left to the caller's BLAS), dO (channel-major) and
 * dx1 = W_ogate^T dgp (row layout), the output-gate path of the gradient of LN(pair).
 * w_out_t / w_ogate_t may be NULL: the kernel then stages the transposed images from w_out / w_ogate read column-wise (no
 * transposed copy needed); likewise w_proj_t / w_gate_t of prd_tri_mul_proj_bwd.
 * x_out, lo_out (either may be null): LN(pair) and LN(O) in row layout [b,N,N,P], the inputs of those weight gradients (the kernel
 * has both in registers).  dO_batch_channels: channel planes between the batches of dO (0 = P, i.e. dO[b][P][N][ldn]; 4P when dO
 * is block 0 of prd_tri_mul_bwd_operands' buffer). */
int prd_tri_mul_out_bwd(float* dz, float* dgp, float* dO, float* dx1, const float* dy, const float* pair, const float* O,
                        const float* w_out, const float* b_out, const float* w_ogate, const float* b_ogate,
                        const float* w_out_t, const float* w_ogate_t, float* x_out, float* lo_out, int dO_batch_channels,
                        int b, int N, int P, hipStream_t stream);
/* Projection stage backward.  dAB = gradient of the operands (channel-major [b][2P][N][ldn]); writes dpair [b,N,N,P] (gradient of
 * the update with respect to its input pair tensor), and dpp / dpg = d(pre-activations of ab_proj / ab_gate) in row layout
 * [b,N,N,2P] by pair position (dW_proj = dpp^T LN(pair), dW_gate = dpg^T LN(pair) are left to the caller's BLAS). */
int prd_tri_mul_proj_bwd(float* dpair, float* dpp, float* dpg, const float* dAB, const float* dx1, const float* pair,
                         const float* mask, const float* w_proj, const float* b_proj, const float* w_gate, const float* b_gate,
                         const float* w_proj_t, const float* w_gate_t, int incoming, int b, int N, int P, int arith, hipStream_t stream);

/* SURVEY 8(f)#4 "persistent per-block kernels", built for one seam of the folding block: the STARTING triangle attention (core +
 * output projection + residual, modules.py:338 -> 236-243 -> 185-225) and the core of the ENDING one (modules.py:339) as ONE persistent
 * launch with two in-kernel grid barriers, instead of prd_tri_attn_core_v2 + prd_tri_attn_out + prd_tri_attn_core_v2.  In: pair
 * [b,N,N,P] (updated IN PLACE by the starting attention), mask [b,N]; w_start = {q.w, k.w, v.w, gate.w, gate.b, out.w, out.b} of
 * pair_attn_starting, w_end = {q.w, k.w, v.w, gate.w, gate.b} of pair_attn_ending.  Out: og [b,N,N,64] = the gated head outputs of the
 * ENDING attention (prd_block_tail applies its output projection).  Bit-identical to the three launches.  bar: 32 uint32 of device
 * memory owned by the caller (counters, memberships, bar[1] = timeout flag), zeroed by the call itself; after the stream has run,
 * bar[1] != 0 means a barrier gave up waiting (the grid was not fully resident: the results are then NOT valid).  Refused (PRD_ERR_UNSUPPORTED) unless
 * prd_tri_attn_pair_supported: split-16 arithmetic, rows on the overlapped-phase short-row core (N <= 320-odd), default kernel
 * switches, at most one workgroup per CU.  Opt-in on the Python side (PRD_PERSISTENT_ATTN=1): measured against the three launches in
 * DESIGN.md 4.3. */
int prd_tri_attn_pair_supported(int N, int P, int arith);
int prd_tri_attn_pair(float* og, float* pair, const float* mask, const float* const* w_start, const float* const* w_end,
                      int b, int N, int P, int H, int c, unsigned* bar, int arith, hipStream_t stream);

/* ---- backward of TriangleAttention (autograd of modules.py:236-243 -> 185-225; used by training.py) -------------------------
 * Core: dog = W_out^T d(update) [b,N,N,64] (a row GEMM by the caller) -> dqkvg[b,N,N,4,64] by pair position =
 * d(W_q x) | d(W_k x) | d(W_v x) | d(gate pre-activation), channels head-major (x = LN(pair row)).  Rows up to N ~ 400. */
int prd_tri_attn_bwd_core(float* dqkvg, const float* dog, const float* pair, const float* mask, const float* wq,
                          const float* wk, const float* wv, const float* wg, const float* bg, int ending,
                          int b, int N, int P, int H, int c, hipStream_t stream);
/* The same gradients in split-16 arithmetic on the 16-bit matrix pipe (tri_attn_bwd_core_v2_kernel, csrc/prd_tri2.hip), for rows of
 * up to 384 positions.  Takes, besides dog, the gated head outputs og [b,N,N,64] of the forward (do . o = dog . og is not
 * recomputed) and, optionally, the softmax statistics lse [b*N,H,N,2] that prd_tri_attn_core_v2_lse wrote in the forward
 * (null: recomputed by one more sweep over the logits).  x_out (may be null): receives LN(pair) [b,N,N,P], the input of the
 * projections' weight gradients.  prd_tri_attn_bwd_core_v2_supported: 1 when (N, P) is served. */
int prd_tri_attn_bwd_core_v2_supported(int N, int P);
int prd_tri_attn_bwd_core_v2(float* dqkvg, const float* dog, const float* og, const float* pair, const float* mask,
                             const float* wq, const float* wk, const float* wv, const float* wg, const float* bg, const float* lse,
                             float* x_out, int ending, int b, int N, int P, int H, int c, hipStream_t stream);
/* The two reductions of the outer-linear backward over T [R][P][S] (R = b N node rows; T = (dy + dy^T) LN(single), the output of the
 * backward's GEMM): dx[r][s] = sum_p T[r][p][s] w1[p][s], and dw1_part[c][p][s] = sum over the rows of chunk c (of `chunks` equal
 * chunks of R) of T[r][p][s] x[r][s] -- the caller adds the chunks (fixed order: no atomics). */
int prd_outer_linear_bwd_reduce(float* dx, float* dw1_part, int chunks, const float* T, const float* w1, const float* x,
                                long long R, int P, int S, hipStream_t stream);
/* out[b][i][j][:] = scale (x[b][i][j][:] + x[b][j][i][:]): the pair symmetrisation in front of the heads (modules.py:403) and its
 * backward.  Not in place; P a multiple of 4. */
int prd_sym_rows(float* out, const float* x, float scale, int b, int N, int P, hipStream_t stream);
/* out[b][i][p][j] = dy[b][i][j][p] + dy[b][j][i][p] (dy [b,N,N,P] -> out [b,N,P,N]): the symmetrised, transposed gradient the
 * backward of the outer-linear update (modules.py:283-287) contracts with LN(single) over j.  P in {32, 64}. */
int prd_sym_transpose(float* out, const float* dy, int b, int N, int P, hipStream_t stream);
/* A linear at every pair position as a row kernel (the activation-gradient GEMMs of the training backward: 2e5 rows, K and OUT
 * at most 256; autograd of nn.Linear over [b,N,N,*], modules.py:236-243, 321-326): out[row][0..OUT) = act(LN?(x[row]) W^T + bias),
 * W [OUT][K] as in nn.Linear.  ln_in: LayerNorm (no affine) of the x rows first (K = 64), xn_out (optional) receives them.
 * act: 0 none, 1 ReLU.  mask_pos (optional) [rows][OUT]: the result is zeroed where mask_pos <= 0 (ReLU backward from recomputed
 * activations).  w_kn = 1: W is given TRANSPOSED in memory, [K][OUT] (the backward of a linear multiplies by the forward's weight
 * as it lies: no transposed copy).  (K, OUT) in {(64,64), (64,256), (256,64)}, split-16 arithmetic (prd_pair_linear_supported); 16-byte aligned row tensors. */
int prd_pair_linear_supported(int K, int OUT, int arith);
int prd_pair_linear(float* out, const float* x, const float* w, const float* bias, long long rows, int K, int OUT,
                    int ln_in, float* xn_out, int act, const float* mask_pos, int w_kn, int arith, hipStream_t stream);
/* d/dx of nn.LayerNorm(C, elementwise_affine=False) applied to the rows of x: dx = LN'(dy; x) (+ res[row][c] when res is given:
 * the gradient that bypasses the update through its residual connection). */
int prd_ln_rows_bwd(float* dx, const float* dy, const float* x, const float* res, long long rows, int C, hipStream_t stream);

/* Backward of an attention-bias head over the pair rows (autograd of ProteinReDiff/modules.py:300-304 and of SPAttention's linear_z,
 * models/AF2_modules.py:454-459: bias[b,h,i,j] = (W' LN(pair[b,i,j]))[h] with W' = W diag(gamma) for the affine form), one pass:
 * dx [b nn, P] = LN'(dbias . W'; x);  xn (may be NULL) = LN(x) and d2 (may be NULL) [b nn, H] = dbias by position, the operands of
 * the weight-gradient reduction (prd_linear_wgrad).  dbias [b, H, nn] as the forward wrote the bias; wf [H, P]; P = 64, H = 4 or 8
 * (PRD_ERR_UNSUPPORTED otherwise); dx, xn, x, wf 16-byte aligned. */
int prd_pair_bias_bwd(float* dx, float* xn, float* d2, const float* dbias, const float* wf, const float* x, int b, long long nn,
                      int H, int P, hipStream_t stream);
/* Weight gradient of a linear applied at every pair position (autograd of nn.Linear over [b,N,N,*] activations, e.g.
 * modules.py:262-274, 321-326): dw[O][I] = sum over rows of dy[row][0..O) (x) x[row][0..I); row pitches lddy / ldx floats (for O > 16: even, and dy / x
 * 8-byte aligned -- PRD_ERR_ALIGN otherwise).
 * I a multiple of 64, O a multiple of 64 or at most 16 (the attention-bias / coordinate-head linears), both at most 256.
 * db (optional, NULL = skip): the bias gradient db[O] = sum over rows of dy, from the same pass over dy.
 * ws: prd_linear_wgrad_workspace(rows, O, I) bytes of slab partials.
 * arith: PRD_ARITH_FP32 = fp32 MFMA; PRD_ARITH_SPLIT16 (wide form) = fp16 hi | lo operands on the 16-bit matrix pipe, dy scaled by a
 * running power of two per wave (gradients have no fixed magnitude; csrc/prd_bwd.hip linear_wgrad_h2_kernel).  db is summed in
 * fp32 either way. */
size_t prd_linear_wgrad_workspace(long long rows, int O, int I);
int prd_linear_wgrad(float* dw, float* db, const float* dy, const float* x, long long rows, int O, int I, int lddy, int ldx,
                     float* ws, size_t ws_bytes, int arith, hipStream_t stream);
/* Gradient of a small embedding table applied at every pair position (modules.py:35-71: bond-type, bond-distance and
 * relative-position tables): dtable[card][C] = sum over rows of dy[row][0..C) into row idx[row] (int64; indices outside
 * [0, card) are ignored).  card <= 128, C <= 64.  row_scale (optional) [rows]: dy[row] is multiplied by it (the mask factors the
 * lookup is multiplied by in the forward).  ws: prd_embed_wgrad_workspace(rows, card, C) bytes. */
size_t prd_embed_wgrad_workspace(long long rows, int card, int C);
int prd_embed_wgrad(float* dtable, const long long* idx, const float* dy, const float* row_scale, long long rows, int card, int C,
                    int lddy, float* ws, size_t ws_bytes, hipStream_t stream);
/* K <= 8 small tables looked up at the same rows (the input stage's bond-feature, bond-distance and relative-position tables): all
 * their gradients in one pass over dy.  idx / row_scale / card: HOST arrays of K device pointers / K device pointers (entries or the
 * whole array may be NULL = no scale) / K cardinalities; dtables [sum card][C] = the K gradients stacked in order.  sum card <= 128,
 * C <= 64; ws: prd_embed_wgrad_workspace(rows, sum card, C) bytes. */
int prd_embed_wgrad_multi(float* dtables, const long long* const* idx, const float* const* row_scale, const int* card, int K,
                          const float* dy, long long rows, int C, int lddy, float* ws, size_t ws_bytes, hipStream_t stream);
/* rbf[b][i][j][0..R) = mask[b][i] mask[b][j] exp(-(R-1)/2 (|z_i - z_j| - centers[r])^2): the radial-basis rows of the pair distances
 * (modules.py:73-82, model.py:352-356) materialised for the weight gradient of the distance embedding (dW = dy^T rbf through
 * prd_linear_wgrad); R a multiple of 4. */
int prd_rbf_rows(float* out, const float* z, const float* centers, const float* mask, int b, int N, int R, hipStream_t stream);

/* TriangleAttention (modules.py:236-243 -> 185-225): out = (residual ? pair : 0) + update(pair).
 * ws: b * N * N * 64 floats. */
int prd_tri_attn(float* out, const float* pair, const float* mask, const float* wq, const float* wk, const float* wv,
                 const float* wg, const float* bg, const float* wo, const float* bo, int ending, int residual,
                 int b, int N, int P, int H, int c, float* ws, size_t ws_bytes, int* queue, int arith, hipStream_t stream);
/* which core kernel prd_tri_attn uses for rows of N positions under arithmetic `arith`: 0 = short rows (K, V, Q and gate
 * of a row resident in LDS: N <= 448 in fp32 mode, N <= 384 with split operands), 1 = long rows on the fp32 kernel (Q / gate
 * re-projected per query block), 2 = long rows on a split-operand kernel (PRD_ARITH_SPLIT16, N <= 1024),
 * 3 = rows whose K / V no longer fit the LDS (N > 960 in fp32 arithmetic, N > 1024 with split operands): key-chunked,
 * prd_tri_attn_core_chunked. */
int prd_tri_attn_variant(int N, int P, int arith);
/* Rows of any length (variant 3): the keys of a row are processed in chunks of <= 960 by consecutive launches of the fp32
 * long-row kernel; every launch attends all queries of the row to its chunk and merges its (gated, normalised) output into og
 * by the softmax statistics (reference m, sum l) kept per (position, head) in `stats` -- the softmax over the union of the
 * chunks (modules.py:216-223), exact up to fp32 rounding.  stats: prd_tri_attn_stats_bytes() bytes = b N N H 2 floats (0 for
 * variants 0-2).  prd_tri_attn takes the statistics from its ws (prd_workspace_bytes("tri_attn") includes them);
 * prd_tri_attn_core returns PRD_ERR_UNSUPPORTED for such rows. */
size_t prd_tri_attn_stats_bytes(int b, int N, int P, int H, int arith);
int prd_tri_attn_core_chunked(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                              const float* wv, const float* wg, const float* bg, int ending,
                              int b, int N, int P, int H, int c, float* stats, size_t stats_bytes, hipStream_t stream);
/* the two launches of prd_tri_attn, exposed separately (og = gated per-head output [b,N,N,64]) */
int prd_tri_attn_core(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                      const float* wv, const float* wg, const float* bg, int ending,
                      int b, int N, int P, int H, int c, int arith, hipStream_t stream);
int prd_tri_attn_out(float* out, const float* pair, const float* og, const float* wo, const float* bo,
                     int residual, int b, int N, int P, int* queue, int arith, hipStream_t stream);
/* Second-generation core (split-16 arithmetic; csrc/prd_tri2.hip): same contract as prd_tri_attn_core, everything on the
 * 32x32x16 fp16 MFMA (Q K^T with fp16 hi+lo operands rounded to nearest: 24 bits).  Rows of up to 384 positions keep K, Q, V
 * and the gate of a row in LDS (one query block per wave, shared blocks merged from partials); longer rows -- as far as K and
 * V fit as fp16 planes, N <= 1024 -- re-project Q and the gate per query block in the wave that sweeps it.  prd_tri_attn_core
 * dispatches to it when prd_tri_attn_v2_supported(N, P, tune) and the arithmetic is split-16 (tune: the PRD_TUNE_* switch word;
 * PRD_TUNE_TA_VARIANT 10 keeps the first generation, PRD_TUNE_TA2_NO_LONG keeps it for the long rows only). */
int prd_tri_attn_v2_supported(int N, int P, int tune);
/* which of its kernels serves rows of N positions: 0 none, 1 short rows with one barrier per phase (tri_attn_core_v2_kernel),
 * 2 short rows with the phases of consecutive rows overlapped (tri_attn_core_v3_kernel; the default where its two K / V buffers
 * fit the LDS), 3 long rows (tri_attn_core_v2l_kernel) */
int prd_tri_attn_v2_form(int N, int P, int tune);
int prd_tri_attn_core_v2(float* og, const float* pair, const float* mask, const float* wq, const float* wk,
                         const float* wv, const float* wg, const float* bg, int ending,
                         int b, int N, int P, int H, int c, int tune, hipStream_t stream);
/* prd_tri_attn_core_v2 that also writes, for rows of up to 384 positions, the softmax statistics of every query:
 * lse [b*N rows][H][N][2] = (m, log2 l) with l = sum over the keys of 2^(logit log2(e) - m); lse may be null. */
int prd_tri_attn_core_v2_lse(float* og, float* lse, const float* pair, const float* mask, const float* wq, const float* wk,
                             const float* wv, const float* wg, const float* bg, int ending,
                             int b, int N, int P, int H, int c, int tune, hipStream_t stream);
/* Triangle attention core whose input row is `pair + og_in W_o^T + b_o`: the residual update of the PREVIOUS triangle attention
 * (its output projection, modules.py:339-340) is applied while the row is loaded, and written to `pair_out` -- which must not
 * alias `pair` -- by the workgroups of head 0, instead of a separate prd_tri_attn_out launch.  gemm mode 1, short rows only
 * (prd_tri_attn_core_fused_supported); og as prd_tri_attn_core.  Equal to prd_tri_attn_out(pair_out, pair, og_in, ...,
 * residual = 1) followed by prd_tri_attn_core(og, pair_out, ...) up to fp32 rounding. */
int prd_tri_attn_core_fused_supported(int N, int P, int arith);
int prd_tri_attn_core_fused(float* og, float* pair_out, const float* pair, const float* og_in, const float* wo_in,
                            const float* bo_in, const float* mask, const float* wq, const float* wk, const float* wv,
                            const float* wg, const float* bg, int ending, int b, int N, int P, int H, int c,
                            hipStream_t stream);
/* single-track gated attention core for heads of width 16 (modules.py:216-223): qkvg = [q/sqrt(c) | k | v | sigmoid(gate)]
 * of shape [b,N,4*H*c] with row pitch ldq floats (one packed prd_gemm; ldq > 4 H c when it is a column block of a wider GEMM
 * output), bias [b,H,N,N], mask [b,N] or NULL -> o[b,N,H*c] = gate * softmax(qk + bias) v */
int prd_single_attn_core(float* o, const float* qkvg, int ldq, const float* bias, const float* mask,
                         int b, int N, int H, int c, hipStream_t stream);
/* pair transition (modules.py:321-326): out = (residual ? pair : 0) + W2 relu(W1 LN(pair) + b1) + b2, hidden = 4P */
int prd_pair_transition(float* out, const float* pair, const float* w1, const float* b1, const float* w2,
                        const float* b2, int residual, int b, int N, int P, int* queue, int arith, hipStream_t stream);
/* fused tail of a folding block, in place on `pair`: output projection of the ENDING triangle attention
 * (og from prd_tri_attn_core; modules.py:341), pair transition (modules.py:342), and optionally the next block's
 * attention bias Linear(LN(pair)) -> bias_out[b,H,N,N] (modules.py:300-304; bias_out NULL = skip). */
int prd_block_tail(float* pair, const float* og, const float* wo, const float* bo, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* bias_w, const float* bias_b, float* bias_out,
                   int b, int N, int P, int H, int* queue, int arith, hipStream_t stream);
/* coordinate head (modules.py:403 + model.py:364-372): symmetrise, LN -> Linear -> ReLU -> Linear(1),
 * eps_raw[b,N,3] = sum_j m_i m_j w_ij (z_i - z_j) rsqrt(|z_i - z_j|^2 + 1e-4)  (mean not yet removed) */
int prd_coord_head(float* eps_raw, const float* pair, const float* z, const float* mask,
                   const float* w1, const float* b1, const float* w2, int b, int N, int P, int arith, hipStream_t stream);
/* remove_mean (utils.py:32-36) applied to eps_raw -> noise_pred */
int prd_remove_mean(float* out, const float* x, const float* mask, int b, int N, int D, hipStream_t stream);
/* reverse-diffusion update (model.py:405-420): z <- (z - w_t eps)/sqrt(alpha_t) [+ sqrt(beta_t) remove_mean(noise)],
 * seq_t <- 2 softmax(seq_pred) - 1, t <- t - 1.  coef = [T][4] = {w_noise, 1/sqrt_alpha, sqrt_beta, 0};
 * noise = [T-1][b][N][3], row T-1-t is consumed at step t > 0 (so one captured hipGraph serves every step). */
int prd_reverse_update(float* z, float* seq_t, int64_t* t, const float* noise_pred, const float* seq_pred,
                       const float* noise, const float* mask, const float* coef,
                       int b, int N, int n_cls, int num_steps, hipStream_t stream);

/* The whole step boundary of the sampling loop in one launch (model.py:373, 405-420 and the next step's model.py:341-346):
 * noise_pred = remove_mean(eps_raw); z, seq_t advanced as in prd_reverse_update; t <- t - 1; and the NEXT step's inputs
 * ebeta_next[b,P] = time embedding of t - 1 (prd_time_embed) and single_next[b,N,S] = prd_single_init of the new seq_t.
 * sync: TWO int32 (PRD_STEP_BOUNDARY_SYNC_INTS; ONE before PRD_VERSION 101), zero before the first launch, owned by the caller and shared only by stream-ordered launches: sync[0] is the
 * arrival counter (the last workgroup to arrive advances t and resets it); sync[1] is a STICKY non-finite flag -- set to 1, never
 * cleared by the library, as soon as a new coordinate or a new sequence entry is inf / NaN.  The reference is fp32 end to end
 * (model.py:377-422) and cannot overflow at 65504; under PRD_ARITH_SPLIT16 an out-of-range operand can (OPERAND RANGE above) and
 * shows up here: the host reads the flag once per sampling loop and re-runs the call under PRD_ARITH_FP32 or fails
 * (protein_redesign_amd.diffusion_model: ProteinReDiffModel.nonfinite_policy).  n_cls must be 21 (20 residue types + 'X'),
 * time_dim <= 512 and even.
 * seq_h != NULL: seq_pred is an OUTPUT -- the sequence head's last layer (model.py:117-122: Linear(S_h, 21, bias = False) with
 * weight w_seq [21, S_h]) is applied here to its ReLU hidden units seq_h [b,N,S_h] (row pitch ldh), instead of a GEMM launch of
 * its own; seq_h == NULL: seq_pred is read. */
int prd_step_boundary(float* z, float* seq_t, int64_t* t, const float* eps_raw, float* seq_pred,
                      const float* noise, const float* mask, const float* coef,
                      float* single_next, const float* static_single, const float* residue_mask, const float* w_rt,
                      float* ebeta_next, const float* freqs, const float* w_beta, int* sync,
                      int b, int N, int n_cls, int num_steps, int S, int P, int time_dim,
                      const float* seq_h, int ldh, const float* w_seq, int S_h, hipStream_t stream);

/* SPAttention's core with wide heads in one launch (models/AF2_modules.py:421-473 -> 251-293, 613-628: the gated attention over the
 * node axis with head width c = single_dim and an additive pair bias; the reference builds a mask bias and drops it, :447 vs 461-463):
 *   o[b,q,h,:] = gate[b,q,h,:] * sum_k softmax_k(q[b,q,h,:] . k[b,k,h,:] + bias[b,h,q,k]) v[b,k,h,:]
 * qkvg = [b,N,4 H c] rows of pitch ldq: [q * 1/sqrt(c) | k | v | sigmoid(gate)] as the packed projection writes them; bias [b,H,N,N]
 * (may be NULL); mask [b,N] (may be NULL: keys with mask < 0.5 get the reference's fill value -2^15); o [b,N,H c].  Logits, softmax
 * never reach memory.  PRD_ARITH_SPLIT16 only, 64 <= c <= 512 a multiple of 64 (prd_spa_attn_core_supported);
 * otherwise PRD_ERR_UNSUPPORTED and the caller runs logits GEMM + softmax + P V GEMM (prd_gemm). */
int prd_spa_attn_core_supported(int N, int c, int arith);
/* bytes of `ws` for these sizes (0: the keys are not split and no workspace is needed): with few (complex, head, query block)
 * triples the KEYS are split over workgroups -- what bounds the operator is the operand stream per CU -- and a second, small launch
 * merges the parts by their softmax statistics */
size_t prd_spa_attn_core_workspace(int b, int N, int H, int c);
int prd_spa_attn_core(float* o, const float* qkvg, int ldq, const float* bias, const float* mask,
                      int b, int N, int H, int c, float* ws, size_t ws_bytes, int arith, hipStream_t stream);

/* bytes of scratch an operator needs: op = "tri_mul" | "tri_attn" */
size_t prd_workspace_bytes(const char* op, int b, int N, int S, int P);

#ifdef __cplusplus
}
#endif
#endif /* PRD_HIP_H */
