/* Host-side check of the C ABI's argument validation (include/prd_hip.h), meant to run under AddressSanitizer on a machine
 * WITHOUT a GPU: every call below must be rejected by the argument checks (negative PRD_ERR_* code) before any HIP API is
 * touched, with no out-of-bounds access, leak or use of uninitialised memory on the host side of libprd_hip.
 * Build + run: python -m protein_redesign_amd.build --asan   (tests/test_host_cpu.py::test_c_abi_argument_checks_under_asan) */
#include <pthread.h>
#include <stdio.h>
#include <string.h>

#include "../../include/prd_hip.h"

static int failures = 0;
#define EXPECT(call, code)                                                                 \
    do {                                                                                   \
        const int got_ = (call);                                                           \
        if (got_ != (code)) { printf("FAIL %s -> %d (want %d)\n", #call, got_, (code)); ++failures; } \
    } while (0)

/* Two threads call the same entry points with DIFFERENT arithmetics and switches at the same time; each must get the answer
 * that belongs to ITS arguments every time (the library has no mode to race on). */
struct thread_arg { int arith; int want_variant; int want_chain; int want_stats; int bad; };
static void* hammer(void* v) {
    struct thread_arg* a = (struct thread_arg*)v;
    for (int it = 0; it < 20000; ++it) {
        if (prd_tri_attn_variant(1000, 64, a->arith) != a->want_variant) ++a->bad;
        if (prd_tri_mul_chain_supported(320, 64, a->arith) != a->want_chain) ++a->bad;
        if ((prd_tri_attn_stats_bytes(1, 1000, 64, 4, a->arith) != 0) != a->want_stats) ++a->bad;
    }
    return 0;
}

int main(void) {
    _Alignas(16) float buf[16];                      /* host memory standing in for device pointers: never dereferenced by the checks */
    int64_t ibuf[4];
    float* p = buf;
    hipStream_t s = 0;
    EXPECT(prd_version(), PRD_VERSION);
    EXPECT(PRD_VERSION >= 101 && PRD_STEP_BOUNDARY_SYNC_INTS == 2, 1);      /* the sync buffer of prd_step_boundary: two int32 since 101 */
    const int A1 = PRD_ARITH_SPLIT16, A0 = PRD_ARITH_FP32;     /* the arithmetic is an argument of every call: no library state */
    PrdGemm g;
    memset(&g, 0, sizeof g);
    EXPECT(prd_gemm(&g, s), PRD_ERR_ARG);                              /* null operands */
    g.A = g.B = p; g.C = p; g.M = g.N = g.K = 8; g.G1 = g.G2 = 1; g.lda = 6; g.ldb = 8; g.ldc = 8; g.arith = 7;
    EXPECT(prd_gemm(&g, s), PRD_ERR_ARG);                              /* unknown arithmetic */
    g.arith = A1;
    EXPECT(prd_gemm(&g, s), PRD_ERR_ALIGN);                            /* lda not a multiple of 4 */
    g.lda = 8; g.a_ln = 1; g.K = 4096; g.lda = g.ldb = 4096;
    EXPECT(prd_gemm(&g, s), PRD_ERR_UNSUPPORTED);                      /* fused LayerNorm beyond the row the kernels keep in registers */
    g.K = 8; g.lda = g.ldb = 8; g.a_ln = 0; g.ln_out = p; g.ldlo = 8;
    EXPECT(prd_gemm(&g, s), PRD_ERR_ARG);                              /* ln_out without a_ln */
    g.ln_out = 0;
    EXPECT(prd_ln_rows(0, p, 0, 0, 4, 4, 4, 4, s), PRD_ERR_ARG);
    EXPECT(prd_ln_rows(p, p, p, 0, 4, 4, 4, 4, s), PRD_ERR_ARG);       /* gamma without beta */
    EXPECT(prd_softmax_rows(p, 4, 8, 4, s), PRD_ERR_ARG);             /* ld < n */
    EXPECT(prd_pair_init(p, p, p, p, p, p, p, 1, 8, 48, 256, A1, s), PRD_ERR_UNSUPPORTED);        /* pair_dim 48 */
    EXPECT(prd_pair_init(p, p, p, p, p, p, p, 1, 8, 64, 100, A1, s), PRD_ERR_UNSUPPORTED);       /* dist_dim % 8 */
    EXPECT(prd_pair_init(p, p, p, p, p, p, p, 1, 8, 64, 256, 5, s), PRD_ERR_ARG);               /* unknown arithmetic */
    EXPECT(prd_pair_bias(p, p, p, 0, p, p, 1, 8, 64, 4, s), PRD_ERR_ARG);                    /* gamma without beta */
    EXPECT(prd_pair_bias(p, p, 0, 0, p, p, 1, 8, 64, 9, s), PRD_ERR_ARG);                    /* more than 8 heads */
    EXPECT(prd_pair_bias2(p, p, 0, 0, p, p, 4, 0, 0, 0, p, p, 4, 1, 8, 64, s), PRD_ERR_ARG);     /* second output missing */
    EXPECT(prd_pair_bias2(p, p, p, 0, p, p, 4, p, 0, 0, p, p, 4, 1, 8, 64, s), PRD_ERR_ARG);     /* gamma without beta */
    EXPECT(prd_opm_pair(p, p, p, p, p, p, 3, 1, 8, 64, 12, A1, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_outer_linear(p, p, p, p, 64, p, p, 1, 1, 8, 64, 36, 0, A1, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_outer_linear(p, p, p, p, 48, p, p, 1, 1, 8, 64, 512, 0, A1, s), PRD_ERR_ALIGN);            /* u row pitch below P */
    EXPECT(prd_tri_mul(p, p, p, p, p, p, p, p, p, p, p, 0, 1, 1, 8, 64, p, 16, 0, A1, s), PRD_ERR_WORKSPACE);
    EXPECT(prd_tri_mul(p, p, p, p, p, p, p, p, p, p, p, 0, 1, 1, 8, 40, p, 1 << 20, 0, A0, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn(p, p, p, p, p, p, p, p, p, p, 0, 1, 1, 8, 64, 4, 16, 0, 0, 0, A1, s), PRD_ERR_ARG);   /* no workspace */
    EXPECT(prd_tri_attn(p, p, p, p, p, p, p, p, p, p, 0, 1, 1, 8, 64, 4, 16, p, 16, 0, A1, s), PRD_ERR_WORKSPACE);
    EXPECT(prd_tri_attn_core(p, p, p, p, p, p, p, p, 0, 1, 8, 64, 2, 32, A1, s), PRD_ERR_UNSUPPORTED);        /* head_dim 32 */
    EXPECT(prd_tri_attn_core(p, p, p, p, p, p, p, p, 0, 1, 100000, 64, 4, 16, A1, s), PRD_ERR_UNSUPPORTED);   /* row too long for LDS */
    EXPECT(prd_tri_attn_variant(320, 64, A1), 0);
    EXPECT(prd_tri_attn_variant(769, 64, A1), 2);       /* split-16: the split-operand long-row kernel */
    EXPECT(prd_tri_attn_variant(769, 64, A0), 1);
    EXPECT(prd_tri_attn_variant(900, 64, A1), 2);                   /* split-16: the round-3 long-row core (K / V as fp16 planes) */
    EXPECT(prd_tri_attn_variant(100000, 64, A1), 3);                 /* any row length: key-chunked */
    EXPECT(prd_tri_attn_variant(320, 48, A1), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn_variant(320, 64, 2), PRD_ERR_ARG);
    EXPECT(prd_single_attn_core(p, p, 256, p, p, 1, 8, 2, 32, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_single_attn_core(p, p, 200, p, p, 1, 8, 4, 16, s), PRD_ERR_ALIGN);                    /* qkvg row pitch below 4 H c */
    EXPECT(prd_block_tail(p, p, p, p, p, p, p, p, 0, 0, p, 1, 8, 64, 4, 0, A1, s), PRD_ERR_ARG);   /* bias_out without bias weights */
    EXPECT(prd_coord_head(p, p, p, p, p, p, 0, 1, 8, 64, A1, s), PRD_ERR_ARG);
    EXPECT(prd_remove_mean(p, p, p, 1, 8, 65, s), PRD_ERR_ARG);
    EXPECT(prd_reverse_update(p, p, ibuf, p, p, p, p, 0, 1, 8, 21, 10, s), PRD_ERR_ARG);
    EXPECT(prd_static_pair(p, p, p, p, ibuf, ibuf, ibuf, ibuf, p, p, p, p, p, 7, 32, 1, 8, 62, s), PRD_ERR_ALIGN);
    EXPECT(prd_time_embed(p, ibuf, p, p, 10, 1, 64, 255, s), PRD_ERR_ARG);                    /* odd time_dim */
    /* entries added in round 2: null / unsupported arguments are refused before any HIP call */
    EXPECT(prd_atom_embed(0, ibuf, p, p, (const int*)ibuf, 9, 1, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_single_init(0, p, p, p, p, 8, 64, 21, s), PRD_ERR_ARG);
    EXPECT(prd_step_boundary(0, p, ibuf, p, p, p, p, p, p, p, p, p, p, p, p, (int*)ibuf, 1, 8, 21, 10, 64, 64, 256, 0, 0, 0, 0, s), PRD_ERR_ARG);
    EXPECT(prd_step_boundary(p, p, ibuf, p, p, p, p, p, p, p, p, p, p, p, p, (int*)ibuf, 1, 8, 21, 10, 64, 64, 256, p, 62, p, 64, s), PRD_ERR_ALIGN);   /* hidden rows closer than their width */
    EXPECT(prd_pair_transition(0, p, p, p, p, p, 1, 1, 8, 64, 0, A1, s), PRD_ERR_ARG);
    EXPECT(prd_pair_transition(p, p, p, p, p, p, 1, 1, 8, 48, 0, A0, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn_out(0, p, p, p, p, 1, 1, 8, 64, 0, A1, s), PRD_ERR_ARG);
    EXPECT(prd_tri_mul_contract(0, p, 1, 8, 64, A1, s), PRD_ERR_ARG);
    EXPECT(prd_tri_mul_contract(p, p, 1, 8, 48, A0, s), PRD_ERR_UNSUPPORTED);
    {
        const float* w8[8] = {p, p, p, p, p, p, p, p};
        const float* w7[8] = {p, p, p, 0, p, p, p, p};
        EXPECT(prd_tri_mul_chain_supported(320, 64, A1), 1);
        EXPECT(prd_tri_mul_chain_supported(320, 48, A1), 0);
        EXPECT(prd_tri_mul_chain_supported(320, 64, A0), 0);        /* fp32 arithmetic has no fused chain */
        EXPECT(prd_tri_mul_chain(0, p, w8, w8, 1, 8, 64, p, 1 << 20, A1, s), PRD_ERR_ARG);
        EXPECT(prd_tri_mul_chain(p, p, w8, w7, 1, 8, 64, p, 1 << 20, A1, s), PRD_ERR_ARG);          /* a missing weight pointer */
        EXPECT(prd_tri_mul_chain(p, p, w8, w8, 1, 8, 64, p, 16, A1, s), PRD_ERR_WORKSPACE);
        EXPECT(prd_tri_mul_chain(p, p, w8, w8, 1, 8, 64, p, 1 << 20, A0, s), PRD_ERR_UNSUPPORTED);     /* fp32 arithmetic has no fused chain */
        EXPECT(prd_tri_mul_chain(p, p, w8, w8, 1, 8, 64, p, 16, A1 | PRD_TUNE(PRD_TUNE_TMS_NW16), s), PRD_ERR_WORKSPACE);  /* switches ride above the arithmetic */
    }
    {   /* the persistent attention pair: every rejection below happens before any HIP call */
        const float* w7[7] = {p, p, p, p, p, p, p};
        const float* w6[7] = {p, p, p, p, p, 0, p};
        const float* w5[5] = {p, p, p, p, p};
        EXPECT(prd_tri_attn_pair_supported(320, 64, A1), 1);
        EXPECT(prd_tri_attn_pair_supported(320, 64, A0), 0);            /* split-16 arithmetic only */
        EXPECT(prd_tri_attn_pair_supported(769, 64, A1), 0);            /* long rows stay three launches */
        EXPECT(prd_tri_attn_pair_supported(320, 48, A1), 0);
        EXPECT(prd_tri_attn_pair(0, p, p, w7, w5, 1, 320, 64, 4, 16, (unsigned*)ibuf, A1, s), PRD_ERR_ARG);
        EXPECT(prd_tri_attn_pair(p, p, p, w6, w5, 1, 320, 64, 4, 16, (unsigned*)ibuf, A1, s), PRD_ERR_ARG);     /* a missing weight pointer */
        EXPECT(prd_tri_attn_pair(p, p, p, w7, w5, 1, 320, 64, 4, 16, 0, A1, s), PRD_ERR_ARG);                    /* no barrier words */
        EXPECT(prd_tri_attn_pair(p, p, p, w7, w5, 1, 320, 64, 2, 32, (unsigned*)ibuf, A1, s), PRD_ERR_UNSUPPORTED);   /* head layout */
        EXPECT(prd_tri_attn_pair(p, p, p, w7, w5, 1, 320, 64, 4, 16, (unsigned*)ibuf, A0, s), PRD_ERR_UNSUPPORTED);
        EXPECT(prd_tri_attn_pair(p, p, p, w7, w5, 1, 769, 64, 4, 16, (unsigned*)ibuf, A1, s), PRD_ERR_UNSUPPORTED);
    }
    EXPECT(prd_tri_attn_core_fused_supported(320, 64, A1), 0);      /* first-generation form: only in the -DPRD_AB library */
    EXPECT(prd_tri_attn_core_fused_supported(769, 64, A1), 0);      /* long rows: no fused form */
    EXPECT(prd_tri_attn_v2_supported(320, 64, 0), 1);
    EXPECT(prd_tri_attn_v2_supported(400, 64, 0), 1);                  /* long-row form */
    EXPECT(prd_tri_attn_v2_supported(1024, 64, 0), 1);
    EXPECT(prd_tri_attn_v2_supported(1025, 64, 0), 0);                 /* more than 32 key tiles */
    EXPECT(prd_tri_attn_v2_form(320, 64, 0), 2);                       /* two K / V buffers fit: overlapped phases */
    EXPECT(prd_tri_attn_v2_form(352, 64, 0), 1);                       /* 11 blocks, three shared: the second buffer does not fit */
    EXPECT(prd_tri_attn_v2_form(769, 64, 0), 3);
    EXPECT(prd_tri_attn_v2_form(1100, 64, 0), 0);
    /* kernel-selection switches are ARGUMENTS (the library reads no environment variable and keeps no state) */
    EXPECT(prd_tri_attn_v2_form(320, 64, PRD_TUNE_TA2_NO_V3), 1);
    EXPECT(prd_tri_attn_v2_supported(769, 64, PRD_TUNE_TA2_NO_LONG), 0);
    EXPECT(prd_tri_attn_variant(1000, 64, A1 | PRD_TUNE(PRD_TUNE_TA2_NO_LONG)), 3);   /* first generation: K / V of 1000 positions do not fit -> key-chunked */
    EXPECT(prd_tri_attn_variant(1000, 64, A1 | PRD_TUNE(10)), 3);                     /* PRD_TUNE_TA_VARIANT 10: first generation */
    EXPECT(prd_tri_attn_variant(1000, 64, A1), 2);                                    /* ... and the default is unchanged by those calls */
    EXPECT(prd_tri_attn_variant(320, 64, -1), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_variant(960, 64, A1), 2);
    EXPECT(prd_tri_attn_variant(960, 64, A0), 1);                   /* last length whose K / V fit the LDS in fp32 (fp32 long-row kernel) */
    EXPECT(prd_tri_attn_variant(961, 64, A0), 3);                   /* fp32 arithmetic: key-chunked */
    EXPECT(prd_tri_attn_variant(961, 64, A1), 2);                   /* split-16: K / V as fp16 planes fit up to N = 1024 */
    EXPECT(prd_tri_attn_variant(1024, 64, A1), 2);
    EXPECT(prd_tri_attn_variant(1025, 64, A1), 3);
    EXPECT((int)prd_tri_attn_stats_bytes(1, 960, 64, 4, A1), 0);
    EXPECT((int)prd_tri_attn_stats_bytes(1, 1000, 64, 4, A1), 0);
    EXPECT((int)(prd_tri_attn_stats_bytes(1, 1000, 64, 4, A0) != (size_t)1000 * 1000 * 4 * 2 * 4), 0);
    EXPECT((int)(prd_tri_attn_stats_bytes(1, 1100, 64, 4, A1) != (size_t)1100 * 1100 * 4 * 2 * 4), 0);
    EXPECT((int)(prd_workspace_bytes("tri_attn", 1, 1000, 0, 64) != (size_t)1000 * 1000 * (64 + 8) * 4), 0);
    EXPECT(prd_tri_attn_core(p, p, p, p, p, p, p, p, 0, 1, 1000, 64, 4, 16, A0, s), PRD_ERR_UNSUPPORTED);   /* needs the statistics buffer */
    EXPECT(prd_tri_attn_core(p, p, p, p, p, p, p, p, 0, 1, 1100, 64, 4, 16, A1, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn_core_chunked(p, p, p, p, p, p, p, p, 0, 1, 1000, 64, 4, 16, 0, 0, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_core_chunked(p, p, p, p, p, p, p, p, 0, 1, 1000, 64, 4, 16, p, 64, s), PRD_ERR_WORKSPACE);
    EXPECT(prd_tri_attn_core_chunked(p, p, p, p, p, p, p, p, 0, 1, 1000, 48, 4, 16, p, (size_t)1 << 30, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn_core_v2(0, p, p, p, p, p, p, p, 0, 1, 8, 64, 4, 16, 0, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_core_v2(p, p, p, p, p, p, p, p, 0, 1, 1100, 64, 4, 16, 0, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_attn_core_fused(p, p, p, p, p, p, p, p, p, p, p, p, 1, 1, 8, 64, 4, 16, s), PRD_ERR_ARG);   /* pair_out aliases pair */
    EXPECT(prd_tri_mul_out_bwd(0, p, p, p, p, p, p, p, p, p, p, p, p, 0, 0, 0, 1, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_tri_mul_out_bwd(p, p, p, p, p, p, p, p, p, p, p, p, p, 0, 0, 32, 1, 8, 64, s), PRD_ERR_ARG);     /* batch pitch below P */
    EXPECT(prd_tri_mul_bwd_operands(0, p, 1, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_tri_mul_bwd_operands(p, p, 1, 8, 48, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_tri_mul_proj_bwd(0, p, p, p, p, p, p, p, p, p, p, p, p, 0, 1, 8, 64, A1, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_bwd_core(0, p, p, p, p, p, p, p, p, 0, 1, 8, 64, 4, 16, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_bwd_core(p, p, p, p, p, p, p, p, p, 0, 1, 100000, 64, 4, 16, s), PRD_ERR_UNSUPPORTED);   /* row beyond the LDS */
    EXPECT(prd_tri_attn_bwd_core_v2(0, p, p, p, p, p, p, p, p, p, 0, 0, 0, 1, 8, 64, 4, 16, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_bwd_core_v2(p, p, p, p, p, p, p, p, p, p, 0, 0, 0, 1, 385, 64, 4, 16, s), PRD_ERR_UNSUPPORTED);  /* more than 12 blocks */
    EXPECT(prd_tri_attn_core_v2_lse(0, 0, p, p, p, p, p, p, p, 0, 1, 8, 64, 4, 16, 0, s), PRD_ERR_ARG);
    EXPECT(prd_tri_attn_core_v2_lse(p, p, p, p, p, p, p, p, p, 0, 1, 769, 64, 4, 16, 0, s), PRD_ERR_UNSUPPORTED);       /* statistics: short rows only */
    EXPECT(prd_tri_attn_bwd_core_v2_supported(384, 64), 1);
    EXPECT(prd_tri_attn_bwd_core_v2_supported(385, 64), 0);
    EXPECT(prd_ln_rows_bwd(0, p, p, 0, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_pair_bias_bwd(0, p, p, p, p, p, 1, 64, 4, 64, s), PRD_ERR_ARG);
    EXPECT(prd_pair_bias_bwd(p, p, p, p, p, p, 1, 64, 3, 64, s), PRD_ERR_UNSUPPORTED);      /* H = 4 or 8 */
    EXPECT(prd_pair_bias_bwd(p, p, p, p, p, p, 1, 64, 4, 32, s), PRD_ERR_UNSUPPORTED);      /* P = 64 */
    EXPECT(prd_pair_linear(0, p, p, p, 100, 64, 64, 0, 0, 0, 0, 0, 1, s), PRD_ERR_ARG);
    EXPECT(prd_pair_linear(p, p, p, p, 100, 64, 128, 0, 0, 0, 0, 0, 1, s), PRD_ERR_UNSUPPORTED);        /* (K, OUT) not served */
    EXPECT(prd_pair_linear(p, p, p, p, 100, 64, 64, 0, 0, 0, 0, 0, 0, s), PRD_ERR_UNSUPPORTED);          /* fp32 arithmetic: the caller's GEMM */
    EXPECT(prd_pair_linear(p, p, p, p, 100, 256, 64, 1, 0, 0, 0, 0, 1, s), PRD_ERR_ARG);                 /* LayerNorm of 256-wide rows */
    EXPECT(prd_pair_linear(p, p + 1, p, p, 100, 64, 64, 0, 0, 0, 0, 0, 1, s), PRD_ERR_ALIGN);
    {
        const long long* ids[2] = {(const long long*)ibuf, (const long long*)ibuf};
        int cards[2] = {8, 200};
        EXPECT(prd_embed_wgrad_multi(0, ids, 0, cards, 2, p, 100, 64, 64, p, 1 << 20, s), PRD_ERR_ARG);
        EXPECT(prd_embed_wgrad_multi(p, ids, 0, cards, 2, p, 100, 64, 64, p, 1 << 20, s), PRD_ERR_UNSUPPORTED);      /* more than 128 rows in total */
        cards[1] = 8;
        EXPECT(prd_embed_wgrad_multi(p, ids, 0, cards, 2, p, 100, 64, 64, p, 16, s), PRD_ERR_WORKSPACE);
    }
    EXPECT(prd_rbf_rows(0, p, p, p, 1, 8, 256, s), PRD_ERR_ARG);
    EXPECT(prd_rbf_rows(p, p, p, p, 1, 8, 30, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_sym_transpose(0, p, 1, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_sym_rows(0, p, 0.5f, 1, 8, 64, s), PRD_ERR_ARG);
    EXPECT(prd_outer_linear_bwd_reduce(0, p, 4, p, p, p, 16, 64, 512, s), PRD_ERR_ARG);
    EXPECT(prd_outer_linear_bwd_reduce(p, p, 0, p, p, p, 16, 64, 512, s), PRD_ERR_ARG);
    EXPECT(prd_sym_rows(p, p, 0.5f, 1, 8, 64, s), PRD_ERR_UNSUPPORTED);                                    /* in place */
    EXPECT(prd_sym_transpose(p, p, 1, 8, 48, s), PRD_ERR_UNSUPPORTED);
    EXPECT(prd_pair_linear_supported(256, 64, 1), 1);
    EXPECT(prd_pair_linear_supported(256, 256, 1), 0);
    EXPECT((int)(prd_linear_wgrad_workspace(102400, 256, 64) != (size_t)200 * (256 * 64 + 256) * 4), 0);
    EXPECT((int)prd_linear_wgrad_workspace(0, 256, 64), 0);
    EXPECT(prd_linear_wgrad(0, p, p, p, 100, 64, 64, 64, 64, p, 1 << 20, 0, s), PRD_ERR_ARG);
    EXPECT(prd_linear_wgrad(p, p, p, p, 100, 96, 64, 96, 64, p, 1 << 20, 0, s), PRD_ERR_UNSUPPORTED);       /* O neither <= 16 nor a multiple of 64 */
    EXPECT(prd_linear_wgrad(p, p, p, p, 100, 64, 64, 65, 64, p, 1 << 20, 0, s), PRD_ERR_ALIGN);            /* odd row pitch */
    EXPECT(prd_linear_wgrad(p, p, p, p, 100, 64, 64, 64, 64, p, 16, 0, s), PRD_ERR_WORKSPACE);
    EXPECT(prd_linear_wgrad(p, p, p + 1, p, 100, 64, 64, 64, 64, p, 1 << 20, 0, s), PRD_ERR_ALIGN);        /* dy 4-byte aligned only: the wide kernel loads 8 bytes */
    EXPECT(prd_linear_wgrad(p, p, p, p + 1, 100, 64, 64, 64, 64, p, 1 << 20, 0, s), PRD_ERR_ALIGN);
    EXPECT(prd_linear_wgrad(p, p, p, p, 100, 64, 64, 64, 64, p + 1, 1 << 20, 0, s), PRD_ERR_ALIGN);        /* partials workspace */
    EXPECT(prd_linear_wgrad(p, p, p + 1, p, 100, 4, 64, 4, 64, p, 16, 0, s), PRD_ERR_WORKSPACE);            /* narrow form: scalar loads, no alignment demand */
    EXPECT(prd_embed_wgrad(0, (const long long*)ibuf, p, 0, 100, 8, 64, 64, p, 1 << 20, s), PRD_ERR_ARG);
    EXPECT(prd_embed_wgrad(p, (const long long*)ibuf, p, 0, 100, 200, 64, 64, p, 1 << 20, s), PRD_ERR_UNSUPPORTED);   /* more than 128 table rows */
    EXPECT(prd_embed_wgrad(p, (const long long*)ibuf, p, 0, 100, 8, 64, 64, p, 16, s), PRD_ERR_WORKSPACE);
    EXPECT((int)(prd_embed_wgrad_workspace(102400, 65, 64) != (size_t)200 * 65 * 64 * 4), 0);
    EXPECT((int)(prd_workspace_bytes("tri_mul", 1, 320, 512, 64) != (size_t)3 * 64 * 320 * 320 * 4), 0);
    EXPECT((int)prd_workspace_bytes("nonsense", 1, 320, 512, 64), 0);
    EXPECT((int)prd_workspace_bytes(0, 1, 320, 512, 64), 0);
    {
        struct thread_arg t0 = {PRD_ARITH_FP32, 3, 0, 1, 0}, t1 = {PRD_ARITH_SPLIT16, 2, 1, 0, 0},
                          t2 = {PRD_ARITH_SPLIT16 | PRD_TUNE(PRD_TUNE_TA2_NO_LONG), 3, 1, 1, 0};
        pthread_t th[3];
        pthread_create(&th[0], 0, hammer, &t0);
        pthread_create(&th[1], 0, hammer, &t1);
        pthread_create(&th[2], 0, hammer, &t2);
        for (int k = 0; k < 3; ++k) pthread_join(th[k], 0);
        EXPECT(t0.bad + t1.bad + t2.bad, 0);
    }
    printf(failures ? "host ABI check: %d failure(s)\n" : "host ABI check: OK\n", failures);
    return failures ? 1 : 0;
}
