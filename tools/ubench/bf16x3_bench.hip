// Micro-benchmark for a possible next step: the row GEMM (weights in LDS, the lane's row in registers) on the bf16
// matrix pipe with a three-way split of both operands (x = hi + mid + lo, each a bf16 by TRUNCATION, so the split of
// an fp32 value is exact; products kept: hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -> error ~2^-24 relative, fp32
// accumulate) against the fp32 v_mfma_f32_32x32x2_f32 form used today.  The split of the row costs VALU instructions,
// which on gfx950 are matrix-pipe time -- both are inside the timed loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16x3_bench.hip -o bf16x3_bench
#include <cstdio>
#include <vector>
#include <cmath>
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 64, NBLK = 2;       // K channels in, 32 * NBLK channels out per row GEMM

// ---- fp32 reference form: A = W[32 outputs][K] from LDS (pitch K+4), B = the lane's 32 channels ----
__device__ __forceinline__ void rowgemm_f32(const float* Wl, const float (&x)[K / 2], f32x16 (&acc)[NBLK], int r, int hi) {
#pragma unroll
    for (int m = 0; m < K / 8; ++m) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const float4 w = *reinterpret_cast<const float4*>(Wl + (nb * 32 + r) * (K + 4) + hi * (K / 2) + 4 * m);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x[4 * m + 0], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x[4 * m + 1], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x[4 * m + 2], acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x[4 * m + 3], acc[nb], 0, 0, 0);
        }
    }
}

// ---- bf16 x 3 form.  LDS image: Wb[plane 3][out 32*NBLK] rows of (K/16 steps x 2 halves + 1 pad) x 16 bytes (8 bf16) ----
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {          // (a, b) truncated to bf16, a in the low half
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
__device__ __forceinline__ void split3(const float (&x)[K / 2], u32x4 (&p)[3][K / 16]) {
    // lane's 32 channels -> 3 planes x (K/16 steps) x 8 bf16; step s uses x[8s .. 8s+7]
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = x[8 * s + 2 * q], b = x[8 * s + 2 * q + 1];
            const float ah = __uint_as_float(__float_as_uint(a) & 0xffff0000u), bh = __uint_as_float(__float_as_uint(b) & 0xffff0000u);
            const float ar = a - ah, br = b - bh;
            const float am = __uint_as_float(__float_as_uint(ar) & 0xffff0000u), bm = __uint_as_float(__float_as_uint(br) & 0xffff0000u);
            const float al = ar - am, bl = br - bm;
            p[0][s][q] = pack_hi16(ah, bh);
            p[1][s][q] = pack_hi16(am, bm);
            p[2][s][q] = pack_hi16(al, bl);
        }
}
__device__ __forceinline__ void rowgemm_bf16x3(const u32x4* Wb, const u32x4 (&p)[3][K / 16], f32x16 (&acc)[NBLK], int r, int hi) {
    constexpr int S = K / 16;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            u32x4 w[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[pl] = Wb[(pl * 32 * NBLK + nb * 32 + r) * (2 * S + 1) + 2 * s + hi];   // pitch 2S+1: conflict-free b128 reads
            // (weight plane, row plane): (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)
            const int wp[6] = {0, 0, 1, 0, 2, 1}, xp[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[wp[t]]), __builtin_bit_cast(bf16x8, p[xp[t]][s]),
                                                                   acc[nb], 0, 0, 0);
        }
}

template <int MODE, int NW>       // 0: fp32, 1: bf16x3 (split inside the loop), 2: bf16x3 MFMAs only (row split hoisted)
__global__ __launch_bounds__(NW * 64) void bench(float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 3 * 32 * NBLK * (2 * (K / 16) + 1) * 4; i += NW * 64) smem[i] = 0.001f * (i % 977);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    float x[K / 2];
#pragma unroll
    for (int s = 0; s < K / 2; ++s) x[s] = 0.01f * (lane + s) + 0.5f;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
    u32x4 p[3][K / 16];
    split3(x, p);
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0) rowgemm_f32(smem, x, acc, r, hi);
        else {
            if (MODE == 1) split3(x, p);
            rowgemm_bf16x3(reinterpret_cast<const u32x4*>(smem), p, acc, r, hi);
        }
        x[it & 31] += acc[0][0] * 1e-30f;          // keep a dependence so nothing is hoisted
    }
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[nb][q];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

// accuracy: one row GEMM of random data through both forms, against a double-precision host reference
__global__ __launch_bounds__(64) void accuracy(const float* W, const float* X, float* out32, float* out16) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                            // fp32 image [64][K+4], K order = lane layout
    u32x4* Wb = reinterpret_cast<u32x4*>(smem + 64 * (K + 4));   // bf16x3 image
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    // lane (r, hi) owns channels ch(s) = 8*(s>>2) + 4*hi + (s&3), s = 0..31, of row r
    for (int o = lane; o < 64; o += 64)
        for (int h = 0; h < 2; ++h)
            for (int s2 = 0; s2 < K / 2; ++s2) Wl[o * (K + 4) + h * (K / 2) + s2] = W[o * K + 8 * (s2 >> 2) + 4 * h + (s2 & 3)];
    for (int o = lane; o < 64; o += 64)
        for (int st = 0; st < K / 16; ++st)
            for (int h = 0; h < 2; ++h) {
                unsigned pk[3][4];
                for (int q = 0; q < 4; ++q) {
                    float v[2], hh[2], mm[2], ll[2];
                    for (int e = 0; e < 2; ++e) {
                        const int s2 = 8 * st + 2 * q + e;
                        v[e] = W[o * K + 8 * (s2 >> 2) + 4 * h + (s2 & 3)];
                        hh[e] = __uint_as_float(__float_as_uint(v[e]) & 0xffff0000u);
                        const float r1 = v[e] - hh[e];
                        mm[e] = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                        ll[e] = r1 - mm[e];
                    }
                    pk[0][q] = pack_hi16(hh[0], hh[1]); pk[1][q] = pack_hi16(mm[0], mm[1]); pk[2][q] = pack_hi16(ll[0], ll[1]);
                }
                for (int pl = 0; pl < 3; ++pl) Wb[(pl * 64 + o) * (2 * (K / 16) + 1) + 2 * st + h] = u32x4{pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]};
            }
    __syncthreads();
    float x[K / 2];
    for (int s2 = 0; s2 < K / 2; ++s2) x[s2] = X[r * K + 8 * (s2 >> 2) + 4 * hi + (s2 & 3)];
    f32x16 a32[NBLK], a16[NBLK];
    for (int nb = 0; nb < NBLK; ++nb) for (int q = 0; q < 16; ++q) { a32[nb][q] = 0.f; a16[nb][q] = 0.f; }
    rowgemm_f32(Wl, x, a32, r, hi);
    u32x4 p[3][K / 16];
    split3(x, p);
    rowgemm_bf16x3(Wb, p, a16, r, hi);
    for (int nb = 0; nb < NBLK; ++nb)
        for (int q = 0; q < 16; ++q) {
            const int o = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * hi;          // D row of register q
            out32[r * 64 + o] = a32[nb][q];
            out16[r * 64 + o] = a16[nb][q];
        }
}

void check_accuracy() {
    std::vector<float> W(64 * K), X(32 * K), o32(32 * 64), o16(32 * 64);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : W) v = rnd();
    for (auto& v : X) v = rnd() * 3.0f;
    float *dW, *dX, *d32, *d16;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&d32, o32.size() * 4); hipMalloc(&d16, o16.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 64 * (K + 4) * 4 + 3 * 64 * (2 * (K / 16) + 1) * 16;
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), lds, 0, dW, dX, d32, d16);
    hipMemcpy(o32.data(), d32, o32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o16.data(), d16, o16.size() * 4, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, nrm = 0;
    for (int r = 0; r < 32; ++r)
        for (int o = 0; o < 64; ++o) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)W[o * K + k] * (double)X[r * K + k];
            e32 += (o32[r * 64 + o] - ref) * (o32[r * 64 + o] - ref);
            e16 += (o16[r * 64 + o] - ref) * (o16[r * 64 + o] - ref);
            nrm += ref * ref;
        }
    printf("relative L2 error vs double: fp32 MFMA %.3e   bf16x3 (6 products) %.3e\n", sqrt(e32 / nrm), sqrt(e16 / nrm));
}

template <int MODE, int NW>
void run(const char* name, int reps) {
    float* out;
    const int grid = 256;
    hipMalloc(&out, grid * NW * 64 * sizeof(float));
    const size_t lds = 3 * 32 * NBLK * (2 * (K / 16) + 1) * 16 + 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<MODE, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<MODE, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * reps * 32.0 * (2.0 * K * 32 * NBLK);      // useful fp32-equivalent flops
    printf("%-34s waves/CU=%2d  %8.3f ms  %7.1f TF/s (fp32-equivalent)\n", name, NW, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    check_accuracy();
    run<0, 4>("fp32 32x32x2", 2000); run<0, 8>("fp32 32x32x2", 2000); run<0, 12>("fp32 32x32x2", 2000);
    run<2, 4>("bf16x3 MFMAs only", 2000); run<2, 8>("bf16x3 MFMAs only", 2000); run<2, 12>("bf16x3 MFMAs only", 2000);
    run<1, 4>("bf16x3 + row split per GEMM", 2000); run<1, 8>("bf16x3 + row split per GEMM", 2000); run<1, 12>("bf16x3 + row split per GEMM", 2000);
    return 0;
}
