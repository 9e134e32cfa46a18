// Can the fp32 pointwise GEMMs run on the f16 matrix pipe without giving up fp32 accuracy?  x = x_hi + x_lo with both
// halves in f16 (22 mantissa bits together), products x_hi.w_hi + x_hi.w_lo + x_lo.w_hi on v_mfma_f32_32x32x16_f16 (exact
// products, fp32 accumulation): 3 x 32 cycles per 16 k instead of 8 x 64 cycles on v_mfma_f32_32x32x2_f32.
// This file measures, on the long-K shapes of the 8x13 level, (1) whether the f16 pipe keeps f16 subnormals, (2) the error
// of both forms against a float64 result, (3) their speed in the split-K form of k_pw_sk.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/f16x3_gemm.hip -o /tmp/f16x3 && /tmp/f16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- fp32 MFMA, split-K over four waves (the library's k_pw_sk without the software pipeline) ----
template <int TN>
__global__ __launch_bounds__(256) void k_f32(const float *__restrict__ X, const float *__restrict__ Wt, float *__restrict__ Y,
                                             int M, int N, int K) {
    __shared__ float red[4][TN][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * (32 * TN);
    const float *xp = X + (size_t)min(m0 + r, M - 1) * K + 4 * hh;
    f32x16 acc[TN];
    for (int t = 0; t < TN; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nsteps = K >> 3, s_lo = (wave * nsteps) >> 2, s_hi = ((wave + 1) * nsteps) >> 2;
    for (int st = s_lo; st < s_hi; ++st) {
        const float4 a = *(const float4 *)(xp + 8 * st);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float4 b = *(const float4 *)(Wt + (size_t)(n0 + t * 32 + r) * K + 4 * hh + 8 * st);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][t][i][lane] = acc[t][i];
    __syncthreads();
    const int rr = m0 + r, g = wave;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + t * 32 + 8 * g + 4 * hh;
        if (col >= N || rr >= M) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = ((red[0][t][4 * g + j][lane] + red[1][t][4 * g + j][lane]) + red[2][t][4 * g + j][lane]) + red[3][t][4 * g + j][lane];
        *(float4 *)(Y + (size_t)rr * N + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ void split8(const float4 a0, const float4 a1, h8 &hi, h8 &lo) {
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { hi[i] = (_Float16)a[i]; lo[i] = (_Float16)(a[i] - (float)hi[i]); }
}

// ---- the same kernel on the f16 pipe: weights pre-split on the host (Whi / Wlo, [N][K] f16), activations split in registers ----
template <int TN, int NPROD>      // NPROD = 3: hi.hi + hi.lo + lo.hi;  4: + lo.lo;  1: hi.hi only (plain f16, for reference)
__global__ __launch_bounds__(256) void k_f16s(const float *__restrict__ X, const _Float16 *__restrict__ Whi,
                                              const _Float16 *__restrict__ Wlo, float *__restrict__ Y, int M, int N, int K) {
    __shared__ float red[4][TN][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * (32 * TN);
    const float *xp = X + (size_t)min(m0 + r, M - 1) * K + 8 * hh;
    f32x16 acc[TN];
    for (int t = 0; t < TN; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nsteps = K >> 4, s_lo = (wave * nsteps) >> 2, s_hi = ((wave + 1) * nsteps) >> 2;
    for (int st = s_lo; st < s_hi; ++st) {
        h8 ahi, alo;
        split8(*(const float4 *)(xp + 16 * st), *(const float4 *)(xp + 16 * st + 4), ahi, alo);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const size_t wo = (size_t)(n0 + t * 32 + r) * K + 8 * hh + 16 * st;
            const h8 bhi = *(const h8 *)(Whi + wo), blo = *(const h8 *)(Wlo + wo);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi, ahi, acc[t], 0, 0, 0);
            if (NPROD >= 3) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi, alo, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, ahi, acc[t], 0, 0, 0);
            }
            if (NPROD >= 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, alo, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][t][i][lane] = acc[t][i];
    __syncthreads();
    const int rr = m0 + r, g = wave;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + t * 32 + 8 * g + 4 * hh;
        if (col >= N || rr >= M) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = ((red[0][t][4 * g + j][lane] + red[1][t][4 * g + j][lane]) + red[2][t][4 * g + j][lane]) + red[3][t][4 * g + j][lane];
        *(float4 *)(Y + (size_t)rr * N + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// subnormal probe: one 32x32x16 product with a = 2^-20 (an f16 subnormal) in every slot, b = 1: 16 * 2^-20 expected
__global__ void k_denorm(float *out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}

static double rnd() { return (double)rand() / RAND_MAX; }

struct Err { double max_abs, rms, max_rel_to_scale; };
static Err cmp(const std::vector<float> &y, const std::vector<double> &ref) {
    double ma = 0, s2 = 0, sc = 0;
    for (size_t i = 0; i < y.size(); ++i) { const double d = fabs((double)y[i] - ref[i]); ma = fmax(ma, d); s2 += d * d; sc = fmax(sc, fabs(ref[i])); }
    return Err{ma, sqrt(s2 / y.size()), ma / sc};
}

template <int TN>
static void run_shape(int M, int K, int N, const char *what, int dist) {
    std::vector<float> X((size_t)M * K), W((size_t)N * K), Y((size_t)M * N);
    std::vector<_Float16> Whi(W.size()), Wlo(W.size());
    srand(1234 + K);
    for (auto &v : X) {
        // dist 0: ReLU6-like activations (half zeros, the rest in (0, 6)); 1: wide dynamic range incl. tiny values
        if (dist == 0) { const double u = rnd(); v = u < 0.5 ? 0.f : (float)(6.0 * rnd() * rnd()); }
        else v = (float)((rnd() - 0.5) * pow(10.0, -6.0 * rnd()) * 20.0);
    }
    for (auto &v : W) v = (float)((rnd() - 0.5) * 2.0 / sqrt((double)K) * (dist ? pow(10.0, -4.0 * rnd()) * 30 : 1.0));
    for (size_t i = 0; i < W.size(); ++i) { Whi[i] = (_Float16)W[i]; Wlo[i] = (_Float16)(W[i] - (float)Whi[i]); }
    std::vector<double> ref((size_t)M * N);
    const int MR = std::min(M, 256);          // rows checked against float64
    for (int m = 0; m < MR; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
            ref[(size_t)m * N + n] = s;
        }
    ref.resize((size_t)MR * N);
    float *dX, *dW, *dY;
    _Float16 *dh, *dl;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMalloc(&dh, W.size() * 2)); CK(hipMalloc(&dl, W.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, Whi.data(), W.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, Wlo.data(), W.size() * 2, hipMemcpyHostToDevice));
    dim3 grid((M + 31) / 32, N / (32 * TN));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.f / 50;
    };
    auto check = [&](const char *name, float us) {
        CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
        std::vector<float> y(Y.begin(), Y.begin() + (size_t)MR * N);
        const Err e = cmp(y, ref);
        printf("  %-22s %7.1f us  %6.1f TFLOP/s   err vs f64: max %.3e  rms %.3e  max/scale %.3e\n", name, us,
               2.0 * M * N * K / us * 1e-6, e.max_abs, e.rms, e.max_rel_to_scale);
    };
    printf("%s  M=%d K=%d N=%d  (%s inputs)\n", what, M, K, N, dist ? "wide-range" : "ReLU6-like");
    float us = time_it([&] { k_f32<TN><<<grid, 256>>>(dX, dW, dY, M, N, K); });
    check("f32 MFMA 32x32x2", us);
    us = time_it([&] { k_f16s<TN, 3><<<grid, 256>>>(dX, dh, dl, dY, M, N, K); });
    check("f16 split, 3 products", us);
    us = time_it([&] { k_f16s<TN, 4><<<grid, 256>>>(dX, dh, dl, dY, M, N, K); });
    check("f16 split, 4 products", us);
    us = time_it([&] { k_f16s<TN, 1><<<grid, 256>>>(dX, dh, dl, dY, M, N, K); });
    check("plain f16 (1 product)", us);
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dY)); CK(hipFree(dh)); CK(hipFree(dl));
}

int main() {
    float *d, h[2];
    CK(hipMalloc(&d, 8));
    k_denorm<<<1, 64>>>(d);
    CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("f16 subnormal through the matrix pipe: 16 * 2^-20 = %.6e expected, got %.6e (operand reads back as %.6e)\n",
           16 * 9.5367431640625e-07, h[0], h[1]);
    for (int dist = 0; dist < 2; ++dist) {
        run_shape<2>(3328, 320, 1280, "features.18", dist);
        run_shape<1>(3328, 960, 160, "project 960->160", dist);
        run_shape<1>(13312, 576, 96, "project 576->96", dist);
    }
    return 0;
}
