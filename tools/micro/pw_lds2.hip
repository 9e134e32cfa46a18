// Would the long-K pointwise layers of the saliency network gain from the operand path of the TransNet kernels (both
// operands through LDS with whole-line loads, csrc/svc_shot.hip: k_shot_conv_lds2)?  The library runs them through k_pw_sk
// (operands straight from global memory, K split over the four waves).  Shapes of the 8x13 / 16x26 levels at B = 32.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/pw_lds2.hip -o /tmp/pw_lds2 && /tmp/pw_lds2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// Y[M][N] = X[M][K] . W[N][K]^T ; workgroup = 128 rows x NT*32 columns; K in slices of 64 (K % 64 == 0)
template <int NT>
__global__ __launch_bounds__(256) void k_pw_lds2(const float *__restrict__ X, const float *__restrict__ Wt, float *__restrict__ Y, int M, int N, int K) {
    extern __shared__ float sm[];
    constexpr int WS = 68;
    float *As = sm, *Bs = sm + 128 * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 32 * NT;
    const int c4 = tid & 15, row0 = tid >> 4, nsl = K >> 6;
    float4 areg[8], breg[2 * NT];
    auto fetch = [&](int sl) {
#pragma unroll
        for (int j = 0; j < 8; ++j) areg[j] = *(const float4 *)(X + (size_t)min(m0 + row0 + 16 * j, M - 1) * K + sl * 64 + c4 * 4);
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) breg[j] = *(const float4 *)(Wt + (size_t)min(n0 + row0 + 16 * j, N - 1) * K + sl * 64 + c4 * 4);
    };
    f32x16 acc[NT];
    for (int n = 0; n < NT; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    fetch(0);
    for (int sl = 0; sl < nsl; ++sl) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) *(float4 *)(As + (row0 + 16 * j) * WS + c4 * 4) = areg[j];
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) *(float4 *)(Bs + (row0 + 16 * j) * WS + c4 * 4) = breg[j];
        __syncthreads();
        if (sl + 1 < nsl) fetch(sl + 1);
        const float *ap = As + (wave * 32 + r) * WS + 4 * hh;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const float4 a = *(const float4 *)(ap + 8 * st);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b = *(const float4 *)(Bs + (n * 32 + r) * WS + 4 * hh + 8 * st);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[n], 0, 0, 0);
            }
        }
    }
    const int m = m0 + wave * 32 + r;
    if (m >= M) return;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + n * 32 + 8 * g + 4 * hh;
            if (col >= N) continue;
            *(float4 *)(Y + (size_t)m * N + col) = make_float4(acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]);
        }
}

template <int NT>
static void run(int M, int K, int N, const char *what, float lib_us) {
    std::vector<float> X((size_t)M * K), W((size_t)N * K), Y((size_t)M * N);
    srand(K + N);
    for (auto &v : X) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto &v : W) v = ((float)rand() / RAND_MAX * 2 - 1) / sqrtf((float)K);
    float *dX, *dW, *dY;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    dim3 grid((M + 127) / 128, (N + 32 * NT - 1) / (32 * NT));
    const size_t lds = (128 + 32 * NT) * 68 * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) k_pw_lds2<NT><<<grid, 256, lds>>>(dX, dW, dY, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) k_pw_lds2<NT><<<grid, 256, lds>>>(dX, dW, dY, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0;
    for (int m = 0; m < M; m += 97) for (int n = 0; n < N; n += 13) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k];
        maxd = fmax(maxd, fabs(s - Y[(size_t)m * N + n]));
    }
    const float us = ms * 1000 / 50;
    printf("%-28s M=%5d K=%4d N=%4d  NT=%d  %3d workgroups  %6.1f us  %5.1f TFLOP/s  (library k_pw_sk: %.1f us)  max err %.1e\n", what, M, K, N, NT,
           grid.x * grid.y, us, 2.0 * M * N * K / us * 1e-6, lib_us, maxd);
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dY));
}

int main() {
    run<2>(3328, 320, 1280, "features.18", 44.7f);
    run<1>(3328, 320, 1280, "features.18", 44.7f);
    run<1>(3328, 960, 160, "block 15/16 project", 24.2f);
    run<2>(3328, 960, 320, "block 17 project", 38.0f);
    run<1>(3328, 960, 320, "block 17 project", 38.0f);
    run<2>(3328, 256, 768, "decoder T1", 25.2f);
    run<1>(3328, 256, 768, "decoder T1", 25.2f);
    run<1>(13312, 320, 128, "skip_2x reduction", 21.7f);
    run<2>(13312, 320, 128, "skip_2x reduction", 21.7f);
    run<1>(13312, 576, 96, "blocks 12-13 project (fused)", 0.f);
    return 0;
}
