// Where does a short-K pointwise layer's time go?  The library's k_pwr on the dominant shapes of the 16x26 / 8x13 levels
// (B = 32: M = 13312; 64 -> 384 and 96 -> 576, ...) against ablations of itself: no stores, no global loads, stores only.
// Measured on MI355X (round 2): 64 -> 384: full 15.6 us = launch + loads 3.4 + MFMAs 4.2 (the algorithmic minimum) +
// stores 6.6 - 2.5 (20 MB at 5 TB/s) -- the three phases run one after the other across the whole chip, because every
// workgroup of the single round starts at the same time; each phase on its own is near its bound.  A variant whose
// epilogue is transposed through LDS (8 whole lines per store instruction instead of 32 pieces of 32 B) is no faster:
// the store path is not the limit, the missing overlap is.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/pw_phases.hip -o /tmp/pw_phases && /tmp/pw_phases
#include "../../retargetvid_amd/csrc/svc_net.hip"

template <int KS, int MODE>     // MODE 1: no stores; 2: no global loads (operands from registers / stale LDS); 3: stores only
__global__ __launch_bounds__(256) void k_ablate(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                                const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N,
                                                int Npad, int ntw, int relu6, int never) {
    constexpr int K = 8 * KS, WS = K + 4;
    extern __shared__ float sm_pwr[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32, n0 = blockIdx.y * (32 * ntw);
    const int ncols = min(32 * ntw, Npad - n0);
    const int rr = m0 + r;
    float4 A[KS];
    if (MODE == 2 || MODE == 3) {
#pragma unroll
        for (int p = 0; p < KS; ++p) A[p] = make_float4(1.f + lane, 2.f, 3.f + p, 4.f);
    } else {
        const float *xp = X + (size_t)min(rr, M - 1) * ldx + 4 * hh;
#pragma unroll
        for (int p = 0; p < KS; ++p) A[p] = *(const float4 *)(xp + 8 * p);
        constexpr int K4 = K / 4;
        for (int i = tid; i < ncols * K4; i += 256) {
            const int row = i / K4, c4 = i - row * K4;
            *(float4 *)(sm_pwr + row * WS + c4 * 4) = *(const float4 *)(Wt + (size_t)(n0 + row) * ldw + c4 * 4);
        }
    }
    __syncthreads();
    if (m0 >= M) return;
    const int nt = ncols >> 5;
    for (int t = 0; t < nt; ++t) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (float)i;
        if (MODE != 3) {
            const float *bq = sm_pwr + (t * 32 + r) * WS + 4 * hh;
#pragma unroll
            for (int p = 0; p < KS; ++p) {
                const float4 b = *(const float4 *)(bq + 8 * p);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, A[p].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, A[p].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, A[p].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, A[p].w, acc, 0, 0, 0);
            }
        }
        if (rr >= M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + t * 32 + 8 * g + 4 * hh;
            if (col >= N) continue;
            float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
            if (MODE == 1) { if (v.x == 12345.f + (float)never) *(float4 *)(Y + (size_t)rr * ldy + col) = v; }      // the MFMAs stay (the test needs their result), the store never happens
            else *(float4 *)(Y + (size_t)rr * ldy + col) = v;
        }
    }
}

template <typename F>
static float time_us(F launch, int iters = 40) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

template <int KS>
static void study(int M, int N) {
    constexpr int K = 8 * KS;
    const int Npad = (N + 31) / 32 * 32, tiles = Npad / 32;
    float *X, *W, *B, *Y;
    hipMalloc(&X, (size_t)M * K * 4); hipMalloc(&W, (size_t)Npad * K * 4); hipMalloc(&B, Npad * 4); hipMalloc(&Y, (size_t)M * N * 4);
    hipMemset(X, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)Npad * K * 4); hipMemset(B, 0, Npad * 4);
    hipFuncSetAttribute((const void *)k_pwr<KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void *)k_ablate<KS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void *)k_ablate<KS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void *)k_ablate<KS, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const UpsAdd ua{nullptr, 0, 0, 0, make_fdiv(1), make_fdiv(1)};
    const double gf = 2.0 * M * K * N * 1e-9, mb = ((double)M * (K + N) + (double)N * K) * 4e-6;
    printf("M=%d K=%d N=%d: %.2f GFLOP (%.1f us at 157.3 TF), %.1f MB in+out (%.1f us at 6.3 TB/s)\n", M, K, N, gf, gf / 157.3e3 * 1e6, mb, mb / 6.3e6 * 1e6);
    for (int ntw = 1; ntw <= 4; ++ntw) {
        const dim3 g((M + 127) / 128, (tiles + ntw - 1) / ntw);
        const size_t lds = (size_t)ntw * 32 * (K + 4) * 4;
        const float t0 = time_us([&] { k_pwr<KS, false><<<g, 256, lds, 0>>>(X, K, W, K, B, Y, N, M, N, Npad, ntw, 1, ua); });
        const float t1 = time_us([&] { k_ablate<KS, 1><<<g, 256, lds, 0>>>(X, K, W, K, B, Y, N, M, N, Npad, ntw, 1, 0); });
        const float t2 = time_us([&] { k_ablate<KS, 2><<<g, 256, lds, 0>>>(X, K, W, K, B, Y, N, M, N, Npad, ntw, 1, 0); });
        const float t3 = time_us([&] { k_ablate<KS, 3><<<g, 256, lds, 0>>>(X, K, W, K, B, Y, N, M, N, Npad, ntw, 1, 0); });
        printf("  ntw=%d (%4d workgroups): full %6.1f us | no stores %6.1f | no global loads %6.1f | stores only %6.1f\n",
               ntw, g.x * g.y, t0, t1, t2, t3);
    }
    hipFree(X); hipFree(W); hipFree(B); hipFree(Y);
}

int main() {
    study<8>(13312, 384);
    study<12>(13312, 576);
    study<20>(3328, 960);
    study<8>(53248, 128);
    return 0;
}
