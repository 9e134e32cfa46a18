// Does an output element of v_mfma_f32_32x32x16_bf16 depend on anything but its own row of A and column of B?
// (round 5: a frame's saliency map must not depend on its neighbours in a tile.)  One tile D = A.B with random bf16 operands;
// then every OTHER column of B (resp. row of A) is multiplied by 2^20 / set to zero / replaced, and column 0 (row 0) of D is
// compared bit for bit.  Also: the same with the accumulator C of the other elements made huge.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/bf16_mfma_coupling.hip -o /tmp/cpl && /tmp/cpl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A[32][16], B[16][32] as float (bf16-representable), C[32][32]; D out.  Lane l: A[r][8h + j], B[8h + j][r]; D col = l & 31, row = (i&3) + 8(i>>2) + 4h
__global__ void k_tile(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    s8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)(__float_as_uint(A[r * 16 + 8 * h + j]) >> 16);
        b[j] = (short)(__float_as_uint(B[(8 * h + j) * 32 + r]) >> 16);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static float bf(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; memcpy(&v, &u, 4); return v; }
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main() {
    float hA[512], hB[512], hC[1024], hD0[1024], hD[1024];
    float *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096)); CK(hipMalloc(&dD, 4096));
    auto run = [&](float *out) {
        CK(hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice));
        k_tile<<<1, 64>>>(dA, dB, dC, dD);
        CK(hipMemcpy(out, dD, 4096, hipMemcpyDeviceToHost));
    };
    int bad_col = 0, bad_row = 0, bad_c = 0, trials = 2000;
    srand(3);
    for (int t = 0; t < trials; ++t) {
        // wide dynamic range inside the dot product, like activations times split planes
        for (int i = 0; i < 512; ++i) { hA[i] = bf(rnd() * powf(2.f, (float)(rand() % 24 - 12))); hB[i] = bf(rnd() * powf(2.f, (float)(rand() % 24 - 12))); }
        for (int i = 0; i < 1024; ++i) hC[i] = rnd() * powf(2.f, (float)(rand() % 20 - 10));
        run(hD0);
        float sB[512]; memcpy(sB, hB, 2048);
        const int mode = t % 3;
        for (int k = 0; k < 16; ++k) for (int c = 1; c < 32; ++c)            // every other COLUMN of B (the other pixels of the tile)
            hB[k * 32 + c] = mode == 0 ? bf(sB[k * 32 + c] * 1048576.f) : mode == 1 ? 0.f : bf(rnd() * 1e6f);
        run(hD);
        for (int row = 0; row < 32; ++row) if (memcmp(&hD[row * 32], &hD0[row * 32], 4)) { ++bad_col; break; }
        memcpy(hB, sB, 2048);
        float sA[512]; memcpy(sA, hA, 2048);
        for (int rr = 1; rr < 32; ++rr) for (int k = 0; k < 16; ++k) hA[rr * 16 + k] = mode == 0 ? bf(sA[rr * 16 + k] * 1048576.f) : mode == 1 ? 0.f : bf(rnd() * 1e6f);
        run(hD);
        if (memcmp(hD, hD0, 32 * 4)) ++bad_row;                                // row 0 of D
        memcpy(hA, sA, 2048);
        float sC[1024]; memcpy(sC, hC, 4096);
        for (int i = 1; i < 1024; ++i) hC[i] = sC[i] * 1e9f;                   // every other accumulator element
        run(hD);
        if (memcmp(hD, hD0, 4)) ++bad_c;
        memcpy(hC, sC, 4096);
    }
    printf("%d trials: column 0 of D changed when the OTHER columns of B changed: %d;  row 0 changed when the other rows of A changed: %d;  D[0][0] changed when the other C changed: %d\n",
           trials, bad_col, bad_row, bad_c);
    return 0;
}
