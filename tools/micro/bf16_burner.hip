// A workgroup that does nothing but issue dense bf16 (or f16, or f32) MFMAs on register operands: no LDS, no global
// loads, one dummy store that never happens.  tools/soak_network_concurrent.py launches it on extra streams beside the
// fp32-pipe network to ask: does ANOTHER kernel's bf16 matrix load alone change the network's results?  (round 5)
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/bf16_burner.hip -o tools/micro/libbf16_burner.so
#include <hip/hip_runtime.h>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k_burn(float *out, int iters, float seed) {
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((threadIdx.x * 7 + i * 13) & 0x7f)); b[i] = (short)(0x3f00 + ((threadIdx.x * 5 + i * 11) & 0x7f)); }
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = seed; acc1[i] = -seed; }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
            }
        } else if (KIND == 1) {
            union { s8 s; h8 h; } ua, ub; ua.s = a; ub.s = b;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ub.h, ua.h, acc1, 0, 0, 0);
            }
        } else {
            const float fa = __uint_as_float((uint32_t)(unsigned short)a[0] << 16), fb = __uint_as_float((uint32_t)(unsigned short)b[0] << 16);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) { acc0[i] *= 0.5f; acc1[i] *= 0.5f; }
    }
    if (acc0[0] + acc1[3] == 12345.678f) out[threadIdx.x] = acc0[1];
}
extern "C" int burn_launch(void *stream, int blocks, int iters, int kind) {
    static float *out = nullptr;
    if (!out && hipMalloc(&out, 4096) != hipSuccess) return -1;
    if (kind == 0) k_burn<0><<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters, 1.0f);
    else if (kind == 1) k_burn<1><<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters, 1.0f);
    else k_burn<2><<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters, 1.0f);
    return (int)hipGetLastError();
}
