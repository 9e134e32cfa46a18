// svc_x3.h -- split-bf16 operands for the bf16 matrix pipe (device code shared by svc_net.hip and svc_shot.hip; the scheme and
// its operand convention are described in svc_net.hip, "Split-bf16 operands").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef X3_NP
#define X3_NP 6
#endif
typedef short bf16x8 __attribute__((ext_vector_type(8)));
union X3Q { bf16x8 v; uint32_t u[4]; uint4 q; };
struct X3 { bf16x8 h, m, l; };
__device__ __forceinline__ uint32_t x3_pack(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (a >> 16) | (b & 0xffff0000)
template <bool RNE>
__device__ __forceinline__ void x3_split1(float x, uint32_t &h, uint32_t &m, uint32_t &l) {      // planes in the upper 16 bits
    if (RNE) {
        auto rne = [](float v) { const uint32_t u = __float_as_uint(v); return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; };
        h = rne(x);
        const float r1 = x - __uint_as_float(h);
        m = rne(r1);
        l = rne(r1 - __uint_as_float(m));
    } else {
        h = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(h);
        m = __float_as_uint(r1) & 0xffff0000u;
        l = __float_as_uint(r1 - __uint_as_float(m));
    }
}
template <bool RNE = false>
__device__ __forceinline__ X3 x3_split(const float4 a0, const float4 a1) {
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x3_split1<RNE>(a[i], h[i], m[i], l[i]);
    X3Q H, Mi, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        H.u[i] = x3_pack(h[2 * i], h[2 * i + 1]);
        Mi.u[i] = x3_pack(m[2 * i], m[2 * i + 1]);
        L.u[i] = x3_pack(l[2 * i], l[2 * i + 1]);
    }
    return X3{H.v, Mi.v, L.v};
}
__device__ __forceinline__ X3 x3_load(const uint4 *p) {      // three consecutive uint4: the planes of one (lane, step)
    X3Q H, Mi, L;
    H.q = p[0]; Mi.q = p[1]; L.q = p[2];
    return X3{H.v, Mi.v, L.v};
}
// acc += w . a over the step's 16 k (weights as the A operand: a lane ends up with one pixel, as in the fp32 kernels)
__device__ __forceinline__ void x3_mma(f32x16 &acc, const X3 &w, const X3 &a) {
#if X3_NP >= 9
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, a.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, a.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, a.l, acc, 0, 0, 0);
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, a.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, a.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, a.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, a.h, acc, 0, 0, 0);
}
