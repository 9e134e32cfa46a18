// Which instruction gives the smoothing kernel its wrong pixels beside bf16-pipe workgroups?  (round 5)
// k_smooth_down's bilinear stage compiles to   v_pk_mul_f32 x2 ; v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]   (the add
// takes the HIGH half of its second source for the LOW result), and the self-check build of the library showed the value
// lx1 * t1[x1] -- that high half -- missing from the sum in lanes 48..63 of one wavefront, with the operands in the
// registers correct.  This kernel issues the sequences in isolation (inline asm, inputs a pure function of block / thread /
// iteration), checks each against the same arithmetic done with scalar VALU instructions in the same thread, and counts the
// disagreements per test and per quarter of the wavefront.  Run beside the library's network passes
// (tools/soak_network_concurrent.py, VICTIM=k VICTIM_LIB=tools/micro/libpkvictim.so).
//   test 0: v_pk_mul, v_pk_mul, v_pk_add op_sel:[0,1] op_sel_hi:[1,0]   (the smoothing kernel's sequence)
//   test 1: the products by scalar v_mul_f32, then v_pk_add op_sel:[0,1] op_sel_hi:[1,0] alone
//   test 2: v_pk_mul, v_pk_mul, then the additions by scalar v_add_f32 (the packed products read half by half)
//   test 3: v_pk_mul, v_pk_mul, v_pk_add WITHOUT op_sel (operands arranged so that no half is swapped)
//   tests 4-7 (second kernel, k_pkvictim_lds): test 0's sequence with the four a-values arriving from an LDS tile through the
//   smoothing kernel's own addressing (bilinear read-out of a 14 x 416 tile), 256- and 1024-thread workgroups, 44 KB of LDS
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/pk_opsel_victim.hip -o tools/micro/libpkvictim.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); }
__global__ __launch_bounds__(256) void k_pkvictim(unsigned long long *cnt, int iters) {
    const int tid = threadIdx.x, q = (tid & 63) >> 4;
    uint32_t s = (blockIdx.x * 256 + tid) * 2654435761u + 12345u;
    unsigned bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const float lx1 = rnd(s), lx0 = 1.f - lx1, a00 = rnd(s), a01 = rnd(s), a10 = rnd(s), a11 = rnd(s);
        // scalar reference: row0 = lx0 a00 + lx1 a01, row1 = lx0 a10 + lx1 a11
        float p00, p01, p10, p11, r0, r1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p00) : "v"(lx0), "v"(a00));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p01) : "v"(lx1), "v"(a01));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p10) : "v"(lx0), "v"(a10));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p11) : "v"(lx1), "v"(a11));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(p00), "v"(p01));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p10), "v"(p11));
        const f2 L = {lx0, lx1};
        {   // test 0: m1 = (lx0 a10, lx1 a01), m2 = (lx0 a00, lx1 a11); d = (m1.lo + m2.hi, m1.hi + m2.lo) = (row1, row0)
            f2 m1 = {a10, a01}, m2 = {a00, a11}, d;
            asm volatile("v_pk_mul_f32 %0, %2, %0\n\tv_pk_mul_f32 %1, %2, %1" : "+v"(m1), "+v"(m2) : "v"(L));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(m1), "v"(m2));
            bad[0] += (d.x != r1) + (d.y != r0);
        }
        {   // test 1
            const f2 m1 = {p10, p01}, m2 = {p00, p11};
            f2 d;
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(m1), "v"(m2));
            bad[1] += (d.x != r1) + (d.y != r0);
        }
        {   // test 2
            f2 m1 = {a10, a01}, m2 = {a00, a11};
            asm volatile("v_pk_mul_f32 %0, %2, %0\n\tv_pk_mul_f32 %1, %2, %1" : "+v"(m1), "+v"(m2) : "v"(L));
            float e0, e1;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(m1.x), "v"(m2.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(m1.y), "v"(m2.x));
            bad[2] += (e1 != r1) + (e0 != r0);
        }
        {   // test 3: m1 = (lx0 a00, lx0 a10) ... no: keep L = (lx0, lx1) and pair the columns: m1 = (lx0 a00, lx1 a11), m2 = (lx1' ...)
            // products arranged as m1 = (lx0 a00, lx1 a11), m2 = (lx1 a01, lx0 a10) needs a swapped L: use two L registers
            const f2 Ls = {lx1, lx0};
            f2 m1 = {a00, a11}, m2 = {a01, a10}, d;
            asm volatile("v_pk_mul_f32 %0, %2, %0\n\tv_pk_mul_f32 %1, %3, %1" : "+v"(m1), "+v"(m2) : "v"(L), "v"(Ls));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(m1), "v"(m2));
            bad[3] += (d.x != r0) + (d.y != r1);
        }
    }
    for (int t = 0; t < 4; ++t)
        if (bad[t]) atomicAdd(cnt + t * 4 + q, (unsigned long long)bad[t]);
}
// test 4 (256 threads) / 5 (1024 threads): the operands come from LDS as in k_smooth_down's bilinear stage
#define TNW 416
#define TROWS 14
template <int NTH>
__global__ __launch_bounds__(NTH) void k_pkvictim_lds(unsigned long long *cnt, int iters, int slot) {
    extern __shared__ float tile[];
    const int tid = threadIdx.x, q = (tid & 63) >> 4, b = blockIdx.x;
    for (int i = tid; i < TROWS * TNW; i += NTH) {
        const uint32_t h = (uint32_t)(b * 7919 + i) * 2654435761u;
        tile[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + (float)(i % TNW) * 0.001f;
    }
    __syncthreads();
    const float scy = 256.0f / 140.0f, scx = 416.0f / 250.0f;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it)
        for (int idx = tid; idx < 7 * 250; idx += NTH) {
            const int oyr = idx / 250, ox = idx - oyr * 250;
            const float sy = fmaxf(scy * (oyr + 0.5f) - 0.5f, 0.f), sx = fmaxf(scx * (ox + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < TROWS - 1 ? 1 : 0), x1 = x0 + (x0 < TNW - 1 ? 1 : 0);
            const float lx1 = sx - x0, lx0 = 1.f - lx1;
            const float *t0 = tile + y0 * TNW, *t1 = tile + y1 * TNW;
            float a00 = t0[x0], a01 = t0[x1], a10 = t1[x0], a11 = t1[x1];
            f2 m1 = {a10, a01}, m2 = {a00, a11}, d;
            const f2 L = {lx0, lx1};
            asm volatile("v_pk_mul_f32 %0, %2, %0\n\tv_pk_mul_f32 %1, %2, %1" : "+v"(m1), "+v"(m2) : "v"(L));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(m1), "v"(m2));
            float p00, p01, p10, p11, r0, r1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p00) : "v"(lx0), "v"(a00));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p01) : "v"(lx1), "v"(a01));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p10) : "v"(lx0), "v"(a10));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p11) : "v"(lx1), "v"(a11));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(p00), "v"(p01));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p10), "v"(p11));
            bad += (d.x != r1) + (d.y != r0);
        }
    if (bad) atomicAdd(cnt + slot * 4 + q, (unsigned long long)bad);
}

// ---- aggressors for the stand-alone run (tools/micro/pk_opsel_run.py): instruction mixes of k_pwr's bf16 form, one kind per launch ------
//   0: v_pk_mov_b32 ... op_sel:[1,0]   1: v_perm_b32   2: v_mfma_f32_32x32x16_bf16 fed by v_perm_b32   3: ds_read_b128 + the MFMA
typedef short ag_b8 __attribute__((ext_vector_type(8)));
typedef float ag_f16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_aggressor(float *sink, int iters, int kind) {
    __shared__ uint4 lds[1024];
    const int tid = threadIdx.x;
    lds[tid] = make_uint4(tid, tid * 3, tid * 5, tid * 7); lds[tid + 256] = lds[tid]; lds[tid + 512] = lds[tid]; lds[tid + 768] = lds[tid];
    __syncthreads();
    f2 a = {(float)tid, 1.5f}, b = {2.5f, (float)(tid & 7)}, d = {0.f, 0.f};
    unsigned u = tid * 2654435761u, w = 0x9e3779b9u;
    ag_f16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (kind == 0) {
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n\tv_pk_mov_b32 %1, %0, %2 op_sel:[1,0]" : "+v"(d), "+v"(a) : "v"(b));
        } else if (kind == 1) {
            asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_perm_b32 %1, %1, %0, %2" : "+v"(u), "+v"(w) : "s"(0x07060302u));
        } else {
            union { ag_b8 v; unsigned q[4]; uint4 x; } A, B;
            if (kind == 3) { A.x = lds[(tid + it * 7) & 1023]; B.x = lds[(tid * 3 + it) & 1023]; }
            else { A.q[0] = u; A.q[1] = w; A.q[2] = u ^ w; A.q[3] = u + w; B = A; }
            asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(A.q[0]) : "v"(w), "s"(0x07060302u));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, acc, 0, 0, 0);
            u = u * 1664525u + 1013904223u;
        }
    }
    if (sink) sink[blockIdx.x * 256 + tid] = d.x + a.y + (float)u + (float)w + acc[0] + acc[7];
}
extern "C" int aggressor_launch(void *stream, int kind, int blocks, int iters) {
    k_aggressor<<<blocks, 256, 0, (hipStream_t)stream>>>(nullptr, iters, kind);
    return (int)hipGetLastError();
}
static unsigned long long *g_cnt = nullptr;
static unsigned long long g_launches = 0;
extern "C" int victim_init() {
    if (hipMalloc(&g_cnt, 32 * 8)) return -1;
    hipMemset(g_cnt, 0, 32 * 8);
    return (int)hipDeviceSynchronize();
}
extern "C" int victim_launch(void *stream) {
    ++g_launches;
    k_pkvictim<<<1024, 256, 0, (hipStream_t)stream>>>(g_cnt, 400);
    k_pkvictim_lds<256><<<640, 256, TROWS * TNW * 4, (hipStream_t)stream>>>(g_cnt, 8, 4);
    k_pkvictim_lds<1024><<<320, 1024, TROWS * TNW * 4, (hipStream_t)stream>>>(g_cnt, 16, 5);
    return (int)hipGetLastError();
}
extern "C" unsigned long long victim_diffs() {
    unsigned long long h[32], t = 0;
    hipDeviceSynchronize();
    hipMemcpy(h, g_cnt, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) t += h[i];
    return t;
}
extern "C" int victim_report(unsigned long long *out32) {
    hipDeviceSynchronize();
    return (int)hipMemcpy(out32, g_cnt, 32 * 8, hipMemcpyDeviceToHost);
}
