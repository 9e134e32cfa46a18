// Micro-benchmark: where does one Prim step of k_prim (svc_tail.hip: prim_regs32) spend its ~1100 cycles?
// Same loop with pieces switched off by template flags (results are wrong when a piece is off; only time matters).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/prim_step tools/micro/prim_step.hip && /tmp/prim_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define TB 1024
#define NW16 16
#define REACH_INF 0x1FFFFu
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
    return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_u32<0x111, 0xF>(v); v = dpp_min_u32<0x112, 0xF>(v); v = dpp_min_u32<0x114, 0xF>(v);
    v = dpp_min_u32<0x118, 0xF>(v); v = dpp_min_u32<0x142, 0xA>(v); v = dpp_min_u32<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
struct E { uint16_t a, b; uint32_t w; };
// flags: 1 = global store of the edge, 2 = second level through LDS + barrier, 4 = wave-level DPP minimum
template <int PT, int F>
__global__ __launch_bounds__(TB) void k(const uint32_t *core_g, const uint16_t *rc_g, E *mst, int N, long long *cyc) {
    __shared__ uint4 slots[2 * NW16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (N + 64 * PT - 1) / (64 * PT);
    for (int i = tid; i < 2 * NW16; i += TB) slots[i] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    __syncthreads();
    if (wave >= nw) return;
    uint32_t reach[PT], corev[PT], rcv[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int p = tid * PT + i;
        reach[i] = REACH_INF; corev[i] = REACH_INF; rcv[i] = 0;
        if (p < N) { corev[i] = core_g[p]; rcv[i] = rc_g[p]; }
    }
    if (tid == 0) corev[0] = REACH_INF;
    uint32_t cur = 0;
    int cr = rc_g[0] & 255, cc = rc_g[0] >> 8;
    uint32_t ccore = core_g[0];
    const long long c0 = clock64();
    for (int step = 0; step < N - 1; ++step) {
        uint32_t best = 0xFFFFFFFFu, bcore = 0, brc = 0;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int dr = (int)(rcv[i] & 255) - cr, dc = (int)(rcv[i] >> 8) - cc;
            const uint32_t m = max(max((uint32_t)(dr * dr + dc * dc), corev[i]), ccore);
            reach[i] = min(reach[i], m);
            const uint32_t key = (reach[i] << 15) | (uint32_t)(tid * PT + i);
            if (key < best) { best = key; bcore = corev[i]; brc = rcv[i]; }
        }
        uint32_t wmin = best;
        if (F & 4) wmin = wave_min_u32(best); else wmin = (uint32_t)__builtin_amdgcn_readlane((int)best, step & 63);
        uint32_t kmin = wmin; uint32_t pc = bcore, prc = brc;
        if (F & 2) {
            uint4 *sl = slots + (step & 1) * NW16;
            if (best == wmin) sl[wave] = make_uint4(wmin, bcore, brc, 0);
            __syncthreads();
            const uint4 t = sl[lane & 15];
            uint32_t k2 = t.x;
            k2 = dpp_min_u32<0x111, 0xF>(k2); k2 = dpp_min_u32<0x112, 0xF>(k2);
            k2 = dpp_min_u32<0x114, 0xF>(k2); k2 = dpp_min_u32<0x118, 0xF>(k2);
            kmin = (uint32_t)__builtin_amdgcn_readlane((int)k2, 15);
            const int src = __ffsll((unsigned long long)__ballot(t.x == kmin)) - 1;
            pc = (uint32_t)__builtin_amdgcn_readlane((int)t.y, src);
            prc = (uint32_t)__builtin_amdgcn_readlane((int)t.z, src);
        } else {
            const int src = __ffsll((unsigned long long)__ballot(best == wmin)) - 1;
            pc = (uint32_t)__builtin_amdgcn_readlane((int)bcore, src);
            prc = (uint32_t)__builtin_amdgcn_readlane((int)brc, src);
        }
        const uint32_t nidx = kmin & 0x7FFFu;
        if ((F & 1) && tid == 0) mst[step] = E{(uint16_t)cur, (uint16_t)nidx, kmin >> 15};
        if ((int)(nidx / PT) == tid) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
                if ((int)(nidx % PT) == i) { corev[i] = REACH_INF; reach[i] = REACH_INF; }
        }
        ccore = pc; cr = prc & 255; cc = prc >> 8; cur = nidx;
    }
    if (tid == 0) { cyc[0] = clock64() - c0; if (!(F & 1)) mst[0] = E{(uint16_t)cur, 0, ccore}; }
}
// Measured (MI355X, N = 1641, clocks per step): full 1303 = update + search of all points 273 + wave DPP minimum 200 + LDS /
// barrier / second-level minimum 830; PT = 4 1163, PT = 8 1278; one wavefront with all points in registers (no LDS level)
// 2629 at 26 points per lane.  In the library on the bench's maps (N = 550..800): PT = 2 / 4 / 8 / 16 -> tail round
// 0.70 / 0.72 / 0.80 / 0.97 ms, and a one-wavefront Prim for N <= 1280 (tried, not kept) 0.78: the chain of latencies of a
// step, not the VALU work, is the bound, and V2 below (9 instead of 13 operations per point) changes nothing (1378).
// V2: fewer VALU operations per point (the update + search of all points is ~60 % of a step on one CU): coordinates kept
// unpacked, 24-bit multiplies, max3 / min / lshl_or, the best key by v_min only -- the winner's payload is looked up
// afterwards by the one lane that holds it
template <int PT, int F>
__global__ __launch_bounds__(TB) void k2(const uint32_t *core_g, const uint16_t *rc_g, E *mst, int N, long long *cyc) {
    __shared__ uint4 slots[2 * NW16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (N + 64 * PT - 1) / (64 * PT);
    for (int i = tid; i < 2 * NW16; i += TB) slots[i] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    __syncthreads();
    if (wave >= nw) return;
    uint32_t reach[PT], corev[PT];
    int pr[PT], pc_[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int p = tid * PT + i;
        reach[i] = REACH_INF; corev[i] = REACH_INF; pr[i] = 0; pc_[i] = 0;
        if (p < N) { corev[i] = core_g[p]; pr[i] = rc_g[p] & 255; pc_[i] = rc_g[p] >> 8; }
    }
    if (tid == 0) corev[0] = REACH_INF;
    uint32_t cur = 0;
    int cr = rc_g[0] & 255, cc = rc_g[0] >> 8;
    uint32_t ccore = core_g[0];
    const uint32_t idx0 = (uint32_t)(tid * PT);
    const long long c0 = clock64();
    for (int step = 0; step < N - 1; ++step) {
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int dr = pr[i] - cr, dc = pc_[i] - cc;
            const uint32_t d2 = (uint32_t)(__mul24(dc, dc) + __mul24(dr, dr));
            const uint32_t m = max(max(d2, corev[i]), ccore);
            reach[i] = min(reach[i], m);
            best = min(best, (reach[i] << 15) | (idx0 + i));
        }
        const uint32_t wmin = wave_min_u32(best);
        uint4 *sl = slots + (step & 1) * NW16;
        if (best == wmin) {
            uint32_t bcore = corev[0], brc = (uint32_t)pr[0] | ((uint32_t)pc_[0] << 8);
#pragma unroll
            for (int i = 1; i < PT; ++i)
                if ((wmin & (PT - 1)) == (uint32_t)i) { bcore = corev[i]; brc = (uint32_t)pr[i] | ((uint32_t)pc_[i] << 8); }
            sl[wave] = make_uint4(wmin, bcore, brc, 0);
        }
        __syncthreads();
        const uint4 t = sl[lane & 15];
        uint32_t k2v = t.x;
        k2v = dpp_min_u32<0x111, 0xF>(k2v); k2v = dpp_min_u32<0x112, 0xF>(k2v);
        k2v = dpp_min_u32<0x114, 0xF>(k2v); k2v = dpp_min_u32<0x118, 0xF>(k2v);
        const uint32_t kmin = (uint32_t)__builtin_amdgcn_readlane((int)k2v, 15);
        const int src = __ffsll((unsigned long long)__ballot(t.x == kmin)) - 1;
        const uint32_t pcv = (uint32_t)__builtin_amdgcn_readlane((int)t.y, src);
        const uint32_t prc = (uint32_t)__builtin_amdgcn_readlane((int)t.z, src);
        const uint32_t nidx = kmin & 0x7FFFu;
        if ((F & 1) && tid == 0) mst[step] = E{(uint16_t)cur, (uint16_t)nidx, kmin >> 15};
        if ((nidx / PT) == (uint32_t)tid) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
                if ((nidx % PT) == (uint32_t)i) { corev[i] = REACH_INF; reach[i] = REACH_INF; }
        }
        ccore = pcv; cr = prc & 255; cc = prc >> 8; cur = nidx;
    }
    if (tid == 0) { cyc[0] = clock64() - c0; if (!(F & 1)) mst[0] = E{(uint16_t)cur, 0, ccore}; }
}
template <int PT, int F>
void run2(const char *name, uint32_t *core, uint16_t *rc, E *mst, int N, long long *cyc, E *ref) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k2<PT, F><<<1, TB>>>(core, rc, mst, N, cyc);
    hipEventRecord(a);
    k2<PT, F><<<1, TB>>>(core, rc, mst, N, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    std::vector<E> got(N - 1), want(N - 1);
    hipMemcpy(got.data(), mst, (N - 1) * 8, hipMemcpyDeviceToHost); hipMemcpy(want.data(), ref, (N - 1) * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N - 1; ++i) bad += got[i].a != want[i].a || got[i].b != want[i].b || got[i].w != want[i].w;
    printf("%-44s PT=%2d  %8.1f us  %6.1f ns/step  %6.0f clk/step  edges differing from V1: %d\n", name, PT, ms * 1e3, ms * 1e6 / (N - 1), (double)c / (N - 1), bad);
}
template <int PT, int F>
void run(const char *name, uint32_t *core, uint16_t *rc, E *mst, int N, long long *cyc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PT, F><<<1, TB>>>(core, rc, mst, N, cyc);
    hipEventRecord(a);
    k<PT, F><<<1, TB>>>(core, rc, mst, N, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s PT=%2d  %8.1f us  %6.1f ns/step  %6.0f clk/step\n", name, PT, ms * 1e3, ms * 1e6 / (N - 1), (double)c / (N - 1));
}
int main() {
    const int N = 1641;
    std::vector<uint32_t> core(N); std::vector<uint16_t> rc(N);
    int n = 0;
    for (int r = 0; r < 140 && n < N; ++r) for (int c = 0; c < 250 && n < N; ++c)
        if ((r - 70) * (r - 70) * 25 + (c - 120) * (c - 120) * 16 < 25 * 16 * 26 * 26 / 4 * 4) { rc[n] = (uint16_t)(r | (c << 8)); core[n] = 13 + (n % 5); ++n; }
    for (; n < N; ++n) { rc[n] = (uint16_t)((n % 140) | ((n % 250) << 8)); core[n] = 20; }
    uint32_t *dcore; uint16_t *drc; E *dmst; long long *dcyc;
    hipMalloc(&dcore, N * 4); hipMalloc(&drc, N * 2); hipMalloc(&dmst, N * 8); hipMalloc(&dcyc, 8);
    hipMemcpy(dcore, core.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(drc, rc.data(), N * 2, hipMemcpyHostToDevice);
    run<2, 7>("full", dcore, drc, dmst, N, dcyc);
    { E *dref; hipMalloc(&dref, N * 8); hipMemcpy(dref, dmst, (N - 1) * 8, hipMemcpyDeviceToDevice);
      run2<2, 7>("V2 full", dcore, drc, dmst, N, dcyc, dref); run2<4, 7>("V2 full", dcore, drc, dmst, N, dcyc, dref);
      run2<3, 7>("V2 full", dcore, drc, dmst, N, dcyc, dref); }
    run<2, 6>("no global edge store", dcore, drc, dmst, N, dcyc);
    run<2, 5>("no LDS/barrier second level", dcore, drc, dmst, N, dcyc);
    run<2, 3>("no wave DPP minimum", dcore, drc, dmst, N, dcyc);
    run<2, 0>("update only", dcore, drc, dmst, N, dcyc);
    run<4, 7>("full", dcore, drc, dmst, N, dcyc);
    run<8, 7>("full", dcore, drc, dmst, N, dcyc);
    run<8, 6>("no global edge store", dcore, drc, dmst, N, dcyc);
    run<8, 5>("no LDS/barrier second level", dcore, drc, dmst, N, dcyc);
    run<32, 5>("one wave, no second level", dcore, drc, dmst, N, dcyc);
    run<32, 4>("one wave, no second level, no store", dcore, drc, dmst, N, dcyc);
    return 0;
}
