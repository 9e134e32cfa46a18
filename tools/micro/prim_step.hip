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
