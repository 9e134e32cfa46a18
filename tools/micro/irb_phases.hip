// In-kernel phase timing of the fused inverted-residual block (k_irb) on the shapes of backbone blocks 2, 3 and 5
// (B = 32): s_memtime stamps of wave 0 at the phase boundaries, accumulated per phase over the chunks of a workgroup
// and averaged over workgroups.  Diagnostic build only: in the library IRB_STAMP expands to nothing.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/irb_phases.hip -o /tmp/irb_phases && /tmp/irb_phases
#include <hip/hip_runtime.h>
#include <vector>
__device__ unsigned long long *g_stamp;          // [workgroups][8]: slot 0 = previous stamp, slots 1..7 = cycles accumulated per phase
#define IRB_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (threadIdx.x == 0) {                                                                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                    \
            unsigned long long *slot_ = g_stamp + (size_t)blockIdx.x * 8;                                    \
            if ((i) != 0) slot_[(i)] += now_ - slot_[0];                                                     \
            slot_[0] = now_;                                                                                 \
        }                                                                                                    \
    } while (0)
#include "../../retargetvid_amd/csrc/svc_net.hip"

template <int S, int TOH, int TOW, int CI, int CE, int CO>
static void study(const char *name, int n, int H, int W) {
    const int OH = H / S, OW = W / S, CoutP = (CO + 31) / 32 * 32;
    const int tx = (OW + TOW - 1) / TOW, ty = (OH + TOH - 1) / TOH, wgs = n * tx * ty;
    float *X, *We, *be, *Wd, *bd, *Wp, *bp, *Y;
    hipMalloc(&X, (size_t)n * H * W * CI * 4); hipMalloc(&We, (size_t)(CE + 32) * CI * 4); hipMalloc(&be, (CE + 32) * 4);
    hipMalloc(&Wd, 9 * (CE + 32) * 4); hipMalloc(&bd, (CE + 32) * 4); hipMalloc(&Wp, (size_t)CoutP * (CE + 32) * 4); hipMalloc(&bp, CoutP * 4);
    hipMalloc(&Y, (size_t)n * OH * OW * CO * 4);
    hipMemset(X, 0, (size_t)n * H * W * CI * 4); hipMemset(We, 0, (size_t)(CE + 32) * CI * 4); hipMemset(be, 0, (CE + 32) * 4);
    hipMemset(Wd, 0, 9 * (CE + 32) * 4); hipMemset(bd, 0, (CE + 32) * 4); hipMemset(Wp, 0, (size_t)CoutP * (CE + 32) * 4); hipMemset(bp, 0, CoutP * 4);
    unsigned long long *st;
    hipMalloc(&st, (size_t)wgs * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &st, sizeof st);
    auto kfn = k_irb<S, TOH, TOW, true, false, CI, CE, CO>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    const size_t lds = IrbGeom<S, TOH, TOW>::lds_floats(CI, CoutP, true, CE) * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) kfn<<<wgs, 256, lds, 0>>>(X, H, W, CI, We, be, CE, Wd, bd, Wp, bp, CO, CoutP, nullptr, Y, CO, OH, OW, tx, ty, nullptr, nullptr);
    hipDeviceSynchronize();
    hipMemset(st, 0, (size_t)wgs * 8 * 8);
    hipEventRecord(a, 0);
    kfn<<<wgs, 256, lds, 0>>>(X, H, W, CI, We, be, CE, Wd, bd, Wp, bp, CO, CoutP, nullptr, Y, CO, OH, OW, tx, ty, nullptr, nullptr);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h((size_t)wgs * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0};
    for (int w = 0; w < wgs; ++w) for (int i = 1; i < 8; ++i) acc[i] += (double)h[(size_t)w * 8 + i];
    for (int i = 1; i < 8; ++i) acc[i] /= wgs;
    const double tot = acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    // s_memtime ticks at 100 MHz on gfx950? (guide: tick = shader cycle) -- report in ticks and as shares
    printf("%s: %d workgroups, %.1f us (stamped build), LDS %zu B; per workgroup (wave 0), ticks: loop-top %.0f | slice requests + expand %.0f | barrier 1 wait %.0f | "
           "slice stores + depthwise %.0f | barrier 2 wait %.0f | project %.0f | tail %.0f  (sum %.0f)\n", name, wgs, ms * 1e3, lds,
           acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], tot);
    hipFree(X); hipFree(We); hipFree(be); hipFree(Wd); hipFree(bd); hipFree(Wp); hipFree(bp); hipFree(Y); hipFree(st);
}

int main() {
    study<2, 4, 8, 16, 96, 24>("block 2  s2 16->96->24  @128x208", 32, 128, 208);
    study<1, 8, 8, 24, 144, 24>("block 3  s1 24->144->24 @64x104", 32, 64, 104);
    study<1, 8, 8, 32, 192, 32>("block 5  s1 32->192->32 @32x52", 32, 32, 52);
    return 0;
}
