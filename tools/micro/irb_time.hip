// Device time of the fused inverted-residual block (k_irb) on the shapes of backbone blocks 2 .. 7 (B = 32), no stamps:
// A/B of kernel variants (-DIRB_XREG=0: pixel operand staged in LDS as in rounds 1-3).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DIRB_XREG=0] tools/micro/irb_time.hip -o /tmp/irb_time && /tmp/irb_time
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include "../../retargetvid_amd/csrc/svc_net.hip"

template <int S, int TOH, int TOW, int CI, int CE, int CO>
static double study(const char *name, int n, int H, int W) {
    const int OH = H / S, OW = W / S, CoutP = (CO + 31) / 32 * 32;
    const int tx = (OW + TOW - 1) / TOW, ty = (OH + TOH - 1) / TOH, wgs = n * tx * ty;
    float *X, *We, *be, *Wd, *bd, *Wp, *bp, *Y;
    hipMalloc(&X, (size_t)n * H * W * CI * 4); hipMalloc(&We, (size_t)(CE + 32) * CI * 4); hipMalloc(&be, (CE + 32) * 4);
    hipMalloc(&Wd, 9 * (CE + 32) * 4); hipMalloc(&bd, (CE + 32) * 4); hipMalloc(&Wp, (size_t)CoutP * (CE + 32) * 4); hipMalloc(&bp, CoutP * 4);
    hipMalloc(&Y, (size_t)n * OH * OW * CO * 4);
    std::vector<float> hx((size_t)n * H * W * CI);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)(CE + 32) * 64 * 9, 0.01f);
    hipMemcpy(We, hw.data(), (size_t)(CE + 32) * CI * 4, hipMemcpyHostToDevice); hipMemset(be, 0, (CE + 32) * 4);
    hipMemcpy(Wd, hw.data(), 9 * (CE + 32) * 4, hipMemcpyHostToDevice); hipMemset(bd, 0, (CE + 32) * 4);
    hipMemcpy(Wp, hw.data(), (size_t)CoutP * (CE + 32) * 4, hipMemcpyHostToDevice); hipMemset(bp, 0, CoutP * 4);
    auto kfn = k_irb<S, TOH, TOW, true, false, CI, CE, CO>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    const bool xreg = IRB_XREG && CI <= 32 && CI % 8 == 0;
    const size_t lds = IrbGeom<S, TOH, TOW>::lds_floats(CI, CoutP, true, CE, xreg) * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 5; ++it) kfn<<<wgs, 256, lds, 0>>>(X, H, W, CI, We, be, CE, Wd, bd, Wp, bp, CO, CoutP, nullptr, Y, CO, OH, OW, tx, ty, nullptr, nullptr);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int it = 0; it < 9; ++it) {
        hipEventRecord(a, 0);
        kfn<<<wgs, 256, lds, 0>>>(X, H, W, CI, We, be, CE, Wd, bd, Wp, bp, CO, CoutP, nullptr, Y, CO, OH, OW, tx, ty, nullptr, nullptr);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kfn, 256, lds);
    std::vector<float> hy(64);
    hipMemcpy(hy.data(), Y, 64 * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hy) cs += v;
    printf("%-36s %5d workgroups  LDS %6zu B  %d workgroups/CU  median %7.1f us  (min %.1f)  checksum %.6f\n", name, wgs, lds, occ, t[4], t[0], cs);
    hipFree(X); hipFree(We); hipFree(be); hipFree(Wd); hipFree(bd); hipFree(Wp); hipFree(bp); hipFree(Y);
    return t[4];
}

int main() {
    printf("IRB_XREG=%d\n", (int)IRB_XREG);
    double s = 0;
    s += study<2, 4, 8, 16, 96, 24>("block 2  s2 16->96->24  @128x208", 32, 128, 208);
    s += study<1, 8, 8, 24, 144, 24>("block 3  s1 24->144->24 @64x104", 32, 64, 104);
    s += study<2, 4, 8, 24, 144, 32>("block 4  s2 24->144->32 @64x104", 32, 64, 104);
    s += 2 * study<1, 8, 8, 32, 192, 32>("block 5,6 s1 32->192->32 @32x52", 32, 32, 52);
    s += study<1, 8, 8, 32, 192, 64>("block 7  s1 32->192->64 @32x52", 32, 32, 52);
    printf("sum over blocks 2..7: %.1f us\n", s);
    return 0;
}
