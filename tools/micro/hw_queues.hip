// How many HIP streams really run at the same time?  K streams, one long single-workgroup spin kernel on each, launched
// together: wall time / kernel time = number of serial groups.  Run with GPU_MAX_HW_QUEUES = 4 (default), 8, 16, 32.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/hw_queues tools/micro/hw_queues.hip && /tmp/hw_queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(long long ticks, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
}
int main() {
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", e ? e : "(unset)");
    const int KMAX = 16;
    std::vector<hipStream_t> st(KMAX);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long ticks = 200000;      // 2 ms at 100 MHz
    for (int k : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < k; ++i) k_spin<<<1, 64, 0, st[i]>>>(ticks, nullptr);
            (void)hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("  %2d streams, one 2 ms kernel each: %.2f ms wall -> %.1f serial groups\n", k, ms, ms / 2.0);
        }
    }
    return 0;
}
