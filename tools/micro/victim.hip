// A stand-in for the kernel whose results change when bf16-pipe workgroups share its CU (k_smooth_down's shape: an LDS tile
// filled by the workgroup, a barrier, then a bilinear read-out with an integer division by multiplication, 44 KB of LDS,
// 640 workgroups of 256 threads).  Everything it computes is a pure function of (block, thread): run it alone once for the
// reference, then beside other work, and count the words that differ.  (round 5)
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC tools/micro/victim.hip -o tools/micro/libvictim.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#define NW 416
#define ROWS 14
__global__ __launch_bounds__(256) void k_victim(float *__restrict__ out, int w, int rows_out, unsigned long long mul, int shift) {
    extern __shared__ float tile[];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < ROWS * NW; i += 256) {
        const uint32_t h = (uint32_t)(b * 7919 + i) * 2654435761u;
        tile[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + (float)(i % NW) * 0.001f;
    }
    __syncthreads();
    const float scy = 256.0f / 140.0f, scx = 416.0f / 250.0f;
    for (int idx = tid; idx < rows_out * w; idx += 256) {
        const uint32_t oyr = (uint32_t)(((unsigned long long)idx * mul) >> shift);          // idx / w
        const uint32_t ox = idx - oyr * w;
        const float sy = fmaxf(scy * (oyr + 0.5f) - 0.5f, 0.f), sx = fmaxf(scx * (ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, ROWS - 1), x1 = min(x0 + 1, NW - 1);
        const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float *t0 = tile + y0 * NW, *t1 = tile + y1 * NW;
        out[(size_t)b * rows_out * w + idx] = ly0 * (lx0 * t0[x0] + lx1 * t0[x1]) + ly1 * (lx0 * t1[x0] + lx1 * t1[x1]);
    }
}
__global__ void k_count_diff(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *cnt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(cnt, 1ull);
}
static float *g_ref = nullptr, *g_out = nullptr;
static unsigned long long *g_cnt = nullptr;
static const int W = 250, RO = 7, NB = 640;
extern "C" int victim_init() {
    const size_t n = (size_t)NB * RO * W;
    if (hipMalloc(&g_ref, n * 4) || hipMalloc(&g_out, n * 4) || hipMalloc(&g_cnt, 8)) return -1;
    hipMemset(g_cnt, 0, 8);
    int l = 0; while ((1u << l) < (unsigned)W) ++l;
    const int s = 32 + l; const unsigned long long m = ((1ull << s) + W - 1) / W;
    k_victim<<<NB, 256, ROWS * NW * 4, 0>>>(g_ref, W, RO, m, s);
    return (int)hipDeviceSynchronize();
}
extern "C" int victim_launch(void *stream) {
    int l = 0; while ((1u << l) < (unsigned)W) ++l;
    const int s = 32 + l; const unsigned long long m = ((1ull << s) + W - 1) / W;
    const size_t n = (size_t)NB * RO * W;
    k_victim<<<NB, 256, ROWS * NW * 4, (hipStream_t)stream>>>(g_out, W, RO, m, s);
    k_count_diff<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint32_t *)g_ref, (const uint32_t *)g_out, n, g_cnt);
    return (int)hipGetLastError();
}
extern "C" unsigned long long victim_diffs() {
    unsigned long long h = 0;
    hipDeviceSynchronize();
    hipMemcpy(&h, g_cnt, 8, hipMemcpyDeviceToHost);
    return h;
}
