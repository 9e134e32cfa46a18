// svc_tail.hip — per-pixel tail of the SmartVidCrop hot path for gfx950:
// threshold, cut blend, filtering-through-clustering (HDBSCAN*), grey CLOSE 5x5,
// centre of focus, IoU.  Integer / byte work: results are bit-identical to the CPU
// oracle (oracle/tail_ref.py, oracle/hdbscan_ref.py).
//
// One workgroup of 1024 threads (16 wavefronts of 64) owns one saliency map; maps are
// independent except for the cut blend, which is handled by running the kernels in
// "rounds" (a map blended from its predecessor runs one round later).
//
// Reference semantics (paths relative to the reference tree):
//   k_threshold   sc_threshold                      smartVidCrop.py:1050-1059
//   k_blend       cut-adjacent blend, u8 wrap        smartVidCrop.py:2369-2373
//   k_compact     coo_matrix gather, raster order    smartVidCrop.py:1089-1091
//   k_core        HDBSCAN core distances                 \
//   k_prim_lvl    Prim on mutual reachability (rounds)   |  hdbscan generic path, call site smartVidCrop.py:1099
//   k_sort        numpy's argsort order of the edges      |  (k_prim_pt / k_prim_big, k_tree: the one-node-per-step Prim and the
//   k_tree_par    hierarchy, EOM, labels                 /   serial union-find builder, for maps the parallel kernels do not take)
//   k_finish      cluster weights / first arg-max / zeroing (smartVidCrop.py:1107-1122), CLOSE 5x5 (:1126-1128),
//                 centroid (:1163-1219); k_centre_argmax: com_km = False (:1165-1178)
//   k_tail_front / k_tail_back  the stages of a round fused into two launches
//   k_iou         bb_intersection_over_union         smartVidCrop.py:927-944
#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>

#include "hdb_tree.h"
#include "svc_internal.h"

#ifndef TB
#define TB 1024                 // threads per frame workgroup
#endif
#define NW16 (TB / 64)          // wavefronts per workgroup
// What the kernels below assume of TB, stated once (round 4's TB = 512 build did not terminate: three block reductions read
// sixteen per-wavefront slots whatever NW16 was, i.e. the other parity's stale minima): the per-wavefront slots of a block
// reduction are read back as slot[lane & (NW16 - 1)] and reduced inside a row of 16 lanes, so NW16 must be a power of two of
// at most 16; k_prim_lvl's rounds run on LVL_WORKERS (8, at most NW16) wavefronts.  Every launch below uses TB, and the fused kernels
// refuse to run under any other block size (tail_block_ok).
static_assert(TB % 64 == 0 && (NW16 & (NW16 - 1)) == 0 && NW16 >= 4 && NW16 <= 16,
              "the tail kernels need 4, 8 or 16 wavefronts per workgroup");
#define RING_R 20               // ring table radius for core distances
#define RING_R1 4                // radius walked by the one-thread-per-point phase
#define DEPTH_SLOT 4096          // maps per svc_cluster_center call
#define REACH_INF 0x1FFFFu      // > any squared distance on a <=256x256 grid (17 bits)

// --------------------------------------------------------------------------------------
// per-frame workspace layout (all offsets in bytes from the frame base)
// --------------------------------------------------------------------------------------
struct FrameWS {
    uint32_t hdr;        // int32[32]: [0]=N  [1]=nsel  [2]=kept cluster  [3]=clustered flag  [16],[17]=k_prim_lvl rounds, rises
    uint32_t pts;        // u32[cap]   row | col<<8 | value<<16
    uint32_t core;       // u32[cap]
    uint32_t mst;        // Edge[cap]  Prim order
    uint32_t ea, eb;     // Edge[cap]  radix ping-pong (ea = sorted result)
    uint32_t labels;     // i32[cap]
    uint32_t reach;      // u32[cap]   (generic large-N Prim only)
    uint32_t srt;        // 16 B x cap  working arrays of k_sort when they do not fit in LDS
    uint32_t sp, ssz, absc, absw, sdn, evc, evs, dparent;
    uint32_t cup, ctp, cleft, cright, cbirthw, cminw, csize, cdn, csplit, cspa, cspb, cacc, csel, crep, cweight;
    uint32_t total;
};

static FrameWS make_layout(int cap, int mc) {
    FrameWS L;
    uint32_t o = 0;
    auto take = [&](size_t bytes) { uint32_t r = o; o += (uint32_t)((bytes + 63) / 64 * 64); return r; };
    L.hdr = take(128);
    L.pts = take(4u * cap); L.core = take(4u * cap);
    L.mst = take(8u * cap); L.ea = take(8u * cap); L.eb = take(8u * cap);
    L.labels = take(4u * cap); L.reach = take(4u * cap); L.srt = take(16u * cap + 128);
    L.sp = take(2u * cap); L.ssz = take(2u * cap); L.absc = take(2u * cap); L.absw = take(4u * cap);
    L.sdn = take(4u * cap); L.evc = take(2u * cap); L.evs = take(2u * cap);
    L.dparent = take(8u * cap);
    L.cup = take(2u * mc); L.ctp = take(4u * mc); L.cleft = take(4u * mc); L.cright = take(4u * mc);
    L.cbirthw = take(4u * mc); L.cminw = take(4u * mc); L.csize = take(4u * mc);
    L.cdn = take(4u * mc); L.csplit = take(4u * mc); L.cspa = take(4u * mc); L.cspb = take(4u * mc);
    L.cacc = take(8u * mc); L.csel = take(1u * mc);
    L.crep = take(4u * mc); L.cweight = take(4u * mc);
    L.total = o;
    return L;
}

struct TailArgs {
    uint8_t *maps;          // [n][h][w]
    uint8_t *ws;            // workspace, one slot per map the call PROCESSES (held maps have none): slot = slot0 + blockIdx.x
    size_t ws_stride;
    int slot0;              // position of the launch's first map in the call's list of processed maps (sorted by round)
    const uint16_t *order;  // the maps this launch works on (one workgroup each): the current round's slice of the list of all maps sorted by round
    int n, h, w;
    FDiv dW;                // division by w
    int mcs, min_samples, select_sum, op_close, clust_filt;
    int prim_pt;            // tuning: smallest points-per-thread variant of k_prim
    const uint32_t *ring;   // sorted neighbour offsets
    const int32_t *ring_delta;  // dr * w + dc of the same offsets
    int n_ring, n_ring1;    // all offsets within RING_R / the prefix within RING_R1
    const uint16_t *ring_cnt;   // [RING_R^2 + 1]: offsets with d2 <= index
    int prim_lvl;           // 1: k_prim_lvl for maps of up to LVL_CAP points (SVC_PRIM_LVL)
    double *xy;
    int32_t *stats;
    FrameWS L;
};

// --------------------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
// wave-wide inclusive prefix sum (64 lanes) as six DPP-fused additions: a Hillis-Steele scan inside every row of 16
// lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals are carried into the rows above
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v = dpp_add_u32<0x111, 0xF>(v);
    v = dpp_add_u32<0x112, 0xF>(v);
    v = dpp_add_u32<0x114, 0xF>(v);
    v = dpp_add_u32<0x118, 0xF>(v);
    v = dpp_add_u32<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add_u32<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

// sum over the wavefront, in every lane (six DPP additions and a read-lane: the butterfly of six ds_bpermute round trips it
// replaces was most of a bisection step in k_core's phase 3)
__device__ __forceinline__ int wave_sum_i32(int v) {
    return __builtin_amdgcn_readlane((int)wave_incl_scan_u32((uint32_t)v), 63);
}

template <typename T>
__device__ __forceinline__ T *carve(uint8_t *&p, size_t count) {
    T *r = (T *)p;
    p += (count * sizeof(T) + 15) / 16 * 16;
    return r;
}

// exclusive prefix sum of one int per thread over the 1024-thread block; total in *tot
__device__ __forceinline__ int block_excl_scan(int v, int *lds16, int *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = (int)wave_incl_scan_u32((uint32_t)v);      // (six DPP additions; was six ds_bpermute round trips)
    if (lane == 63) lds16[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NW16; ++i) {
        int c = lds16[i];
        if (i < wave) base += c;
        total += c;
    }
    __syncthreads();
    *tot = total;
    return base + inc - v;
}

// --------------------------------------------------------------------------------------
// threshold / blend / IoU
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_threshold(uint8_t *maps, size_t n, int t) {
    size_t i16 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i16 >= n) return;
    if (i16 + 16 <= n && (((uintptr_t)(maps + i16)) & 15) == 0) {
        uint4 v = *(uint4 *)(maps + i16);
        uint32_t *p = (uint32_t *)&v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x = p[j], r = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint32_t c = (x >> (8 * b)) & 255u;
                r |= ((int)c < t ? 0u : c) << (8 * b);
            }
            p[j] = r;
        }
        *(uint4 *)(maps + i16) = v;
    } else {
        for (size_t i = i16; i < n && i < i16 + 16; ++i) maps[i] = maps[i] < t ? 0 : maps[i];
    }
}

__global__ __launch_bounds__(256) void k_blend(uint8_t *maps, const uint16_t *order, int hw) {
    const int f = order[blockIdx.y];                        // the maps of this round (see svc_cluster_center)
    uint8_t *cur = maps + (size_t)f * hw;
    const uint8_t *prev = cur - hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256)
        cur[i] = (uint8_t)(((cur[i] + prev[i]) & 255) >> 1);
}

__global__ __launch_bounds__(256) void k_iou(const int4 *__restrict__ a, const int4 *__restrict__ b, size_t n,
                                             double *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 A = a[i], B = b[i];
    long long xa = max(A.x, B.x), ya = max(A.y, B.y), xb = min(A.z, B.z), yb = min(A.w, B.w);
    long long inter = max(0LL, xb - xa + 1) * max(0LL, yb - ya + 1);
    long long aa = (long long)(A.z - A.x + 1) * (A.w - A.y + 1), ab = (long long)(B.z - B.x + 1) * (B.w - B.y + 1);
    out[i] = (double)inter / (double)(aa + ab - inter);
}

// --------------------------------------------------------------------------------------
// resize_factor != 1 (best settings): OpenCV INTER_LINEAR on single-channel u8 maps for the maps of
// the current round, and the centre of the INTER_NEAREST-shrunk map.
// tab (int32): xofs[ow] | xa[ow][2] | yofs[oh] | ya[oh][2] | xmax
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_resize(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                    const int *__restrict__ tab, const uint16_t *__restrict__ order,
                                                    int h, int w, int oh, int ow) {
    const int f = order[blockIdx.y];
    const int *xofs = tab, *xa = tab + ow, *yofs = tab + 3 * ow, *ya = tab + 3 * ow + oh;
    const int xmax = tab[3 * ow + 3 * oh];
    const uint8_t *src = in + (size_t)f * h * w;
    uint8_t *dst = out + (size_t)f * oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < oh * ow; i += gridDim.x * 256) {
        const int oy = i / ow, ox = i - oy * ow;
        const int sy = yofs[oy];
        const uint8_t *r0 = src + (size_t)min(max(sy, 0), h - 1) * w, *r1 = src + (size_t)min(max(sy + 1, 0), h - 1) * w;
        const int sx = xofs[ox], sx1 = min(sx + 1, w - 1);
        int h0, h1;
        if (ox < xmax) {
            h0 = r0[sx] * xa[2 * ox] + r0[sx1] * xa[2 * ox + 1];
            h1 = r1[sx] * xa[2 * ox] + r1[sx1] * xa[2 * ox + 1];
        } else {
            h0 = r0[sx] * 2048;
            h1 = r1[sx] * 2048;
        }
        const int v = (((ya[2 * oy] * (h0 >> 4)) >> 16) + ((ya[2 * oy + 1] * (h1 >> 4)) >> 16) + 2) >> 2;
        dst[i] = (uint8_t)min(max(v, 0), 255);
    }
}

// centre of the non-zero samples of the nearest-neighbour shrunk map (sx = min(floor(x*factor), w-1)), times factor
__global__ __launch_bounds__(256) void k_centre_nearest(const uint8_t *__restrict__ maps, int h, int w, int sh, int sw,
                                                        int factor, double *__restrict__ xy) {
    const int f = blockIdx.x;
    const uint8_t *m = maps + (size_t)f * h * w;
    __shared__ unsigned long long red[3 * 4];
    unsigned long long cnt = 0, sr = 0, sc = 0;
    for (int i = threadIdx.x; i < sh * sw; i += 256) {
        const int r = i / sw, c = i - r * sw;
        if (m[(size_t)min(r * factor, h - 1) * w + min(c * factor, w - 1)]) { ++cnt; sr += r; sc += c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); sr += __shfl_xor(sr, o); sc += __shfl_xor(sc, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = cnt; red[4 + (threadIdx.x >> 6)] = sr; red[8 + (threadIdx.x >> 6)] = sc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = red[0] + red[1] + red[2] + red[3];
        sr = red[4] + red[5] + red[6] + red[7];
        sc = red[8] + red[9] + red[10] + red[11];
        if (cnt) {
            xy[2 * f] = (double)sc / (double)cnt * (double)factor;
            xy[2 * f + 1] = (double)sr / (double)cnt * (double)factor;
        } else {
            xy[2 * f] = xy[2 * f + 1] = __longlong_as_double(0x7FF8000000000000LL);
        }
    }
}

// com_km = 0 (sc_find_center_of_mass with km=False, smartVidCrop.py:1165-1178): the position of the first maximum of the
// final map in raster order; NaN for an all-zero map.  key = value << 16 | (65535 - index): its maximum is that pixel.
__global__ __launch_bounds__(256) void k_centre_argmax(const uint8_t *__restrict__ maps, int h, int w, FDiv dW, double *__restrict__ xy) {
    const int f = blockIdx.x, hw = h * w;
    const uint8_t *m = maps + (size_t)f * hw;
    __shared__ uint32_t red[4];
    uint32_t best = 0;
    for (int i = threadIdx.x; i < hw; i += 256) best = max(best, ((uint32_t)m[i] << 16) | (uint32_t)(65535 - i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(red[0], red[1]), max(red[2], red[3]));
        if (best >> 16) {
            uint32_t c;
            const uint32_t r = fdivmod(65535u - (best & 0xFFFFu), dW, c);
            xy[2 * f] = (double)c;
            xy[2 * f + 1] = (double)r;
        } else {
            xy[2 * f] = xy[2 * f + 1] = __longlong_as_double(0x7FF8000000000000LL);
        }
    }
}

// --------------------------------------------------------------------------------------
// k_compact: non-zero pixels in raster order
// --------------------------------------------------------------------------------------
// map <-> LDS copies, 8 bytes per lane when the map allows it (a 140x250 map does), bytes otherwise
__device__ __forceinline__ void copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int n) {
    if ((((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        // eight loads of a thread in flight before the first store: written as one load and one store per iteration, a map
        // (35 000 bytes, 4.3 words per thread) costs a memory round trip per word -- 5 x ~1.5 us at the head of k_compact,
        // k_core and k_finish each
        const int n8 = n >> 3;
        for (int i0 = threadIdx.x; i0 < n8; i0 += 8 * TB) {
            uint2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TB;
                v[u] = i < n8 ? ((const uint2 *)src)[i] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TB;
                if (i < n8) ((uint2 *)dst)[i] = v[u];
            }
        }
        for (int i = (n8 << 3) + threadIdx.x; i < n; i += TB) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n; i += TB) dst[i] = src[i];
    }
}

__device__ __forceinline__ void compact_body(const TailArgs &A) {
    const int f = A.order[blockIdx.x];
    __shared__ int lds16[NW16];
    extern __shared__ uint8_t sm_compact[];                  // the map: read once with whole lines, scanned from LDS
    const int hw = A.h * A.w;
    const uint8_t *gmap = A.maps + (size_t)f * hw;
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    uint32_t *pts = (uint32_t *)(ws + A.L.pts);
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    uint8_t *map = sm_compact;
    copy_bytes(map, gmap, hw);
    __syncthreads();
    const int per = (hw + TB - 1) / TB;
    const int lo = min(hw, (int)threadIdx.x * per), hi = min(hw, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += map[i] != 0;
    int total;
    int pos = block_excl_scan(cnt, lds16, &total);
    if (cnt) {
        uint32_t c; uint32_t r = fdivmod((uint32_t)lo, A.dW, c);     // one division per thread, then walk
        for (int i = lo; i < hi; ++i) {
            const uint32_t v = map[i];
            if (v) pts[pos++] = r | (c << 8) | (v << 16);
            if (++c == (uint32_t)A.w) { c = 0; ++r; }
        }
    }
    if (threadIdx.x == 0) {
        hdr[0] = total;
        hdr[1] = 0;
        hdr[2] = -1;
        hdr[3] = (A.clust_filt && total > A.mcs + 1) ? 1 : 0;
        hdr[23] = 0;                                                     // k_tree_par sets it when it has done the map's hierarchy
        hdr[30] = 0;                                                     // k_tail_back sets it when it has finished the map (zeroing, CLOSE, centre)
    }
}
__global__ __launch_bounds__(TB) void k_compact(TailArgs A) { compact_body(A); }

// --------------------------------------------------------------------------------------
// k_core: squared distance to the k-th nearest other point, three exact phases:
//  1. one thread per point walks the neighbour offsets (sorted by distance) of the inner
//     ring (d2 <= RING_R1^2) over the occupancy map in LDS — enough for points inside blobs;
//  2. one wavefront per remaining point scans the whole ring table 64 offsets at a time
//     (ballot + popcount), out to RING_R;
//  3. points with fewer than k neighbours within RING_R: wave-wide bisection on the count
//     of points within distance t over all N points.
// --------------------------------------------------------------------------------------
// k_core phase 3, maps of up to 64 * NI points: the smallest t with #{d2 <= t} >= k + 1 (self included) for the point (r, c),
// by bisection over the point's distances to all N points held in registers (NI per lane)
template <int NI>
__device__ __forceinline__ uint32_t core_kth_regs(const uint16_t *rcl, int N, int r, int c, int k, uint32_t lo, uint32_t hi) {
    const int lane = threadIdx.x & 63;
    uint32_t d[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = lane + 64 * i;
        d[i] = 0xFFFFFFFFu;
        if (j < N) {
            const uint32_t u = rcl[j];
            const int dr = (int)(u & 255) - r, dc = (int)(u >> 8) - c;
            d[i] = (uint32_t)(dr * dr + dc * dc);
        }
    }
    auto enough = [&](uint32_t t) -> bool {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) cnt += d[i] <= t;
        return wave_sum_i32(cnt) >= k + 1;
    };
    // the answer is usually within a few times the lower bound, the upper bound is the image diagonal: gallop first
    for (uint32_t t = 2 * lo; t < hi; t *= 2) {
        if (enough(t)) { hi = t; break; }
        lo = t + 1;
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (enough(mid)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ void core_body(const TailArgs &A) {
    const int f = A.order[blockIdx.x];
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    const int32_t *hdr = (const int32_t *)(ws + A.L.hdr);
    if (!hdr[3]) return;
    const int N = hdr[0];
    extern __shared__ uint8_t sm_core[];
    const int hw = A.h * A.w;
    uint8_t *occ = sm_core;                                   // [hw]
    uint32_t *ring = (uint32_t *)(sm_core + (hw + 15) / 16 * 16);   // [n_ring]
    int32_t *dl1 = (int32_t *)(ring + A.n_ring);                    // [n_ring1] linear form of the inner offsets
    __shared__ int n_fb, n_fb2;
    const uint8_t *map = A.maps + (size_t)f * hw;
    const long long tc0 = wall_clock64();
    copy_bytes(occ, map, hw);
    for (int i = threadIdx.x; i < A.n_ring; i += TB) ring[i] = A.ring[i];
    for (int i = threadIdx.x; i < A.n_ring1; i += TB) dl1[i] = A.ring_delta[i];
    if (threadIdx.x == 0) { n_fb = 0; n_fb2 = 0; }
    __syncthreads();
    const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
    uint32_t *core = (uint32_t *)(ws + A.L.core);
    int32_t *fb = (int32_t *)(ws + A.L.labels);               // scratch: points left for phase 2
    int32_t *fb2 = (int32_t *)(ws + A.L.reach);               // scratch: points left for phase 3
    int k = A.min_samples > 0 ? A.min_samples : A.mcs;
    k = min(N - 1, k);
    if (k == 0) k = 1;
    const int h = A.h, w = A.w;
    // phase 1: thread per point, ring offsets in groups of eight.  The offsets are the same for every thread, so
    // they (and their linear form dr*w + dc) are broadcast reads of a small LDS table, and the eight occupancy probes of a
    // group are independent LDS reads in flight together; only a group that contains the k-th neighbour is walked
    // in order.  A point at least RING_R1 away from every border needs no bounds tests: 3 vector instructions
    // per probe.
    const int nr1 = A.n_ring1;
    for (int pbase = 0; pbase < N; pbase += TB) {          // uniform trip count; lanes past N carry a dummy answer
        const int p = pbase + threadIdx.x;
        const bool live = p < N;
        const uint32_t v = live ? pts[p] : 0u;
        const int r = v & 255, c = (v >> 8) & 255;
        const bool interior = r >= RING_R1 && r < h - RING_R1 && c >= RING_R1 && c < w - RING_R1;
        const uint8_t *op = occ + r * w + c;
        int cnt = 0;
        uint32_t res = live ? 0xFFFFFFFFu : 0u;
        // (a wave-uniform loop, left when every lane has its answer: i0 stays scalar, and so do the table reads)
        for (int i0 = 0; i0 < nr1 && __ballot(res == 0xFFFFFFFFu) != 0ull; i0 += 8) {
            bool hit[8];
            if (interior) {
#pragma unroll
                for (int j = 0; j < 8; ++j) hit[j] = (i0 + j < nr1) && op[dl1[min(i0 + j, nr1 - 1)]] != 0;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t o = ring[min(i0 + j, nr1 - 1)];
                    const int rr = r + (int)(o & 255) - 128, cc = c + (int)((o >> 8) & 255) - 128;
                    const bool in = (i0 + j < nr1) && (unsigned)rr < (unsigned)h && (unsigned)cc < (unsigned)w;
                    hit[j] = in && occ[in ? rr * w + cc : 0] != 0;
                }
            }
            int add = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) add += hit[j] ? 1 : 0;
            if (res != 0xFFFFFFFFu) continue;
            if (cnt + add >= k) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (res == 0xFFFFFFFFu && hit[j] && ++cnt == k) res = ring[min(i0 + j, nr1 - 1)] >> 16;
            } else {
                cnt += add;
            }
        }
        if (!live) continue;
        if (res != 0xFFFFFFFFu) core[p] = res;
        else fb[atomicAdd(&n_fb, 1)] = p;
    }
    __syncthreads();
    int32_t *stamp = (int32_t *)(ws + A.L.hdr);                 // [5] phase 1, [6] phase 2, [7] end (10 ns units; debug door)
    if (threadIdx.x == 0) stamp[5] = (int)(wall_clock64() - tc0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int nf = n_fb;
    for (int q = wave; q < nf; q += NW16) {
        const int p = fb[q];
        const uint32_t v = pts[p];
        const int r = v & 255, c = (v >> 8) & 255;
        int cum = 0;
        bool done = false;
        for (int base = 0; base < A.n_ring && !done; base += 256) {     // four independent probe chains per lane
            bool hit[4];
            uint32_t o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + 64 * u + lane;
                hit[u] = false;
                o[u] = 0;
                if (i < A.n_ring) {
                    o[u] = ring[i];
                    const int rr = r + (int)(o[u] & 255) - 128, cc = c + (int)((o[u] >> 8) & 255) - 128;
                    hit[u] = (unsigned)rr < (unsigned)h && (unsigned)cc < (unsigned)w && occ[rr * w + cc];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (done) break;
                const unsigned long long bal = __ballot(hit[u]);
                const int cnt = __popcll(bal);
                if (cum + cnt >= k) {
                    if (hit[u] && __popcll(bal & lt) == k - cum - 1) core[p] = o[u] >> 16;
                    done = true;
                } else {
                    cum += cnt;
                }
            }
        }
        if (!done && lane == 0) fb2[atomicAdd(&n_fb2, 1)] = p;
    }
    __syncthreads();
    const int nf2 = n_fb2;
    if (threadIdx.x == 0) { stamp[6] = (int)(wall_clock64() - tc0); stamp[7] = stamp[6]; }
    if (nf2 == 0) return;
    // phase 3 counts over all points many times: their coordinates go to LDS (over the occupancy map and the ring
    // table, which are no longer needed) when they fit
    uint16_t *rcl = (uint16_t *)sm_core;
    const bool in_lds = (size_t)N * 2 <= (size_t)(hw + 15) / 16 * 16 + (size_t)A.n_ring * 4;
    if (in_lds) {
        for (int j = threadIdx.x; j < N; j += TB) rcl[j] = (uint16_t)(pts[j] & 0xFFFFu);
        __syncthreads();
    }
    const uint32_t maxd = (uint32_t)((h - 1) * (h - 1) + (w - 1) * (w - 1));
    for (int q = wave; q < nf2; q += NW16) {
        const int p = fb2[q];
        const uint32_t v = pts[p];
        const int r = v & 255, c = (v >> 8) & 255;
        uint32_t lo = RING_R * RING_R + 1, hi = maxd;         // smallest t with #{d2 <= t} >= k+1 (self included)
        if (in_lds && N <= 64 * 32) {
            // all N distances of this point in registers (8, 16, 24 or 32 per lane), the bisection then touches no memory
            const int ni = (N + 63) >> 6;
            const uint32_t ans = ni <= 8 ? core_kth_regs<8>(rcl, N, r, c, k, lo, hi) : ni <= 16 ? core_kth_regs<16>(rcl, N, r, c, k, lo, hi)
                               : ni <= 24 ? core_kth_regs<24>(rcl, N, r, c, k, lo, hi) : core_kth_regs<32>(rcl, N, r, c, k, lo, hi);
            if (lane == 0) core[p] = ans;
            continue;
        }
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            int cnt = 0;
            if (in_lds) {
                for (int j = lane; j < N; j += 64) {
                    const uint32_t u = rcl[j];
                    const int dr = (int)(u & 255) - r, dc = (int)(u >> 8) - c;
                    cnt += (uint32_t)(dr * dr + dc * dc) <= mid;
                }
            } else {
                for (int j = lane; j < N; j += 64) {
                    const uint32_t u = pts[j];
                    const int dr = (int)(u & 255) - r, dc = (int)((u >> 8) & 255) - c;
                    cnt += (uint32_t)(dr * dr + dc * dc) <= mid;
                }
            }
            cnt = wave_sum_i32(cnt);
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        if (lane == 0) core[p] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) stamp[7] = (int)(wall_clock64() - tc0);
}
__global__ __launch_bounds__(TB) void k_core(TailArgs A) { core_body(A); }

// --------------------------------------------------------------------------------------
// k_prim: the library's Prim over the mutual-reachability graph, start node 0, lowest
// index wins ties, edge = (last added node, new node, weight).  Each thread keeps the
// running reachability and core distance of PT points in registers; coordinates live
// in LDS; one min-reduction of (reach | index) per step with the winner's core distance
// as payload, one barrier per step.  For N <= 32768 the key fits 32 bits and the
// wavefront reduction is six DPP-fused v_min_u32; larger N use a 64-bit key.
// --------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, CTRL, ROW_MASK, 0xF, false);
    return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_u32<0x111, 0xF>(v);      // row_shr:1
    v = dpp_min_u32<0x112, 0xF>(v);      // row_shr:2
    v = dpp_min_u32<0x114, 0xF>(v);      // row_shr:4
    v = dpp_min_u32<0x118, 0xF>(v);      // row_shr:8   -> lane 15 of every row holds the row minimum
    v = dpp_min_u32<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_min_u32<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int PT>
__device__ __forceinline__ void prim_regs32(const uint32_t *__restrict__ core_g, hdb::Edge *__restrict__ mst, int N,
                                            const uint16_t *rc16, uint4 *slots) {
    // Thread t owns points t*PT .. t*PT+PT-1 entirely in registers (coordinates, core distance,
    // reachability); wavefronts that own no point leave, so the barrier spans only ceil(N/(64*PT)) waves.
    // A point that joined the tree gets core = reach = REACH_INF: its key sorts behind every live one and
    // no update can lower it, so the inner loop has no liveness test.  The winner's core distance and
    // coordinates travel with its key, so one step has a single LDS round trip (slot write, barrier, read).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (N + 64 * PT - 1) / (64 * PT);
    for (int i = tid; i < 2 * NW16; i += TB) slots[i] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    __syncthreads();
    if (wave >= nw) return;
    uint32_t reach[PT], corev[PT], rcv[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int p = tid * PT + i;
        reach[i] = REACH_INF;
        corev[i] = REACH_INF;
        rcv[i] = 0;
        if (p < N) { corev[i] = core_g[p]; rcv[i] = rc16[p]; }
    }
    if (tid == 0) corev[0] = REACH_INF;                    // point 0 starts the tree
    uint32_t cur = 0;
    const uint32_t cv0 = rc16[0];
    int cr = cv0 & 255, cc = cv0 >> 8;
    uint32_t ccore = core_g[0];
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
        const uint32_t wmin = wave_min_u32(best);
        uint4 *sl = slots + (step & 1) * NW16;
        if (best == wmin) sl[wave] = make_uint4(wmin, bcore, brc, 0);   // keys are unique: one lane
        __syncthreads();
        // second level: lanes 0..15 of every row hold one slot each; row minimum by DPP, the winner's
        // payload by readlane
        const uint4 t = sl[lane & (NW16 - 1)];
        uint32_t k2 = t.x;
        k2 = dpp_min_u32<0x111, 0xF>(k2);
        k2 = dpp_min_u32<0x112, 0xF>(k2);
        k2 = dpp_min_u32<0x114, 0xF>(k2);
        k2 = dpp_min_u32<0x118, 0xF>(k2);
        const uint32_t kmin = (uint32_t)__builtin_amdgcn_readlane((int)k2, 15);
        const int src = __ffsll((unsigned long long)__ballot(t.x == kmin)) - 1;
        const uint32_t nidx = kmin & 0x7FFFu;
        if (tid == 0) mst[step] = hdb::Edge{(uint16_t)cur, (uint16_t)nidx, kmin >> 15};
        if ((int)(nidx / PT) == tid) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
                if ((int)(nidx % PT) == i) { corev[i] = REACH_INF; reach[i] = REACH_INF; }
        }
        ccore = (uint32_t)__builtin_amdgcn_readlane((int)t.y, src);
        const uint32_t cv = (uint32_t)__builtin_amdgcn_readlane((int)t.z, src);
        cr = cv & 255; cc = cv >> 8;
        cur = nidx;
    }
}

// generic path for very large N: 64-bit keys, reachability kept in global memory
__device__ __forceinline__ void prim_global(const uint32_t *__restrict__ core_g, uint32_t *__restrict__ reach_g,
                                            hdb::Edge *__restrict__ mst, int N, const uint16_t *rc16,
                                            unsigned long long *slots) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < N; p += TB) reach_g[p] = REACH_INF;        // bit 31 marks "in tree"
    if (tid == 0) reach_g[0] = 0x80000000u;
    __syncthreads();
    uint32_t cur = 0, cv = rc16[0];
    int cr = cv & 255, cc = cv >> 8;
    uint32_t ccore = core_g[0];
    for (int step = 0; step < N - 1; ++step) {
        unsigned long long best = ~0ull;
        for (int p = tid; p < N; p += TB) {
            uint32_t rv = reach_g[p];
            if (rv & 0x80000000u) continue;
            const uint32_t v = rc16[p], cj = core_g[p];
            const int dr = (int)(v & 255) - cr, dc = (int)(v >> 8) - cc;
            uint32_t m = max(max((uint32_t)(dr * dr + dc * dc), cj), ccore);
            if (m < rv) { rv = m; reach_g[p] = rv; }
            const unsigned long long key = ((unsigned long long)rv << 33) | ((unsigned long long)p << 17) | cj;
            best = key < best ? key : best;
        }
        best = wave_min_u64(best);
        unsigned long long *sl = slots + (step & 1) * NW16;
        if (lane == 0) sl[wave] = best;
        __syncthreads();
        unsigned long long w = sl[0];
#pragma unroll
        for (int i = 1; i < NW16; ++i) { unsigned long long t = sl[i]; w = t < w ? t : w; }
        const uint32_t nidx = (uint32_t)(w >> 17) & 0xFFFFu;
        if (tid == 0) mst[step] = hdb::Edge{(uint16_t)cur, (uint16_t)nidx, (uint32_t)(w >> 33)};
        if ((int)(nidx & (TB - 1)) == tid) reach_g[nidx] = 0x80000000u;   // same thread re-reads it next step
        ccore = (uint32_t)w & 0x1FFFFu;
        cv = rc16[nidx];
        cr = cv & 255; cc = cv >> 8;
        cur = nidx;
    }
}

// The register-resident variant (points per thread) the legacy Prim uses for a map of N points: the first of 2, 4, 8
// that holds N and is >= the handle's SVC_PRIM_PT, else 16 / 32, else 0 = the global-memory form.
__device__ __forceinline__ int prim_variant(int N, int pt) {
    if (N <= 2 * TB && pt <= 2) return 2;
    if (N <= 4 * TB && pt <= 4) return 4;
    if (N <= 8 * TB && pt <= 8) return 8;
    if (N <= 16 * TB) return 16;
    if (N <= 32 * TB) return 32;
    return 0;
}

struct PrimFrame {
    uint8_t *ws; int N; uint16_t *rc16; unsigned long long *slots;
    const uint32_t *core; hdb::Edge *mst;
};
// common prologue of the legacy kernels: false = nothing to do for this map
__device__ __forceinline__ bool prim_frame(const TailArgs &A, uint8_t *sm, PrimFrame &P, int n_min) {
    P.ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    const int32_t *hdr = (const int32_t *)(P.ws + A.L.hdr);
    if (!hdr[3]) return false;
    P.N = hdr[0];
    if (P.N <= n_min) return false;
    P.slots = (unsigned long long *)sm;                               // [2][NW16] (uint4 or u64)
    P.rc16 = (uint16_t *)(sm + 2 * NW16 * 16);                        // [N]
    P.core = (const uint32_t *)(P.ws + A.L.core);
    P.mst = (hdb::Edge *)(P.ws + A.L.mst);
    return true;
}
__device__ __forceinline__ void prim_stage_points(const TailArgs &A, const PrimFrame &P) {
    const uint32_t *pts = (const uint32_t *)(P.ws + A.L.pts);
    for (int p = threadIdx.x; p < P.N; p += TB) P.rc16[p] = (uint16_t)(pts[p] & 0xFFFFu);
    __syncthreads();
}

// Legacy Prim (one node per step), one kernel per register-resident variant so that each has its own register
// budget: k_prim_pt<2|4|8> carry no private segment (profiles/r03_kernel_resources.txt); launched back to back, a
// map is taken by the kernel whose variant it is.  n_min: maps of at most n_min points are left to k_prim_lvl.
template <int PT>
__global__ __launch_bounds__(TB) void k_prim_pt(TailArgs A, int n_min) {
    extern __shared__ uint8_t sm_prim[];
    PrimFrame P;
    if (!prim_frame(A, sm_prim, P, n_min)) return;
    if (prim_variant(P.N, A.prim_pt) != PT) return;
    prim_stage_points(A, P);
    const long long t0 = wall_clock64();
    prim_regs32<PT>(P.core, P.mst, P.N, P.rc16, (uint4 *)P.slots);
    if (threadIdx.x == 0) ((int32_t *)(P.ws + A.L.hdr))[12] = (int)(wall_clock64() - t0);
}

// maps of more than 8192 points: 16 / 32 points per thread (these spill: the price of N up to 32768 in registers) or
// reachability in global memory
__global__ __launch_bounds__(TB) void k_prim_big(TailArgs A, int n_min) {
    extern __shared__ uint8_t sm_prim[];
    PrimFrame P;
    if (!prim_frame(A, sm_prim, P, n_min)) return;
    const int v = prim_variant(P.N, A.prim_pt);
    if (v != 16 && v != 32 && v != 0) return;
    prim_stage_points(A, P);
    const long long t0 = wall_clock64();
    if (v == 16) prim_regs32<16>(P.core, P.mst, P.N, P.rc16, (uint4 *)P.slots);
    else if (v == 32) prim_regs32<32>(P.core, P.mst, P.N, P.rc16, (uint4 *)P.slots);
    else prim_global(P.core, (uint32_t *)(P.ws + A.L.reach), P.mst, P.N, P.rc16, P.slots);
    if (threadIdx.x == 0) ((int32_t *)(P.ws + A.L.hdr))[12] = (int)(wall_clock64() - t0);
}

// --------------------------------------------------------------------------------------
// k_prim_lvl: the same Prim -- identical (last node, new node, weight) sequence -- emitted in ROUNDS of up to 64
// nodes (tools/sim/prim_levels.py is the executable specification, checked against oracle/hdbscan_ref.prim_mst).
//
// On a pixel grid the mutual-reachability weights are small integers and the library's Prim spends nearly all of its
// N - 1 steps on plateaus: the minimum reach m of the points outside the tree stays put while the points of reach m
// are taken in index order (first minimum wins).  State: the level m; F = bitmap (by point index) of the points
// outside the tree whose reach is m; R = each point's exact reach against the first `done` tree nodes, in registers,
// caught up only when the level has to rise.
//   round   the first 64 members of F, f1 < f2 < ..., are candidates (lane i of the four working wavefronts = candidate
//           i; each of them walks a quarter of the disc's cells).  Candidate i probes the grid disc d2 <= m around
//           itself through the occupancy bitmap (cell -> point index = prefix count + popcount): a point outside
//           the tree with mr = max(d2, core_j, core_i) < m is a DROP (once f_i is in the tree the level falls), one
//           with mr == m that is not in F is an ENTRANT.  Adding f1..fi leaves f(i+1) the library's next pick iff
//           there was no drop and no entrant so far has a smaller index than f(i+1): the accepted prefix ends at the
//           first i that breaks this (a prefix minimum over the candidates' lowest entrants).
//   commit  accepted nodes get their edges (weight m), join the tree (top bit of their core distance), leave F;
//           their entrants join F.  After a drop F is rebuilt from the last node's disc at the new, lower level.
//   rise    F empty: the nodes added since the last rise become batches of <= 64 with a bounding box; a (batch, chunk of
//           64 points) block is relaxed -- all pairs, the nodes through scalar registers -- only when its lower bound
//           max(box distance^2, smallest core distances) is <= the minimum of R before the rise (other blocks stay
//           pending; a jump between blobs first tightens the bound with every chunk's nearest batch); m = block
//           minimum of R, F = {R == m}.
//   m > RING_R^2 (jumps between far-apart blobs): one node per round, then a rise -- the legacy algorithm's step.
// Golden maps (N = 660 .. 2 980): 28 .. 81 rounds and 6 .. 15 rises per map instead of N - 1 steps.
// --------------------------------------------------------------------------------------
#ifndef LVL_CAP
#define LVL_CAP 8192                   // points per map (above: k_prim_lvl_big)
#endif
static_assert(LVL_CAP % TB == 0, "k_prim_lvl keeps LVL_CAP / TB points per thread");
#define LVL_PT (LVL_CAP / TB)
#define LVL_TREE 0x80000000u
#define LVL_NONE 0xFFFFFFFFu
#define LVL_RINF 0x7FFFFFFFu
#define LVL_RING2 (RING_R * RING_R)
#define LVL_PAD RING_R                 // the occupancy grid carries a border of empty cells: discs need no bounds tests
#define LVL_NB 256                     // batches of tree nodes a map can have pending (then everything is settled at once)
#define LVL_NEAR 64u                   // a rise whose bound exceeds this is a jump between blobs: the bound is tightened first

#ifndef LVL_WORKERS
#define LVL_WORKERS 8                   // wavefronts that run the rounds (lane = candidate), each probing its share of the candidates' discs.  Round 6: 8 instead of 4
#endif                                  // (two per SIMD): a round's probe is one group of LDS gathers instead of two -- the Prim 313 -> 273 us at 1 800 points, 1 036 -> 935 at 10 000; 2: 404
static_assert(LVL_WORKERS >= 1 && LVL_WORKERS <= NW16 && (LVL_WORKERS & (LVL_WORKERS - 1)) == 0, "LVL_WORKERS: a power of two of at most the workgroup's wavefronts (every worker probes its share of the disc)");
struct OccW { uint32_t bits, base; };                                  // 32 grid cells: occupancy, points before them
typedef short lvl_s2 __attribute__((ext_vector_type(2)));

static size_t lvl_lds_bytes(int h, int w, int n_ring) {
    const size_t cap = (size_t)std::min(h * w, LVL_CAP);
    const size_t cells = (size_t)(h + 2 * LVL_PAD) * (w + 2 * LVL_PAD);
    auto up = [](size_t b) { return (b + 15) / 16 * 16; };
    return up((cells + 31) / 32 * sizeof(OccW)) + up(cap * 8) + up(cap * 4) + up((cap + 63) / 64 * 8) +
           up((size_t)n_ring * 8) + up(cap * 2) + up(64 * LVL_WORKERS * 4) + up(64 * LVL_WORKERS * 8) + up(2 * NW16 * 4) + up(16) +
           up(LVL_NB * 16) + up((cap + 63) / 64 * (LVL_NB / 32) * 4) + up((LVL_RING2 + 1) * 2) + 64;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int j) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), j);
    return ((unsigned long long)hi << 32) | lo;
}
// inclusive prefix minimum over the 64 lanes (the scan wave_min_u32 ends with a readlane of)
__device__ __forceinline__ uint32_t wave_prefix_min_u32(uint32_t v) {
    v = dpp_min_u32<0x111, 0xF>(v);
    v = dpp_min_u32<0x112, 0xF>(v);
    v = dpp_min_u32<0x114, 0xF>(v);
    v = dpp_min_u32<0x118, 0xF>(v);
    v = dpp_min_u32<0x142, 0xA>(v);
    v = dpp_min_u32<0x143, 0xC>(v);
    return v;
}

// inclusive prefix MAXIMUM over the 64 lanes (wave_incl_scan_u32's DPP pattern; identity 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
    return t > v ? t : v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v) {
    v = dpp_max_u32<0x111, 0xF>(v);
    v = dpp_max_u32<0x112, 0xF>(v);
    v = dpp_max_u32<0x114, 0xF>(v);
    v = dpp_max_u32<0x118, 0xF>(v);
    v = dpp_max_u32<0x142, 0xA>(v);
    v = dpp_max_u32<0x143, 0xC>(v);
    return v;
}
// position of the n-th (0-based) set bit of the 64-bit word hi:lo (n < its population)
__device__ __forceinline__ int select_bit64(uint32_t lo, uint32_t hi, int n) {
    const int clo = __popc(lo);
    uint32_t w = lo;
    int pos = 0;
    if (n >= clo) { w = hi; n -= clo; pos = 32; }
    int c = __popc(w & 0xFFFFu);
    if (n >= c) { n -= c; pos += 16; w >>= 16; }
    c = __popc(w & 0xFFu);
    if (n >= c) { n -= c; pos += 8; w >>= 8; }
    c = __popc(w & 0xFu);
    if (n >= c) { n -= c; pos += 4; w >>= 4; }
    c = __popc(w & 0x3u);
    if (n >= c) { n -= c; pos += 2; w >>= 2; }
    if (n >= (int)(w & 1u)) pos += 1;
    return pos;
}
// The first 64 members of the bitmap F (NF64 words of 64 points) in index order -> cw[0 ..]; returns how many (<= 64).  Called by a
// whole wavefront; `heads` = 64 words of LDS scratch of that wavefront.
// Round 6: RANK-parallel.  Lane k holds word k and the prefix sum of the populations tells where its members start; every word that
// has members writes its lane number at the rank of its first one, a running maximum over the ranks gives every rank r its word, and
// lane r then picks the (r - start)-th set bit of that word (five popcount halvings): no loop over the words.  The loop it replaces
// took one pass per word that holds members -- the members of a level are a frontier, spread over 20 - 30 words at 1 800 points --:
// 0.85 us of a 2.3 us round.  (Round 3's other parallel form, every lane writing ITS OWN word's bits one after the other, was the
// slower one: that is a loop over the bits.)
__device__ __forceinline__ int lvl_extract(const unsigned long long *F64, int NF64, uint32_t *cw, uint32_t *heads, uint32_t &mycand,
                                           bool have_w0 = false, unsigned long long w0 = 0ull) {
    const int lane = threadIdx.x & 63;
    int ncand = 0;
#ifdef LVL_EXTRACT_LOOP                                     // rounds 3-5 (A/B builds): one pass per word that holds members, all lanes placing that word's members
    for (int wb = 0; wb < NF64 && ncand < 64; wb += 64) {
        const int k = wb + lane;
        const unsigned long long W = k < NF64 ? F64[k] : 0ull;
        unsigned long long nz = __ballot(W != 0ull);
        while (nz && ncand < 64) {
            const int src = __builtin_ctzll(nz);
            nz &= nz - 1ull;
            const unsigned long long Wk = readlane_u64(W, src);
            if ((Wk >> lane) & 1ull) {
                const int rank = ncand + __popcll(Wk & ((1ull << lane) - 1ull));
                if (rank < 64) cw[rank] = (uint32_t)((wb + src) * 64 + lane);
            }
            ncand += __popcll(Wk);
        }
    }
    __builtin_amdgcn_wave_barrier();
    ncand = min(ncand, 64);
    mycand = lane < ncand ? cw[lane] : LVL_NONE;
    return ncand;
#endif
    for (int wb = 0; wb < NF64 && ncand < 64; wb += 64) {
        const int k = wb + lane;
        const unsigned long long W = (have_w0 && wb == 0) ? w0 : (k < NF64 ? F64[k] : 0ull);      // have_w0: word `lane` of F as the caller read it (nothing has written F since)
        const uint32_t pc = (uint32_t)__popcll(W);
        const uint32_t incl = wave_incl_scan_u32(pc), excl = incl - pc;
        const int total = __builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0) continue;
        heads[lane] = 0u;
        __builtin_amdgcn_wave_barrier();                    // (a wavefront's LDS accesses are served in program order; the fences are for the compiler)
        if (pc != 0u && excl < 64u) heads[excl] = (uint32_t)lane + 1u;
        __builtin_amdgcn_wave_barrier();
        const uint32_t head = wave_incl_scan_max_u32(heads[lane]);
        const int src = head ? (int)head - 1 : 0;            // the word that holds rank `lane` (of this pass)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)W);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(W >> 32));
        const uint32_t ex = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)excl);
        const int rank = ncand + lane;
        const uint32_t cand = (uint32_t)((wb + src) * 64 + select_bit64(lo, hi, lane - (int)ex));
        if (NF64 <= 64) {                                   // one pass (maps of up to 4 096 points): rank = lane, the candidate stays in its register
            mycand = lane < total ? cand : LVL_NONE;
            return min(total, 64);
        }
        if (lane < total && rank < 64) cw[rank] = cand;
        ncand += total;
    }
    __builtin_amdgcn_wave_barrier();
    ncand = min(ncand, 64);
    mycand = lane < ncand ? cw[lane] : LVL_NONE;
    return ncand;
}

struct LvlLds {
    OccW *occ;              // padded grid, 32 cells per word
    uint2 *tnode;           // tree nodes in the order they joined: (row | col << 16, core distance)
    uint32_t *corei;        // core distance per point; top bit = in the tree
    uint32_t *F;            // bitmap over point indices (32-bit words; read in pairs by the extraction)
    uint2 *ring;            // (d2, linear offset in the padded grid) per ring offset, ascending d2
    uint16_t *rc;           // row | col << 8 per point
    uint32_t *cand;         // [workers][64] candidates of the round (every worker extracts its own copy)
    uint2 *slot;            // [workers][64] (drop: smallest mr below the level or NONE, lowest entrant or NONE) per candidate, per worker's share of the disc
    uint32_t *red;          // [2][NW16] block reductions (alternating halves)
    int *ctl;               // outcome of a round for the wavefronts that do not work in it: accepted, dropped, level, last node
    uint4 *btab;            // [LVL_NB] batches of tree nodes: (start | len << 16, box, smallest core distance, -)
    uint32_t *proc;         // [chunks][LVL_NB / 32] batch already relaxed against the chunk
    uint16_t *rcnt;         // [RING_R^2 + 1] ring offsets with d2 <= index
    int gw;                 // padded grid width
};


// The disc d2 <= level around the candidate of every lane (on = the lane has one), the ring cells k = wave, wave +
// LVL_WORKERS, ... of it (the other workers take the rest).  The ring entries travel through scalar registers; the
// dependent LDS reads of an iteration (occupancy word; core distance and F word of the point found) are gathers.
//   MARK = false: dmin = smallest mr below the level (a drop), nmin = lowest index with mr == level outside F (an
//                 entrant), entmask bit it = iteration it found an entrant (iterations < 32).
//   MARK = true:  F |= the points outside the tree whose mr is exactly `level` (tree members carry the top bit in
//                 their core distance, so their mr never equals a level).
//   rcv / corev: the candidate's own row | col << 8 and core distance word as read here (the commit needs them again);
//   j4: the point indices found in the worker's first four cells (the entrants among them are marked without looking them up again).
template <bool MARK>
__device__ __forceinline__ void lvl_walk(const LvlLds &S, int wave, int nk, uint32_t level, bool on, uint32_t cand, int gw,
                                         uint32_t &dmin, uint32_t &nmin, uint32_t &entmask, uint32_t &rcv, uint32_t &corev, uint32_t (&j4)[4]) {
    const int lane = threadIdx.x & 63;
    dmin = nmin = LVL_NONE;
    entmask = 0;
    const uint32_t c0 = on ? cand : 0u;
    const uint32_t v = S.rc[c0];
    const uint32_t cell0 = on ? ((v & 255) + LVL_PAD) * (uint32_t)gw + (v >> 8) + LVL_PAD : 0u;   // cell 0: in the empty border
    const uint32_t craw = S.corei[c0];
    const uint32_t ci = on ? (craw & ~LVL_TREE) : LVL_NONE;
    rcv = v; corev = craw;
#pragma unroll
    for (int u = 0; u < 4; ++u) j4[u] = 0u;
    const int nit = (nk - wave + LVL_WORKERS - 1) / LVL_WORKERS;       // this worker's cells
    for (int ib = 0; ib < nit; ib += 64) {
        const int kk = LVL_WORKERS * (ib + lane) + wave;
        const uint2 mine = S.ring[kk < nk ? kk : 0];
        const int n64 = min(64, nit - ib);
        // four ring cells at a time: their occupancy reads go out together, then the core / F reads of the points found
        for (int it0 = 0; it0 < n64; it0 += 4) {
            uint32_t d2[4], cell[4], j[4], cj[4], fw[4];
            OccW ow[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = min(it0 + u, n64 - 1);
                d2[u] = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, it);
                cell[u] = (it0 + u < n64) ? cell0 + (uint32_t)__builtin_amdgcn_readlane((int)mine.y, it) : 0u;
                ow[u] = S.occ[cell[u] >> 5];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                in[u] = on && (it0 + u < n64) && ((ow[u].bits >> (cell[u] & 31u)) & 1u);
                j[u] = in[u] ? ow[u].base + (uint32_t)__popc(ow[u].bits & ((1u << (cell[u] & 31u)) - 1u)) : 0u;
                cj[u] = S.corei[j[u]];
                if (!MARK) fw[u] = S.F[j[u] >> 5];
                if (!MARK && ib + it0 == 0) j4[u] = j[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t mr = max(max(d2[u], cj[u]), ci);
                if (MARK) {
                    if (in[u] && mr == level) atomicOr(&S.F[j[u] >> 5], 1u << (j[u] & 31u));
                } else {
                    if (in[u] && mr < level) dmin = min(dmin, mr);
                    if (in[u] && mr == level && !((fw[u] >> (j[u] & 31u)) & 1u)) {
                        nmin = min(nmin, j[u]);
                        if (ib + it0 + u < 32) entmask |= 1u << (ib + it0 + u);
                    }
                }
            }
        }
    }
}

// One batch of <= 64 tree nodes (positions start .. start + len in tnode) against one chunk of 64 points (one per
// lane): r = min(r, max(d2, core_j, core_t)).  Every lane reads the node from LDS itself (one address for the whole
// wavefront: a broadcast read on the LDS pipe, two nodes per ds_read_b128) -- the earlier form, nodes read once by the
// lanes and handed round with v_readlane, spent 2 of its 6.5 VALU slots per node (+ the SGPR hazard nops) on that.
__device__ __forceinline__ uint32_t lvl_relax_block(const LvlLds &S, int start, int len, uint32_t rcv, uint32_t cj, uint32_t r) {
    const uint2 *tn = S.tnode + start;
    if (len == 64) {
#pragma unroll 4          // (8 and 16 cost the kernel its last registers: 2-3 spills)
        for (int j = 0; j < 64; ++j) {
            const uint2 t = tn[j];
            const lvl_s2 d = __builtin_bit_cast(lvl_s2, rcv) - __builtin_bit_cast(lvl_s2, t.x);
            r = min(r, max(max((uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false), cj), t.y));
        }
        return r;
    }
#pragma unroll 4
    for (int j = 0; j < len; ++j) {
        const uint2 t = tn[j];
        const lvl_s2 d = __builtin_bit_cast(lvl_s2, rcv) - __builtin_bit_cast(lvl_s2, t.x);
        r = min(r, max(max((uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false), cj), t.y));
    }
    return r;
}

// box = rmin | rmax << 8 | cmin << 16 | cmax << 24: squared distance between two boxes (0 when they overlap)
__device__ __forceinline__ uint32_t lvl_box_d2(uint32_t a, uint32_t b) {
    const int ar0 = a & 255, ar1 = (a >> 8) & 255, ac0 = (a >> 16) & 255, ac1 = a >> 24;
    const int br0 = b & 255, br1 = (b >> 8) & 255, bc0 = (b >> 16) & 255, bc1 = b >> 24;
    const int dr = max(0, max(br0 - ar1, ar0 - br1)), dc = max(0, max(bc0 - ac1, ac0 - bc1));
    return (uint32_t)(dr * dr + dc * dc);
}

// One sweep of a rise over one chunk: relax the pending batches whose lower bound max(box distance^2, smallest core
// distances) is <= limit (NEAREST: only the one with the smallest bound).  proc = the chunk's "already relaxed" bits.
template <bool NEAREST>
__device__ __forceinline__ uint32_t lvl_sweep_chunk(const LvlLds &S, const uint4 *btab, uint32_t *proc, int nb, uint32_t limit, bool all,
                                                    uint4 cb, uint32_t rcv, uint32_t cj, uint32_t r) {
    const int lane = threadIdx.x & 63;
    uint32_t bestkey = LVL_NONE;
    for (int bb = 0; bb < nb; bb += 64) {
        const int b = bb + lane;
        uint32_t lb = LVL_NONE;
        if (b < nb && !((proc[b >> 5] >> (b & 31)) & 1u)) {
            const uint4 e = btab[b];
            lb = all ? 0u : max(max(lvl_box_d2(e.y, cb.x), e.z), cb.y);
        }
        if (NEAREST) {
            if (lb <= limit) bestkey = min(bestkey, (min(lb, 0xFFFFFFu) << 8) | (uint32_t)b);
            continue;
        }
        unsigned long long todo = __ballot(lb <= limit && lb != LVL_NONE);
        if (lane == 0 && todo) {
            proc[bb >> 5] |= (uint32_t)todo;
            if (bb + 32 < LVL_NB) proc[(bb >> 5) + 1] |= (uint32_t)(todo >> 32);
        }
        while (todo) {
            const int b0 = bb + __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t sl = btab[b0].x;
            r = lvl_relax_block(S, (int)(sl & 0xFFFFu), (int)(sl >> 16), rcv, cj, r);
        }
    }
    if (NEAREST) {
        bestkey = wave_min_u32(bestkey);
        if (bestkey != LVL_NONE) {
            const int b0 = (int)(bestkey & 255u);
            if (lane == 0) proc[b0 >> 5] |= 1u << (b0 & 31);
            const uint32_t sl = btab[b0].x;
            r = lvl_relax_block(S, (int)(sl & 0xFFFFu), (int)(sl >> 16), rcv, cj, r);
        }
    }
    return r;
}

__device__ __forceinline__ void prim_lvl_body(const TailArgs &A) {
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (!hdr[3]) return;
    const int N = hdr[0];
    if (N > LVL_CAP) return;                                          // k_prim_big takes it
    extern __shared__ uint8_t sm_lvl[];
    __shared__ int lds16[NW16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gw = A.w + 2 * LVL_PAD, gcells = (A.h + 2 * LVL_PAD) * gw, nocc = (gcells + 31) >> 5;
    const int NF64 = (N + 63) >> 6;
    const int cap = min(A.h * A.w, LVL_CAP);
    LvlLds S;
    {
        uint8_t *p = sm_lvl;
        S.occ = carve<OccW>(p, nocc);
        S.tnode = carve<uint2>(p, cap);
        S.corei = carve<uint32_t>(p, cap);
        S.F = carve<uint32_t>(p, (size_t)((cap + 63) / 64) * 2);
        S.ring = carve<uint2>(p, A.n_ring);
        S.rc = carve<uint16_t>(p, cap);
        S.cand = carve<uint32_t>(p, 64 * LVL_WORKERS);
        S.slot = carve<uint2>(p, 64 * LVL_WORKERS);
        S.red = carve<uint32_t>(p, 2 * NW16);
        S.ctl = carve<int>(p, 4);
        S.btab = carve<uint4>(p, LVL_NB);
        S.proc = carve<uint32_t>(p, (size_t)((cap + 63) / 64) * (LVL_NB / 32));
        S.rcnt = carve<uint16_t>(p, LVL_RING2 + 1);
        S.gw = gw;
    }
    const long long t0 = wall_clock64();
    const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
    const uint32_t *core_g = (const uint32_t *)(ws + A.L.core);
    hdb::Edge *mst = (hdb::Edge *)(ws + A.L.mst);
    for (int i = tid; i < nocc; i += TB) S.occ[i] = OccW{0u, 0u};
    for (int i = tid; i <= LVL_RING2; i += TB) S.rcnt[i] = A.ring_cnt[i];
    for (int i = tid; i < A.n_ring; i += TB) {
        const uint32_t o = A.ring[i];
        S.ring[i] = make_uint2(o >> 16, (uint32_t)(((int)(o & 255) - 128) * gw + ((int)((o >> 8) & 255) - 128)));
    }
    __syncthreads();
    for (int p = tid; p < N; p += TB) {
        const uint32_t v = pts[p];
        S.rc[p] = (uint16_t)(v & 0xFFFFu);
        S.corei[p] = core_g[p];
        const int cell = ((int)(v & 255) + LVL_PAD) * gw + (int)((v >> 8) & 255) + LVL_PAD;
        atomicOr(&S.occ[cell >> 5].bits, 1u << (cell & 31));
    }
    __syncthreads();
    {
        // points before every word: the thread's words are consecutive, one block scan over the per-thread sums
        const int per = (nocc + TB - 1) / TB, lo = min(nocc, tid * per), hi = min(nocc, lo + per);
        int mine = 0;
        for (int i = lo; i < hi; ++i) mine += __popc(S.occ[i].bits);
        int tot;
        int ex = block_excl_scan(mine, lds16, &tot);
        for (int i = lo; i < hi; ++i) { S.occ[i].base = (uint32_t)ex; ex += __popc(S.occ[i].bits); }
    }
    if (tid == 0) {
        const uint32_t v = S.rc[0];
        S.tnode[0] = make_uint2((v & 255) | ((v >> 8) << 16), S.corei[0]);
        S.corei[0] |= LVL_TREE;
    }
    __syncthreads();
    // chunk c = 64 consecutive points; wavefront w owns chunks w, w + 16, ... (neighbouring chunks -- the ones a rise has
    // work for -- on different SIMDs); lane = point
    uint32_t R[LVL_PT];
#pragma unroll
    for (int q = 0; q < LVL_PT; ++q) {
        R[q] = LVL_RINF;
        const int c = q * NW16 + wave;
        if (c < NF64)
            for (int i = lane; i < LVL_NB / 32; i += 64) S.proc[c * (LVL_NB / 32) + i] = 0u;
    }
    int cnt = 1, done = 0, nb = 0, parity = 0;
    uint32_t m = 0, cur = 0, swept = 0;
    bool need_rise = true;
    unsigned long long fkeep = 0ull;                                // word `lane` of F as the F-empty test read it (maps of <= 4 096 points)
    bool fkeep_ok = false;
    int n_rounds = 0, n_rises = 0;
    long long ph[5] = {0, 0, 0, 0, 0}, tp = wall_clock64();        // phase stamps (10 ns units): rise, extract, probe, accept + commit, mark
    const bool stamps = (A.prim_lvl & 2) != 0;                      // SVC_PRIM_LVL=3: phase stamps (each costs a scalar memory round trip)
#define LVL_PHASE(i) do { if (stamps) { const long long tn_ = wall_clock64(); ph[i] += tn_ - tp; tp = tn_; } } while (0)
    if (tid == 0) hdr[24] = (int)(tp - t0);
    __syncthreads();
    // minimum of R over the block (all wavefronts get it); one barrier
    auto block_min_R = [&]() -> uint32_t {
        uint32_t best = LVL_RINF;
#pragma unroll
        for (int q = 0; q < LVL_PT; ++q) best = min(best, R[q]);
        best = wave_min_u32(best);
        uint32_t *red = S.red + parity * NW16;
        parity ^= 1;
        if (lane == 0) red[wave] = best;
        __syncthreads();
        uint32_t k2 = red[lane & (NW16 - 1)];
        k2 = dpp_min_u32<0x111, 0xF>(k2);
        k2 = dpp_min_u32<0x112, 0xF>(k2);
        k2 = dpp_min_u32<0x114, 0xF>(k2);
        k2 = dpp_min_u32<0x118, 0xF>(k2);
        return (uint32_t)__builtin_amdgcn_readlane((int)k2, 15);
    };
    while (cnt < N) {
        if (need_rise) {
            // ---- rise: the nodes added since the last one become batches; R is caught up block by block (a batch
            // against a chunk) where the block's lower bound allows a value <= the bound; new level, new F
            ++n_rises;
            fkeep_ok = false;                                        // F is rewritten below
            uint32_t cjq[LVL_PT], rcq[LVL_PT];
            uint4 cbq[LVL_PT];
            uint32_t qmask = 0;
#pragma unroll
            for (int q = 0; q < LVL_PT; ++q) {
                cjq[q] = LVL_TREE; rcq[q] = 0;
                const int c = q * NW16 + wave;
                if (c < NF64) {
                    const int p = c * 64 + lane;
                    if (p < N) {
                        cjq[q] = S.corei[p];
                        const uint32_t v = S.rc[p];
                        rcq[q] = (v & 255) | ((v >> 8) << 16);
                    }
                    if (cjq[q] & LVL_TREE) R[q] = LVL_RINF;
                    if (__ballot(!(cjq[q] & LVL_TREE))) {
                        qmask |= 1u << q;
                        // the box and the smallest core distance of the chunk's points that are still OUTSIDE the tree: tighter lower
                        // bounds than the whole chunk's (a jump between blobs otherwise relaxes the finished blob's blocks again)
                        const bool live = !(cjq[q] & LVL_TREE);
                        const uint32_t r_ = rcq[q] & 0xFFFFu, c_ = rcq[q] >> 16;
                        const uint32_t rmin = wave_min_u32(live ? r_ : 255u), rmax = 255u - wave_min_u32(live ? 255u - r_ : 255u);
                        const uint32_t cmin = wave_min_u32(live ? c_ : 255u), cmax = 255u - wave_min_u32(live ? 255u - c_ : 255u);
                        const uint32_t kmin = wave_min_u32(live ? cjq[q] : LVL_RINF);
                        cbq[q] = make_uint4(rmin | (rmax << 8) | (cmin << 16) | (cmax << 24), kmin, 0u, 0u);
                    }
                }
            }
            const int nnew = (cnt - done + 63) >> 6;
            const bool flush = nb + nnew > LVL_NB;                   // table full: settle every pending block, start afresh
            if (flush) {
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q)
                    if (qmask & (1u << q)) {
                        const int c = q * NW16 + wave;
                        R[q] = lvl_sweep_chunk<false>(S, S.btab, S.proc + c * (LVL_NB / 32), nb, 0u, true, cbq[q], rcq[q], cjq[q], R[q]);
                    }
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q) {
                    const int c = q * NW16 + wave;
                    if (c < NF64) for (int i = lane; i < LVL_NB / 32; i += 64) S.proc[c * (LVL_NB / 32) + i] = 0u;
                }
                nb = 0;
                swept = 0;
                __syncthreads();                                       // every wavefront is done with the old table
            }
            for (int g = wave; g < nnew; g += NW16) {
                const int s0 = done + 64 * g, len = min(64, cnt - s0);
                const uint2 tn = S.tnode[s0 + min(lane, len - 1)];
                const uint32_t r = tn.x & 0xFFFFu, cc = tn.x >> 16;
                const uint32_t rmin = wave_min_u32(r), rmax = 255u - wave_min_u32(255u - r);
                const uint32_t cmin = wave_min_u32(cc), cmax = 255u - wave_min_u32(255u - cc);
                const uint32_t kmin = wave_min_u32(tn.y);
                if (lane == 0) S.btab[nb + g] = make_uint4((uint32_t)s0 | ((uint32_t)len << 16), rmin | (rmax << 8) | (cmin << 16) | (cmax << 24), kmin, 0u);
            }
            const bool single = cnt - done == 1;                       // one new node (the start, a jump beyond the ring table)
            if (single) {
                // ... is relaxed against every chunk right away: the bound below is then tight and one sweep is enough
                const uint2 tn = S.tnode[done];
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q)
                    if (qmask & (1u << q)) {
                        const lvl_s2 d = __builtin_bit_cast(lvl_s2, rcq[q]) - __builtin_bit_cast(lvl_s2, tn.x);
                        R[q] = min(R[q], max(max((uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false), cjq[q]), tn.y));
                        const int c = q * NW16 + wave;
                        if (lane == 0) S.proc[c * (LVL_NB / 32) + (nb >> 5)] |= 1u << (nb & 31);
                    }
            }
            nb += nnew;
            done = cnt;
            const uint32_t ub0 = block_min_R();                        // (its barrier also publishes the new batches)
            // swept: every pending block has a lower bound ABOVE it (the limit of the last sweep over all pending blocks)
            const bool need_sweep = !(single && ub0 <= swept);
            if (need_sweep) {
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q)
                    if (qmask & (1u << q)) {
                        const int c = q * NW16 + wave;
                        R[q] = lvl_sweep_chunk<false>(S, S.btab, S.proc + c * (LVL_NB / 32), nb, single ? ub0 : min(ub0, LVL_NEAR), false, cbq[q], rcq[q], cjq[q], R[q]);
                    }
                swept = single ? ub0 : min(ub0, LVL_NEAR);
            }
            if (ub0 > LVL_NEAR && !single) {
                // a jump between blobs: the nearest pending batch of every chunk tightens the bound, then the rest
                const uint32_t ub1 = block_min_R();
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q)
                    if (qmask & (1u << q)) {
                        const int c = q * NW16 + wave;
                        R[q] = lvl_sweep_chunk<true>(S, S.btab, S.proc + c * (LVL_NB / 32), nb, ub1, false, cbq[q], rcq[q], cjq[q], R[q]);
                    }
                const uint32_t ub2 = block_min_R();
#pragma unroll
                for (int q = 0; q < LVL_PT; ++q)
                    if (qmask & (1u << q)) {
                        const int c = q * NW16 + wave;
                        R[q] = lvl_sweep_chunk<false>(S, S.btab, S.proc + c * (LVL_NB / 32), nb, ub2, false, cbq[q], rcq[q], cjq[q], R[q]);
                    }
                swept = max(swept, ub2);
            }
            m = need_sweep ? block_min_R() : ub0;                      // (no sweep: nothing changed since ub0)
#pragma unroll
            for (int q = 0; q < LVL_PT; ++q) {
                const int c = q * NW16 + wave;
                if (c < NF64) {
                    const unsigned long long bal = __ballot(R[q] == m);
                    if (lane == 0) ((unsigned long long *)S.F)[c] = bal;
                }
            }
            __syncthreads();
            need_rise = false;
            LVL_PHASE(0);
        }
        // ---- a round.  Wavefronts 0..3 (one per SIMD) work, lane i = the i-th candidate; the others only meet the
        // barriers and read the outcome (they are needed again at the next rise).
        const bool worker = wave < LVL_WORKERS;
        const bool slow = m > (uint32_t)LVL_RING2;
        const int nk = slow ? 0 : (int)S.rcnt[m];
        const bool fast = nk <= 32 * LVL_WORKERS;                       // every worker's share of the disc fits a 32-bit entrant mask
        int ncand = 0, a = 1;
        uint32_t mycand = LVL_NONE, m2 = LVL_NONE, entmask = 0;
        uint32_t rcv = 0, corev = 0, j4[4] = {0u, 0u, 0u, 0u};         // from the probe: the candidate's own row | col << 8 and core word, the points of its first four cells
        bool dropped = false;
        if (worker) {
            // the first 64 members of F in index order (each worker for itself: no barrier; lvl_extract).  Round 6: a dependent LDS round
            // trip costs this chain 0.2 - 0.4 us, so three of a round's were taken out: F's words come from the registers the F-empty test
            // of the previous round read them into (fkeep: nothing writes F in between unless the level rose), the commit re-uses what the
            // probe read of the candidate, and the entrants of the first four cells are marked from the indices the probe found.
            uint32_t *cw = S.cand + wave * 64;
            ncand = lvl_extract((const unsigned long long *)S.F, NF64, cw, (uint32_t *)(S.slot + wave * 64), mycand, fkeep_ok, fkeep);     // (the slot rows are scratch until the probe writes them)
            LVL_PHASE(1);
            if (!slow) {
                uint32_t dmin, nmin;
                lvl_walk<false>(S, wave, nk, m, lane < ncand, mycand, gw, dmin, nmin, entmask, rcv, corev, j4);
                S.slot[wave * 64 + lane] = make_uint2(dmin, nmin);
            }
        }
        if (!slow) {
            __syncthreads();
            LVL_PHASE(2);
        }
        ++n_rounds;
        if (worker) {
            if (!slow) {
                // ---- accepted prefix (every worker computes the same)
                uint2 sl = S.slot[lane];
#pragma unroll
                for (int w2 = 1; w2 < LVL_WORKERS; ++w2) {
                    const uint2 t = S.slot[w2 * 64 + lane];
                    sl.x = min(sl.x, t.x); sl.y = min(sl.y, t.y);
                }
                if (lane >= ncand) sl = make_uint2(LVL_NONE, LVL_NONE);
                const uint32_t pend = wave_prefix_min_u32(sl.y);
                const uint32_t nextf = (uint32_t)__builtin_amdgcn_update_dpp((int)mycand, (int)mycand, 0x130, 0xF, 0xF, false);   // wave_shl:1 = the next lane's candidate
                const bool stop = lane < ncand && (sl.x != LVL_NONE || (lane + 1 < ncand && pend < nextf));
                const unsigned long long bal = __ballot(stop);
                a = bal ? __builtin_ctzll(bal) + 1 : ncand;
                m2 = (uint32_t)__builtin_amdgcn_readlane((int)sl.x, a - 1);
                dropped = m2 != LVL_NONE;
            }
            // ---- commit: edges, tree membership, F
            const uint32_t prevc = (uint32_t)__builtin_amdgcn_update_dpp((int)mycand, (int)mycand, 0x138, 0xF, 0xF, false);       // wave_shr:1 = the previous lane's
            if (wave == 0 && lane < a) {
                const uint32_t from = lane == 0 ? cur : prevc;
                mst[cnt - 1 + lane] = hdb::Edge{(uint16_t)from, (uint16_t)mycand, m};
                const uint32_t v = slow ? (uint32_t)S.rc[mycand] : rcv;           // (a slow round has no probe)
                const uint32_t cj = slow ? S.corei[mycand] : corev;
                S.tnode[cnt + lane] = make_uint2((v & 255) | ((v >> 8) << 16), cj);
                S.corei[mycand] = cj | LVL_TREE;
                if (!dropped) atomicAnd(&S.F[mycand >> 5], ~(1u << (mycand & 31u)));
            }
            if (!slow && !dropped && fast && lane < a && entmask) {
                // entrants of the accepted candidates: those of the first four cells from the indices the probe kept, the others from the
                // cells it remembered
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((entmask >> u) & 1u) atomicOr(&S.F[j4[u] >> 5], 1u << (j4[u] & 31u));
                const uint32_t v = rcv;
                const uint32_t cell0 = ((v & 255) + LVL_PAD) * (uint32_t)gw + (v >> 8) + LVL_PAD;
                uint32_t em = entmask & ~15u;
                while (em) {
                    const int it = __builtin_ctz(em);
                    em &= em - 1u;
                    const uint32_t cell = cell0 + S.ring[LVL_WORKERS * it + wave].y;
                    const OccW ow = S.occ[cell >> 5];
                    const uint32_t j = ow.base + (uint32_t)__popc(ow.bits & ((1u << (cell & 31u)) - 1u));
                    atomicOr(&S.F[j >> 5], 1u << (j & 31u));
                }
            }
            if (dropped) for (int i = tid; i < 2 * NF64; i += 64 * LVL_WORKERS) S.F[i] = 0u;
            if (wave == 0 && lane == 0) {
                S.ctl[0] = a;
                S.ctl[1] = dropped ? 1 : 0;
                S.ctl[2] = (int)(dropped ? m2 : m);
                S.ctl[3] = (int)(uint32_t)__builtin_amdgcn_readlane((int)mycand, 0);      // (placeholder, overwritten below)
            }
            if (wave == 0) {
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)mycand, a - 1);
                if (lane == 0) S.ctl[3] = (int)last;
            }
        }
        __syncthreads();
        LVL_PHASE(3);
        a = S.ctl[0];
        dropped = S.ctl[1] != 0;
        const uint32_t mnew = (uint32_t)S.ctl[2];
        cur = (uint32_t)S.ctl[3];
        cnt += a;
        if (cnt >= N) break;
        if (slow) { need_rise = true; continue; }                       // beyond the ring table: one node, then a rise
        if (dropped || !fast) {
            // a drop (F is rebuilt from the last node's disc at the new, lower level) or a disc too large for the
            // entrant masks: a second walk, now that this round's nodes carry their tree bit
            if (worker) {
                const bool on = dropped ? lane == a - 1 : lane < a;
                uint32_t d_, n_, e_;
                uint32_t r_, c_, j_[4];
                lvl_walk<true>(S, wave, (int)S.rcnt[mnew], mnew, on, mycand, gw, d_, n_, e_, r_, c_, j_);
            }
            m = mnew;
            __syncthreads();
            LVL_PHASE(4);
        }
        // F empty -> the level rises (every wavefront reads the same words).  Maps of up to 4 096 points: the words stay in registers for the
        // next round's extraction (no write to F lies between this read and that one unless the level rises, which clears fkeep_ok).
        {
            bool any = false;
            if (NF64 <= 64) {
                fkeep = lane < NF64 ? ((const unsigned long long *)S.F)[lane] : 0ull;
                any = __ballot(fkeep != 0ull) != 0ull;
                fkeep_ok = true;
            } else {
                for (int wb = 0; wb < NF64; wb += 64) {
                    const int k = wb + lane;
                    any = any || (__ballot(k < NF64 && ((const unsigned long long *)S.F)[k] != 0ull) != 0ull);
                }
            }
            need_rise = !any;
        }
    }
    if (tid == 0) {
        hdr[12] = (int)(wall_clock64() - t0); hdr[16] = n_rounds; hdr[17] = n_rises;
        for (int i = 0; i < 5; ++i) hdr[18 + i] = (int)ph[i];
    }
}
__global__ __launch_bounds__(TB) void k_prim_lvl(TailArgs A) { prim_lvl_body(A); }

// --------------------------------------------------------------------------------------
// k_prim_lvl_big: the same rounds for maps of MORE than LVL_CAP points (up to the whole 255 x 255 grid).  Per-point state
// does not fit in LDS there (35 000 points x (tree node 8 + core 4 + row/col 2 + reach 4 + block flags 0.5) B = 650 KB),
// so it lives in the frame's workspace -- L2-resident, read with gathers -- and LDS keeps what every probe touches: the
// occupancy grid, F, the ring table, the batch table, one box + minimum per chunk of 64 points.  The reach R of a chunk is
// loaded, relaxed and stored per sweep instead of sitting in registers; a batch's 64 tree nodes are staged into a per-wave
// LDS slab (one coalesced load) before they are broadcast to the lanes.  Same (last node, new node, weight) sequence as
// k_prim_lvl and the library (tests: test_tail_maximum_size_all_pixels_set, the 288-map comparison with the one-node-per-step
// kernels, tools/soak_tail.py).  Replaces the one-node-per-step k_prim_big (3 171 spilled VGPRs; 12 ms per map at N = 10 k,
// 3.3 s at 25 k: profiles/r04_tail_vs_N.txt) on the default path.
// --------------------------------------------------------------------------------------
#define LVL_NBB 576                    // batches of tree nodes a big map can have pending: 36 864 nodes >= a whole 140 x 250 map (18 flag words per chunk)
static size_t lvl_big_lds_bytes(int h, int w, int n_ring) {
    const size_t cap = (size_t)h * w;
    const size_t cells = (size_t)(h + 2 * LVL_PAD) * (w + 2 * LVL_PAD);
    const size_t nf = (cap + 63) / 64;
    auto up = [](size_t b) { return (b + 15) / 16 * 16; };
    return up((cells + 31) / 32 * sizeof(OccW)) + up(nf * 8) + up((size_t)n_ring * 8) + up(64 * LVL_WORKERS * 4) + up(64 * LVL_WORKERS * 8) +
           up(2 * NW16 * 4) + up(16) + up(LVL_NBB * 16) + up((LVL_RING2 + 1) * 2) + up(nf * 8) + up(nf * 4) + up(NW16 * 64 * 8) +
           up(nf * (LVL_NBB / 32) * 4) + 64;
}

// lvl_sweep_chunk with the batch staged through the wave's LDS slab (tnode is in global memory here)
template <bool NEAREST>
__device__ __forceinline__ uint32_t lvl_sweep_chunk_big(const LvlLds &S, uint2 *stage, const uint4 *btab, uint32_t *proc, int nb, uint32_t limit,
                                                        bool all, uint2 cb, uint32_t rcv, uint32_t cj, uint32_t r) {
    const int lane = threadIdx.x & 63;
    LvlLds T = S;
    T.tnode = stage;
    auto relax = [&](int b0) {
        const uint32_t sl = btab[b0].x;
        const int start = (int)(sl & 0xFFFFu), len = (int)(sl >> 16);
        __builtin_amdgcn_wave_barrier();                                // the slab's previous readers are done
        stage[lane] = S.tnode[start + min(lane, len - 1)];
        __builtin_amdgcn_wave_barrier();
        r = lvl_relax_block(T, 0, len, rcv, cj, r);
    };
    uint32_t bestkey = LVL_NONE;
    for (int bb = 0; bb < nb; bb += 64) {
        const int b = bb + lane;
        uint32_t lb = LVL_NONE;
        if (b < nb && !((proc[b >> 5] >> (b & 31)) & 1u)) {
            const uint4 e = btab[b];
            lb = all ? 0u : max(max(lvl_box_d2(e.y, cb.x), e.z), cb.y);
        }
        if (NEAREST) {
            if (lb <= limit) bestkey = min(bestkey, (min(lb, 0x3FFFFFu) << 10) | (uint32_t)b);
            continue;
        }
        unsigned long long todo = __ballot(lb <= limit && lb != LVL_NONE);
        if (lane == 0 && todo) {
            proc[bb >> 5] |= (uint32_t)todo;
            if (bb + 32 < LVL_NBB) proc[(bb >> 5) + 1] |= (uint32_t)(todo >> 32);
        }
        while (todo) {
            const int b0 = bb + __builtin_ctzll(todo);
            todo &= todo - 1ull;
            relax(b0);
        }
    }
    if (NEAREST) {
        bestkey = wave_min_u32(bestkey);
        if (bestkey != LVL_NONE) {
            const int b0 = (int)(bestkey & 1023u);
            if (lane == 0) proc[b0 >> 5] |= 1u << (b0 & 31);
            relax(b0);
        }
    }
    return r;
}

__device__ __forceinline__ void prim_lvl_big_body(const TailArgs &A) {
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (!hdr[3]) return;
    const int N = hdr[0];
    if (N <= LVL_CAP) return;                                         // k_prim_lvl (inside k_tail_front) has done it
    extern __shared__ uint8_t sm_lvl[];
    __shared__ int lds16[NW16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gw = A.w + 2 * LVL_PAD, gcells = (A.h + 2 * LVL_PAD) * gw, nocc = (gcells + 31) >> 5;
    const int NF64 = (N + 63) >> 6;
    const int capf = (A.h * A.w + 63) >> 6;
    LvlLds S;
    uint2 *cbx;                 // per chunk: box of its points outside the tree, their smallest core distance (LVL_NONE: none left)
    uint32_t *cmin;             // per chunk: minimum of R
    uint2 *stage;               // [NW16][64] a batch's tree nodes, per wavefront
    {
        uint8_t *p = sm_lvl;
        S.occ = carve<OccW>(p, nocc);
        S.F = carve<uint32_t>(p, (size_t)capf * 2);
        S.ring = carve<uint2>(p, A.n_ring);
        S.cand = carve<uint32_t>(p, 64 * LVL_WORKERS);
        S.slot = carve<uint2>(p, 64 * LVL_WORKERS);
        S.red = carve<uint32_t>(p, 2 * NW16);
        S.ctl = carve<int>(p, 4);
        S.btab = carve<uint4>(p, LVL_NBB);
        S.rcnt = carve<uint16_t>(p, LVL_RING2 + 1);
        cbx = carve<uint2>(p, capf);
        cmin = carve<uint32_t>(p, capf);
        stage = carve<uint2>(p, NW16 * 64);
        S.proc = carve<uint32_t>(p, (size_t)capf * (LVL_NBB / 32));     // [chunks][LVL_NBB / 32] batch already relaxed against the chunk
        S.gw = gw;
    }
    // per-point state in the frame's workspace (regions of stages that run later or not at all for this map)
    S.tnode = (uint2 *)(ws + A.L.ea);                                 // k_sort's ping-pong buffer
    S.corei = (uint32_t *)(ws + A.L.reach);
    S.rc = (uint16_t *)(ws + A.L.sp);
    uint32_t *Rg = (uint32_t *)(ws + A.L.absw);
    uint2 *mystage = stage + wave * 64;
    const long long t0 = wall_clock64();
    const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
    const uint32_t *core_g = (const uint32_t *)(ws + A.L.core);
    hdb::Edge *mst = (hdb::Edge *)(ws + A.L.mst);
    for (int i = tid; i < nocc; i += TB) S.occ[i] = OccW{0u, 0u};
    for (int i = tid; i <= LVL_RING2; i += TB) S.rcnt[i] = A.ring_cnt[i];
    for (int i = tid; i < A.n_ring; i += TB) {
        const uint32_t o = A.ring[i];
        S.ring[i] = make_uint2(o >> 16, (uint32_t)(((int)(o & 255) - 128) * gw + ((int)((o >> 8) & 255) - 128)));
    }
    __syncthreads();
    for (int p = tid; p < N; p += TB) {
        const uint32_t v = pts[p];
        S.rc[p] = (uint16_t)(v & 0xFFFFu);
        S.corei[p] = core_g[p];
        Rg[p] = LVL_RINF;
        const int cell = ((int)(v & 255) + LVL_PAD) * gw + (int)((v >> 8) & 255) + LVL_PAD;
        atomicOr(&S.occ[cell >> 5].bits, 1u << (cell & 31));
    }
    for (int c = tid; c < NF64; c += TB) cmin[c] = LVL_RINF;
    for (int i = tid; i < NF64 * (LVL_NBB / 32); i += TB) S.proc[i] = 0u;
    __syncthreads();
    {
        const int per = (nocc + TB - 1) / TB, lo = min(nocc, tid * per), hi = min(nocc, lo + per);
        int mine = 0;
        for (int i = lo; i < hi; ++i) mine += __popc(S.occ[i].bits);
        int tot;
        int ex = block_excl_scan(mine, lds16, &tot);
        for (int i = lo; i < hi; ++i) { S.occ[i].base = (uint32_t)ex; ex += __popc(S.occ[i].bits); }
    }
    if (tid == 0) {
        const uint32_t v = S.rc[0];
        S.tnode[0] = make_uint2((v & 255) | ((v >> 8) << 16), S.corei[0]);
        S.corei[0] |= LVL_TREE;
    }
    __syncthreads();
    int cnt = 1, done = 0, nb = 0, parity = 0;
    uint32_t m = 0, cur = 0, swept = 0;
    bool need_rise = true;
    int n_rounds = 0, n_rises = 0;
    if (tid == 0) hdr[24] = (int)(wall_clock64() - t0);
    // minimum of cmin over the block (all wavefronts get it); one barrier
    auto block_min_R = [&]() -> uint32_t {
        uint32_t best = LVL_RINF;
        for (int c = wave + lane * NW16; c < NF64; c += 64 * NW16) best = min(best, cmin[c]);      // this wavefront's own chunks
        best = wave_min_u32(best);
        uint32_t *red = S.red + parity * NW16;
        parity ^= 1;
        if (lane == 0) red[wave] = best;
        __syncthreads();
        uint32_t k2 = red[lane & (NW16 - 1)];
        k2 = dpp_min_u32<0x111, 0xF>(k2);
        k2 = dpp_min_u32<0x112, 0xF>(k2);
        k2 = dpp_min_u32<0x114, 0xF>(k2);
        k2 = dpp_min_u32<0x118, 0xF>(k2);
        return (uint32_t)__builtin_amdgcn_readlane((int)k2, 15);
    };
    // one sweep over this wavefront's chunks that still hold points outside the tree: R is loaded, relaxed, stored
    auto sweep_all = [&](int mode, uint32_t limit, bool all) {      // mode 0: every batch within the limit, 1: the nearest one
        for (int c = wave; c < NF64; c += NW16) {
            const uint2 cb = cbx[c];
            if (cb.y == LVL_NONE) continue;
            const int p = c * 64 + lane;
            uint32_t cj = LVL_TREE, rcv = 0, r = LVL_RINF;
            if (p < N) {
                cj = S.corei[p];
                const uint32_t v = S.rc[p];
                rcv = (v & 255) | ((v >> 8) << 16);
                r = Rg[p];
            }
            uint32_t *proc = S.proc + c * (LVL_NBB / 32);
            const uint32_t r0 = r;
            if (mode == 0) r = lvl_sweep_chunk_big<false>(S, mystage, S.btab, proc, nb, limit, all, cb, rcv, cj, r);
            else r = lvl_sweep_chunk_big<true>(S, mystage, S.btab, proc, nb, limit, all, cb, rcv, cj, r);
            if (cj & LVL_TREE) r = LVL_RINF;
            if (p < N && r != r0) Rg[p] = r;
            const uint32_t mn = wave_min_u32(r);
            if (lane == 0) cmin[c] = mn;
        }
    };
    while (cnt < N) {
        if (need_rise) {
            ++n_rises;
            // per chunk: tree members get an infinite reach; box and smallest core distance of the points still outside
            for (int c = wave; c < NF64; c += NW16) {
                const int p = c * 64 + lane;
                uint32_t cj = LVL_TREE, rcv = 0;
                if (p < N) {
                    cj = S.corei[p];
                    const uint32_t v = S.rc[p];
                    rcv = (v & 255) | ((v >> 8) << 16);
                }
                const bool live = !(cj & LVL_TREE);
                uint32_t r = LVL_RINF;
                if (p < N) {
                    r = live ? Rg[p] : LVL_RINF;
                    if (!live) Rg[p] = LVL_RINF;
                }
                const uint32_t mn = wave_min_u32(r);
                uint2 cb = make_uint2(0u, LVL_NONE);
                if (__ballot(live)) {
                    const uint32_t r_ = rcv & 0xFFFFu, c_ = rcv >> 16;
                    const uint32_t rmin = wave_min_u32(live ? r_ : 255u), rmax = 255u - wave_min_u32(live ? 255u - r_ : 255u);
                    const uint32_t cmn = wave_min_u32(live ? c_ : 255u), cmx = 255u - wave_min_u32(live ? 255u - c_ : 255u);
                    const uint32_t kmin = wave_min_u32(live ? cj : LVL_RINF);
                    cb = make_uint2(rmin | (rmax << 8) | (cmn << 16) | (cmx << 24), kmin);
                }
                if (lane == 0) { cbx[c] = cb; cmin[c] = mn; }
            }
            __builtin_amdgcn_wave_barrier();
            // batches of <= 64 of the nodes [first, cnt) -> table entries from position `at` on
            auto add_batches = [&](int first, int nbat, int at) {
                for (int g = wave; g < nbat; g += NW16) {
                    const int s0 = first + 64 * g, len = min(64, cnt - s0);
                    const uint2 tn = S.tnode[s0 + min(lane, len - 1)];
                    const uint32_t r = tn.x & 0xFFFFu, cc = tn.x >> 16;
                    const uint32_t rmin = wave_min_u32(r), rmax = 255u - wave_min_u32(255u - r);
                    const uint32_t cmn = wave_min_u32(cc), cmx = 255u - wave_min_u32(255u - cc);
                    const uint32_t kmin = wave_min_u32(tn.y);
                    if (lane == 0) S.btab[at + g] = make_uint4((uint32_t)s0 | ((uint32_t)len << 16), rmin | (rmax << 8) | (cmn << 16) | (cmx << 24), kmin, 0u);
                }
            };
            int nnew = (cnt - done + 63) >> 6;
            while (nb + nnew > LVL_NBB) {
                // table full: settle every pending block, start afresh (N^2 work: the table is sized so that a 140 x 250 map never
                // fills it).  More new nodes than the table holds (maps beyond 36 864 points) are entered a table at a time
                if (nb) {
                    sweep_all(0, 0u, true);
                    for (int c = wave; c < NF64; c += NW16)
                        for (int i = lane; i < LVL_NBB / 32; i += 64) S.proc[c * (LVL_NBB / 32) + i] = 0u;
                    nb = 0;
                    swept = 0;
                    __syncthreads();                                   // every wavefront is done with the old table
                }
                if (nnew > LVL_NBB) {
                    add_batches(done, LVL_NBB, 0);
                    nb = LVL_NBB;
                    done += 64 * LVL_NBB;
                    nnew -= LVL_NBB;
                    __syncthreads();                                   // the table is published
                }
            }
            add_batches(done, nnew, nb);
            const bool single = cnt - done == 1;                       // one new node (the start, a jump beyond the ring table)
            if (single) {
                const uint2 tn = S.tnode[done];
                for (int c = wave; c < NF64; c += NW16) {
                    if (cbx[c].y == LVL_NONE) continue;
                    const int p = c * 64 + lane;
                    uint32_t r = LVL_RINF;
                    if (p < N) {
                        const uint32_t cj = S.corei[p];
                        if (!(cj & LVL_TREE)) {
                            const uint32_t v = S.rc[p];
                            const uint32_t rcv = (v & 255) | ((v >> 8) << 16);
                            const lvl_s2 d = __builtin_bit_cast(lvl_s2, rcv) - __builtin_bit_cast(lvl_s2, tn.x);
                            const uint32_t r0 = Rg[p];
                            r = min(r0, max(max((uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false), cj), tn.y));
                            if (r != r0) Rg[p] = r;
                        }
                    }
                    const uint32_t mn = wave_min_u32(r);
                    if (lane == 0) {
                        cmin[c] = mn;
                        S.proc[c * (LVL_NBB / 32) + (nb >> 5)] |= 1u << (nb & 31);
                    }
                }
            }
            nb += nnew;
            done = cnt;
            const uint32_t ub0 = block_min_R();                        // (its barrier also publishes the new batches)
            const bool need_sweep = !(single && ub0 <= swept);
            if (need_sweep) {
                sweep_all(0, single ? ub0 : min(ub0, LVL_NEAR), false);
                swept = single ? ub0 : min(ub0, LVL_NEAR);
            }
            if (ub0 > LVL_NEAR && !single) {
                const uint32_t ub1 = block_min_R();
                sweep_all(1, ub1, false);
                const uint32_t ub2 = block_min_R();
                sweep_all(0, ub2, false);
                swept = max(swept, ub2);
            }
            m = need_sweep ? block_min_R() : ub0;
            for (int c = wave; c < NF64; c += NW16) {
                unsigned long long bal = 0ull;
                if (cmin[c] == m) {
                    const int p = c * 64 + lane;
                    bal = __ballot(p < N && Rg[p] == m);
                }
                if (lane == 0) ((unsigned long long *)S.F)[c] = bal;
            }
            __syncthreads();
            need_rise = false;
        }
        // ---- a round (as in prim_lvl_body)
        const bool worker = wave < LVL_WORKERS;
        const bool slow = m > (uint32_t)LVL_RING2;
        const int nk = slow ? 0 : (int)S.rcnt[m];
        const bool fast = nk <= 32 * LVL_WORKERS;
        int ncand = 0, a = 1;
        uint32_t mycand = LVL_NONE, m2 = LVL_NONE, entmask = 0;
        bool dropped = false;
        if (worker) {
            uint32_t *cw = S.cand + wave * 64;
            ncand = lvl_extract((const unsigned long long *)S.F, NF64, cw, (uint32_t *)(S.slot + wave * 64), mycand);     // (the slot rows are scratch until the probe writes them)
            if (!slow) {
                uint32_t dmin, nmin;
                uint32_t r_, c_, j_[4];
                lvl_walk<false>(S, wave, nk, m, lane < ncand, mycand, gw, dmin, nmin, entmask, r_, c_, j_);
                S.slot[wave * 64 + lane] = make_uint2(dmin, nmin);
            }
        }
        if (!slow) __syncthreads();
        ++n_rounds;
        if (worker) {
            if (!slow) {
                uint2 sl = S.slot[lane];
#pragma unroll
                for (int w2 = 1; w2 < LVL_WORKERS; ++w2) {
                    const uint2 t = S.slot[w2 * 64 + lane];
                    sl.x = min(sl.x, t.x); sl.y = min(sl.y, t.y);
                }
                if (lane >= ncand) sl = make_uint2(LVL_NONE, LVL_NONE);
                const uint32_t pend = wave_prefix_min_u32(sl.y);
                const uint32_t nextf = (uint32_t)__builtin_amdgcn_update_dpp((int)mycand, (int)mycand, 0x130, 0xF, 0xF, false);
                const bool stop = lane < ncand && (sl.x != LVL_NONE || (lane + 1 < ncand && pend < nextf));
                const unsigned long long bal = __ballot(stop);
                a = bal ? __builtin_ctzll(bal) + 1 : ncand;
                m2 = (uint32_t)__builtin_amdgcn_readlane((int)sl.x, a - 1);
                dropped = m2 != LVL_NONE;
            }
            const uint32_t prevc = (uint32_t)__builtin_amdgcn_update_dpp((int)mycand, (int)mycand, 0x138, 0xF, 0xF, false);
            if (wave == 0 && lane < a) {
                const uint32_t from = lane == 0 ? cur : prevc;
                mst[cnt - 1 + lane] = hdb::Edge{(uint16_t)from, (uint16_t)mycand, m};
                const uint32_t v = S.rc[mycand];
                const uint32_t cj = S.corei[mycand];
                S.tnode[cnt + lane] = make_uint2((v & 255) | ((v >> 8) << 16), cj);
                S.corei[mycand] = cj | LVL_TREE;
                if (!dropped) atomicAnd(&S.F[mycand >> 5], ~(1u << (mycand & 31u)));
            }
            if (!slow && !dropped && fast && lane < a && entmask) {
                const uint32_t v = S.rc[mycand];
                const uint32_t cell0 = ((v & 255) + LVL_PAD) * (uint32_t)gw + (v >> 8) + LVL_PAD;
                uint32_t em = entmask;
                while (em) {
                    const int it = __builtin_ctz(em);
                    em &= em - 1u;
                    const uint32_t cell = cell0 + S.ring[LVL_WORKERS * it + wave].y;
                    const OccW ow = S.occ[cell >> 5];
                    const uint32_t j = ow.base + (uint32_t)__popc(ow.bits & ((1u << (cell & 31u)) - 1u));
                    atomicOr(&S.F[j >> 5], 1u << (j & 31u));
                }
            }
            if (dropped) for (int i = tid; i < 2 * NF64; i += 64 * LVL_WORKERS) S.F[i] = 0u;
            if (wave == 0) {
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)mycand, a - 1);
                if (lane == 0) {
                    S.ctl[0] = a;
                    S.ctl[1] = dropped ? 1 : 0;
                    S.ctl[2] = (int)(dropped ? m2 : m);
                    S.ctl[3] = (int)last;
                }
            }
        }
        __syncthreads();
        a = S.ctl[0];
        dropped = S.ctl[1] != 0;
        const uint32_t mnew = (uint32_t)S.ctl[2];
        cur = (uint32_t)S.ctl[3];
        cnt += a;
        if (cnt >= N) break;
        if (slow) { need_rise = true; continue; }
        if (dropped || !fast) {
            if (worker) {
                const bool on = dropped ? lane == a - 1 : lane < a;
                uint32_t d_, n_, e_;
                uint32_t r_, c_, j_[4];
                lvl_walk<true>(S, wave, (int)S.rcnt[mnew], mnew, on, mycand, gw, d_, n_, e_, r_, c_, j_);
            }
            m = mnew;
            __syncthreads();
        }
        {
            bool any = false;
            for (int wb = 0; wb < NF64; wb += 64) {
                const int k = wb + lane;
                any = any || (__ballot(k < NF64 && ((const unsigned long long *)S.F)[k] != 0ull) != 0ull);
            }
            need_rise = !any;
        }
    }
    if (tid == 0) { hdr[12] = (int)(wall_clock64() - t0); hdr[16] = n_rounds; hdr[17] = n_rises; }
}
__global__ __launch_bounds__(TB) void k_prim_lvl_big(TailArgs A) { prim_lvl_big_body(A); }

// One 1x5 (ROWS) or 5x1 pass of the separable grey CLOSE over a map in LDS: MAX = dilate, else erode; samples
// outside the image are ignored (OpenCV's morphology border).  A thread produces four consecutive outputs
// along the pass direction from one sliding window of eight samples (2 LDS reads per output instead of 5).
// The pass runs over a WINDOW of the image (rows r0 .. r0+wh, columns c0 .. c0+ww; row stride w): samples outside the window
// count as outside the image.  finish_body passes the kept cluster's bounding box widened by 4 pixels (clipped to the
// image): everything further out is zero before and after the CLOSE, and inside the window the results are those of the
// whole-image pass -- a dilated value is non-zero within 2 pixels of the box only, so an eroded value next to an inner
// window edge already has the zeros of the window's own last two lines among its samples.
template <bool MAX, bool ROWS>
__device__ __forceinline__ void close_pass(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int r0, int c0, int wh, int ww) {
    const int len = ROWS ? ww : wh, lines = ROWS ? wh : ww;  // length of a line along the pass, number of lines
    const int stride = ROWS ? 1 : w, lstride = ROWS ? w : 1;
    const int groups = (len + 3) >> 2;
    const int ident = MAX ? 0 : 255;
    const int origin = r0 * w + c0;
    for (int it = threadIdx.x; it < lines * groups; it += TB) {
        // consecutive threads take consecutive positions ACROSS lines for the column pass (adjacent bytes in LDS)
        const int line = ROWS ? it / groups : it % lines, g = ROWS ? it - line * groups : it / lines;
        const int p0 = 4 * g;
        const uint8_t *sp = src + origin + line * lstride;
        int win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = p0 - 2 + j;
            win[j] = (unsigned)q < (unsigned)len ? (int)sp[q * stride] : ident;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (p0 + j >= len) break;
            int v = win[j];
#pragma unroll
            for (int d = 1; d < 5; ++d) v = MAX ? max(v, win[j + d]) : min(v, win[j + d]);
            dst[origin + line * lstride + (p0 + j) * stride] = (uint8_t)v;
        }
    }
}

// --------------------------------------------------------------------------------------
// k_tree (the serial builder, kept for maps k_tree_par does not take): hierarchy + EOM (hdb_tree.h on one wavefront; its
// arrays live in LDS when N <= TREE_LDS_CAP), labels, cluster weights.  k_finish: zeroing, CLOSE 5x5, centroid.
// --------------------------------------------------------------------------------------
#define TREE_LDS_CAP 4352      // points: 28 B/point of hierarchy state in LDS
#define TREE_LDS_CLUSTERS 512  // condensed clusters kept in LDS (57 B each)
#define FIN_LDS_BYTES (TREE_LDS_CAP * 28 + TREE_LDS_CLUSTERS * 60)

struct TreeShared {
    int nsel, best, ok;
};

// hdb::build_batched<64> on one wavefront (see the comment there): lane j resolves edge j of the batch and
// classifies its sides (big / fresh / linked to an earlier lane); six rounds of pointer jumping along the links
// give every lane the edge that started its tree (a cluster's absorption chain or a small tree of unions); the
// longest prefix of attachments is applied by all lanes at once, and the edge that ends it is resolved afresh
// and applied by lane 0 through hdb::merge.
__device__ __forceinline__ bool build_wave(hdb::Tree &t, const hdb::Edge *edges, int n, int mcs) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    t.n = n;
    t.nclusters = 0;
    int i0 = 0;
    while (i0 < n - 1) {
        const int m = min(64, n - 1 - i0);
        hdb::Edge e = hdb::Edge{0, 0, 1};
        hdb::Resolved pre = hdb::Resolved{0, 0, hdb::NONE16, hdb::NONE16, 0, 0, 0, 0};
        const bool in = lane < m;
        if (in) {
            e = edges[i0 + lane];
            pre = hdb::resolve_ro(t, e);
            // the walk to the top cluster is not compressed by resolve_ro; with numpy's edge order a map has dozens of
            // clusters whose union-find chains grow with every split, so shorten them here: the cluster that absorbed
            // a root now points straight to its top (lanes race with equal or equally valid ancestors)
            if (pre.ca != hdb::NONE16) { const uint16_t c0 = t.absc[pre.ra]; if (c0 != pre.ca) t.cup[c0] = (uint16_t)pre.ca; }
            if (pre.cb != hdb::NONE16) { const uint16_t c0 = t.absc[pre.rb]; if (c0 != pre.cb) t.cup[c0] = (uint16_t)pre.cb; }
        }
        // First lane of the batch that contains each small root: the table is absw[] (unused for un-absorbed roots),
        // a scatter-min of lane numbers, read back with atomic loads (the minimum is formed in L2 / LDS, not in this
        // CU's vector L1), then restored.
        const bool a_small = in && pre.ca == hdb::NONE16, b_small = in && pre.cb == hdb::NONE16;
        if (a_small) atomicMin(&t.absw[pre.ra], (uint32_t)lane);
        if (b_small) atomicMin(&t.absw[pre.rb], (uint32_t)lane);
        uint32_t fa = (uint32_t)lane, fb = (uint32_t)lane;
        if (a_small) fa = __hip_atomic_load(&t.absw[pre.ra], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b_small) fb = __hip_atomic_load(&t.absw[pre.rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a_small) __hip_atomic_store(&t.absw[pre.ra], hdb::NO_LANE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b_small) __hip_atomic_store(&t.absw[pre.rb], hdb::NO_LANE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // sides: B big at the start, F fresh small root, L small root that an earlier lane contains
        const bool aB = in && !a_small, bB = in && !b_small;
        const bool aF = a_small && fa == (uint32_t)lane, bF = b_small && fb == (uint32_t)lane;
        uint32_t par = (uint32_t)lane, clus = hdb::NONE16, add = 0;
        bool bad = false;
        if (aF && bF) add = pre.sa + pre.sb;                          // starts a small tree
        else if (aF && bB) { clus = pre.cb; add = pre.sa; }           // starts / continues the chain of cluster cb
        else if (aB && bF) { clus = pre.ca; add = pre.sb; }
        else if (aF && b_small) { par = fb; add = pre.sa; }           // F + L: attaches to the tree of lane fb
        else if (bF && a_small) { par = fa; add = pre.sb; }
        else bad = in;                                                // two existing things meet: ends the prefix
        // the lane that started this lane's tree: links point to earlier lanes, six rounds of pointer jumping
        uint32_t root = par;
        for (int round = 0; round < 6; ++round) {         // links point to earlier lanes: at most six rounds, usually one or two
            const uint32_t up = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(root << 2), (int)root);
            if (!__ballot(up != root)) break;
            root = up;
        }
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(root << 2), (int)clus);     // cluster of the chain, NONE16: a small tree
        const bool small_lane = in && !bad && c == hdb::NONE16;
        // small trees: running size (a tree gives birth to a cluster where it reaches min_cluster_size: ends the
        // prefix), previous lane of the tree, lanes of the tree.  Trees of one lane need no loop.
        uint32_t run = add;
        unsigned long long mygrp = 1ull << lane;
        unsigned long long todo = __ballot(small_lane && root != (uint32_t)lane);
        while (todo) {                                   // once per small tree with attachments
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)root, __builtin_ctzll(todo));
            const bool mine = small_lane && root == r0;
            const unsigned long long grp = __ballot(mine);
            const uint32_t sc = wave_incl_scan_u32(mine ? add : 0u);
            if (mine) { run = sc; mygrp = grp; }
            todo &= ~grp;
        }
        const unsigned long long stop = __ballot(bad || (small_lane && (int)run >= mcs));
        const int P = stop ? min(m, (int)__builtin_ctzll(stop)) : m;
        if (P > 0) {
            const bool act = lane < P;
            const uint32_t node = (uint32_t)(n + i0 + lane);
            // absorptions: grouped by cluster (several chains of one batch may feed the same cluster)
            {
                const bool actc = act && c != hdb::NONE16;
                const bool abig = !aF;                   // the non-fresh side is the cluster's side
                const uint32_t r = abig ? pre.rb : pre.ra, s = abig ? pre.sb : pre.sa, ns = abig ? pre.nb : pre.na;
                uint32_t prev = 0;
                unsigned long long todoc = __ballot(actc);
                while (todoc) {                          // once per distinct cluster (nearly always one)
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, __builtin_ctzll(todoc));
                    const unsigned long long grp = __ballot(actc && c == c0);
                    const uint32_t before = t.cdn[c0];
                    if (actc && c == c0) {
                        const unsigned long long lower = grp & below;
                        prev = lower ? (uint32_t)(n + i0 + 63 - __builtin_clzll(lower)) : before;
                    }
                    if (lane == 0) t.cdn[c0] = (uint32_t)(n + i0 + 63 - __builtin_clzll(grp));
                    todoc &= ~grp;
                }
                if (actc) {
                    t.dparent[abig ? prev : ns] = node << 1;
                    t.dparent[abig ? ns : prev] = (node << 1) | 1u;
                    t.absc[r] = (uint16_t)c; t.absw[r] = e.w;
                    t.evc[i0 + lane] = (uint16_t)c; t.evs[i0 + lane] = (uint16_t)s;
                    atomicAdd(&t.csize[c], s);
                }
            }
            // unions inside small trees; the tree's union-find root is its starter's a side
            {
                const bool acts = act && c == hdb::NONE16;
                const uint32_t R = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(root << 2), (int)pre.ra);
                const unsigned long long grpP = mygrp & (P == 64 ? ~0ull : ((1ull << P) - 1ull));
                const int last = grpP ? 63 - __builtin_clzll(grpP) : lane;
                if (acts) {
                    t.evc[i0 + lane] = (uint16_t)hdb::NONE16; t.evs[i0 + lane] = 0;
                    if (root == (uint32_t)lane) {
                        t.dparent[pre.na] = node << 1;
                        t.dparent[pre.nb] = (node << 1) | 1u;
                        t.sp[pre.rb] = (uint16_t)R;
                    } else {
                        const unsigned long long lower = mygrp & below;
                        const uint32_t pn = (uint32_t)(n + i0 + 63 - __builtin_clzll(lower));      // the starter is always below
                        const uint32_t rf = aF ? pre.ra : pre.rb, nf = aF ? pre.na : pre.nb;
                        t.dparent[aF ? nf : pn] = node << 1;            // the a side is the left child
                        t.dparent[aF ? pn : nf] = (node << 1) | 1u;
                        t.sp[rf] = (uint16_t)R;
                    }
                    if (lane == last) { t.ssz[R] = (uint16_t)run; t.sdn[R] = node; }
                }
            }
            i0 += P;
        }
        if (P < m) {                                     // the roots may have moved (unions): resolve afresh
            int ok = 1;
            if (lane == 0) {
                const hdb::Edge ep = edges[i0];
                ok = hdb::merge(t, i0, n, mcs, ep.w, hdb::resolve(t, ep)) ? 1 : 0;
            }
            if (!__builtin_amdgcn_readfirstlane(ok)) return false;
            ++i0;
        }
    }
    if (lane == 0) t.dparent[2 * n - 2] = 0xFFFFFFFFu;
    t.nclusters = __builtin_amdgcn_readfirstlane(t.nclusters);
    return true;
}

__device__ __forceinline__ double readlane_f64(double v, int j) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), j);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// hdb::accumulate on one wavefront: the reciprocals and the per-row terms of 64 rows are computed
// by the lanes in parallel; only the float64 additions run in the library's order (one after the
// other, into a register that is written back when the cluster changes), so the sums are bit-identical.
__device__ __forceinline__ void accumulate_wave(hdb::Tree &t, const hdb::Edge *edges) {
    const int lane = threadIdx.x & 63;
    uint32_t cur = hdb::NONE16;
    double acc = 0.0;
    for (int top = t.n - 2; top >= 0; top -= 64) {
        const int i = top - lane;
        uint32_t c = hdb::NONE16, s = 0;
        double term = 0.0, ta = 0.0, tb = 0.0;
        if (i >= 0) {
            c = t.evc[i];
            if (c != hdb::NONE16) {
                s = t.evs[i];
                const uint32_t bw = t.cbirthw[c];
                const double lam = 1.0 / (double)edges[i].w;
                const double birth = bw ? 1.0 / (double)bw : 0.0;
                term = (lam - birth) * 1.0;
                if (s == 0) {
                    ta = (lam - birth) * (double)t.cspa[c];
                    tb = (lam - birth) * (double)t.cspb[c];
                }
            }
        }
        unsigned long long live = __ballot(c != hdb::NONE16);
        if (!live) continue;
        // the common batch -- every row a single point falling out of one and the same cluster -- as 64 additions
        // with compile-time lane numbers (rows without an event add +0.0, which changes nothing)
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, __builtin_ctzll(live));
        if (__ballot(c != hdb::NONE16 && (c != c0 || s != 1)) == 0ull) {
            if (c0 != cur) {
                if (cur != hdb::NONE16 && lane == 0) t.cacc[cur] = acc;
                cur = c0;
                acc = t.cacc[c0];
            }
#pragma unroll
            for (int j = 0; j < 64; ++j) acc += readlane_f64(term, j);
            continue;
        }
        while (live) {
            const int j = __builtin_ctzll(live);
            live &= live - 1;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
            const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)s, j);
            if (cj != cur) {
                if (cur != hdb::NONE16 && lane == 0) t.cacc[cur] = acc;
                cur = cj;
                acc = t.cacc[cj];
            }
            if (sj == 0) {
                acc += readlane_f64(ta, j);
                acc += readlane_f64(tb, j);
            } else {
                const double tj = readlane_f64(term, j);
                for (uint32_t k = 0; k < sj; ++k) acc += tj;
            }
        }
    }
    if (cur != hdb::NONE16 && lane == 0) t.cacc[cur] = acc;
}

// Hierarchy, labels, cluster weights and the choice of the kept cluster for one map.
// LDS = true: the hot per-point / per-cluster state lives in the dynamic LDS buffer `sm` (all
// pointers derive from it, so the compiler emits ds_* instead of flat_* accesses);
// LDS = false: everything in the per-frame global workspace.  Returns the kept cluster
// (or -1), or -2 when the LDS cluster tables overflowed and the caller must redo in global.
template <bool LDS>
__device__ __forceinline__ int cluster_phase(const TailArgs &A, uint8_t *ws, uint8_t *sm, TreeShared &S, int32_t *hdr,
                                             int N, long long t0) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int hw = A.h * A.w;
    hdb::Edge *ea = (hdb::Edge *)(ws + A.L.ea);
    const hdb::Edge *edges = ea;
    hdb::Tree t;
    t.dparent = (uint32_t *)(ws + A.L.dparent);
    t.nclusters = 0; t.n = N;
    if (LDS) {
        uint8_t *p = sm;
        hdb::Edge *el = carve<hdb::Edge>(p, TREE_LDS_CAP);
        t.absw = carve<uint32_t>(p, TREE_LDS_CAP); t.sdn = carve<uint32_t>(p, TREE_LDS_CAP);
        t.sp = carve<uint16_t>(p, TREE_LDS_CAP); t.ssz = carve<uint16_t>(p, TREE_LDS_CAP);
        t.absc = carve<uint16_t>(p, TREE_LDS_CAP); t.evc = carve<uint16_t>(p, TREE_LDS_CAP);
        t.evs = carve<uint16_t>(p, TREE_LDS_CAP);
        t.cacc = carve<double>(p, TREE_LDS_CLUSTERS);
        t.ctp = carve<int32_t>(p, TREE_LDS_CLUSTERS); t.cleft = carve<int32_t>(p, TREE_LDS_CLUSTERS);
        t.cright = carve<int32_t>(p, TREE_LDS_CLUSTERS); t.cbirthw = carve<uint32_t>(p, TREE_LDS_CLUSTERS);
        t.cminw = carve<uint32_t>(p, TREE_LDS_CLUSTERS); t.csize = carve<uint32_t>(p, TREE_LDS_CLUSTERS);
        t.cdn = carve<uint32_t>(p, TREE_LDS_CLUSTERS); t.csplit = carve<uint32_t>(p, TREE_LDS_CLUSTERS);
        t.cspa = carve<uint32_t>(p, TREE_LDS_CLUSTERS); t.cspb = carve<uint32_t>(p, TREE_LDS_CLUSTERS);
        t.crep = carve<int32_t>(p, TREE_LDS_CLUSTERS); t.cup = carve<uint16_t>(p, TREE_LDS_CLUSTERS);
        t.csel = carve<uint8_t>(p, TREE_LDS_CLUSTERS);
        t.cap_clusters = TREE_LDS_CLUSTERS;
        for (int i = tid; i < N - 1; i += nthr) el[i] = ea[i];
        edges = el;
    } else {
        t.sp = (uint16_t *)(ws + A.L.sp); t.ssz = (uint16_t *)(ws + A.L.ssz); t.absc = (uint16_t *)(ws + A.L.absc);
        t.absw = (uint32_t *)(ws + A.L.absw); t.sdn = (uint32_t *)(ws + A.L.sdn);
        t.evc = (uint16_t *)(ws + A.L.evc); t.evs = (uint16_t *)(ws + A.L.evs);
        t.cup = (uint16_t *)(ws + A.L.cup); t.ctp = (int32_t *)(ws + A.L.ctp); t.cleft = (int32_t *)(ws + A.L.cleft);
        t.cright = (int32_t *)(ws + A.L.cright); t.cbirthw = (uint32_t *)(ws + A.L.cbirthw);
        t.cminw = (uint32_t *)(ws + A.L.cminw); t.csize = (uint32_t *)(ws + A.L.csize);
        t.cdn = (uint32_t *)(ws + A.L.cdn); t.csplit = (uint32_t *)(ws + A.L.csplit);
        t.cspa = (uint32_t *)(ws + A.L.cspa); t.cspb = (uint32_t *)(ws + A.L.cspb); t.cacc = (double *)(ws + A.L.cacc);
        t.csel = (uint8_t *)(ws + A.L.csel); t.crep = (int32_t *)(ws + A.L.crep);
        t.cap_clusters = hdb::max_clusters(hw, A.mcs);
    }
    {
        const int per = (N + nthr - 1) / nthr;
        hdb::init_points(t, min(N, tid * per), min(N, tid * per + per));
    }
    __syncthreads();
    const bool built = build_wave(t, edges, N, A.mcs);       // the block is one wavefront (k_tree)
    if (tid == 0) hdr[13] = (int)(wall_clock64() - t0);
    if (built) accumulate_wave(t, edges);
    if (tid == 0) {
        const bool ok = built;
        if (ok) S.nsel = hdb::choose(t);
        S.ok = ok;
        hdr[4] = t.nclusters;
        hdr[9] = (int)(wall_clock64() - t0);
    }
    __syncthreads();
    if (!S.ok) return -2;
    t.nclusters = hdr[4];
    const int nsel = S.nsel;
    uint32_t *cweight = (uint32_t *)(ws + A.L.cweight);
    for (int c = tid; c < t.nclusters; c += nthr) cweight[c] = 0;
    __syncthreads();
    const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
    int32_t *labels = (int32_t *)(ws + A.L.labels);
    for (int p = tid; p < N; p += nthr) {
        const int c = hdb::point_cluster(t, (uint32_t)p, nsel);
        labels[p] = c;
        if (c >= 0) {
            const uint32_t val = pts[p] >> 16;
            if (A.select_sum == 1) atomicAdd(&cweight[c], val); else atomicMax(&cweight[c], val);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int b = -1;
        for (int c = 0; c < t.nclusters; ++c) {
            if (t.crep[c] != c) continue;
            if (b < 0 || cweight[c] > cweight[b] || (cweight[c] == cweight[b] && hdb::cluster_before(t, c, b))) b = c;
        }
        S.best = b;
        hdr[1] = nsel;
        hdr[2] = b;
        hdr[10] = (int)(wall_clock64() - t0);
    }
    __syncthreads();
    return S.best;
}

// --------------------------------------------------------------------------------------
// k_sort: the MST edges in the order numpy's default argsort gives them (hdbscan sorts the edge list with
// np.argsort before the single-linkage pass; call site smartVidCrop.py:1099).  That sort is an UNSTABLE
// introsort, and on a pixel grid nearly all weights tie, so the order of equal weights is part of the
// reference's result: it decides which components merge first (oracle/npsort_ref.py has the restatement and
// what a different order costs).  The sequential routine is emulated exactly, level by level:
//   * all ranges of more than 16 elements that are alive at one depth of the recursion are partitioned at the same
//     time.  A range's median-of-three and pivot parking are done by one thread; the Hoare scan is data parallel:
//     with L = positions (ascending) whose key is >= the pivot and R = positions (descending) whose key is <= it,
//     the sequential scan swaps L[k] with R[k] while L[k] < R[k] and ends with the left pointer on
//     min(L[K], R[K-1]) -- two prefix sums over the whole array give every stopper its k, the swaps are disjoint.
//   * the depth bookkeeping of the explicit stack is kept per range: the larger side is "pushed" and heap-sorted
//     (by one thread, exactly as numpy's heapsort) if it is reached below the depth limit; the other side
//     continues without a check.  Only adversarial inputs get there.
//   * ranges of at most 16 elements are insertion-sorted by numpy: a stable sort of the range, done here as a rank
//     computation (one thread per element) at the very end.
// Working arrays (16 bytes per edge) live in LDS for N <= ~6000 and in the frame's workspace above that.
// --------------------------------------------------------------------------------------
#ifndef SORT_LDS_BYTES
#define SORT_LDS_BYTES (104 * 1024)
#endif
#define SORT_NONE 0xFFFFu
#define SORT_SMALL 15            // ranges with hi - lo > 15 are partitioned (npsort_ref.SMALL_QUICKSORT)

struct SortArrays {              // n entries each
    uint32_t *key, *scan;
    uint16_t *idx, *seg, *lpos, *rpos;
};
struct SortSegs {                // capacity n / 17 + 2 ranges, two generations
    uint16_t *lo[2], *hi[2], *pi, *ls, *rs;
    int16_t *dep[2];
    uint32_t *vp, *K;
};

__device__ __forceinline__ void sort_swap(const SortArrays &a, int i, int j) {
    const uint32_t k = a.key[i]; a.key[i] = a.key[j]; a.key[j] = k;
    const uint16_t x = a.idx[i]; a.idx[i] = a.idx[j]; a.idx[j] = x;
}

// numpy's aheapsort on positions lo .. lo+n-1 (keys and indices move together)
__device__ __forceinline__ void sort_heapsort(const SortArrays &a, int lo, int n) {
#define HK(i) a.key[lo + (i) - 1]
#define HI(i) a.idx[lo + (i) - 1]
    for (int l = n >> 1; l > 0; --l) {
        const uint32_t tk = HK(l); const uint16_t ti = HI(l);
        int i = l, j = l << 1;
        while (j <= n) {
            if (j < n && HK(j) < HK(j + 1)) ++j;
            if (tk < HK(j)) { HK(i) = HK(j); HI(i) = HI(j); i = j; j += j; } else break;
        }
        HK(i) = tk; HI(i) = ti;
    }
    for (; n > 1;) {
        const uint32_t tk = HK(n); const uint16_t ti = HI(n);
        HK(n) = HK(1); HI(n) = HI(1);
        --n;
        int i = 1, j = 2;
        while (j <= n) {
            if (j < n && HK(j) < HK(j + 1)) ++j;
            if (tk < HK(j)) { HK(i) = HK(j); HI(i) = HI(j); i = j; j += j; } else break;
        }
        HK(i) = tk; HI(i) = ti;
    }
#undef HK
#undef HI
}

// out[p] = index (into the caller's key order) of the element numpy's argsort puts at position p.
// a.key[0..n) holds the keys on entry.  All TB threads of the block call this.
__device__ __forceinline__ void np_argsort_block(const SortArrays &a, const SortSegs &st, int n, uint16_t *out, int *sh /* [NW16 + 4] */) {
    const int tid = threadIdx.x;
    int *cnt = sh + NW16;                                  // cnt[0], cnt[1]: ranges of the two generations
    for (int i = tid; i < n; i += TB) { a.idx[i] = (uint16_t)i; a.seg[i] = 0; a.lpos[i] = 0; a.rpos[i] = (uint16_t)(n - 1); }
    if (tid == 0) {
        cnt[0] = cnt[1] = 0;
        if (n - 1 > SORT_SMALL) {
            int d = 0;
            for (int v = n >> 1; v; v >>= 1) ++d;          // npy_get_msb(n)
            st.lo[0][0] = 0; st.hi[0][0] = (uint16_t)(n - 1); st.dep[0][0] = (int16_t)(2 * d);
            cnt[0] = 1;
        }
    }
    __syncthreads();
    if (cnt[0] == 0) {                                     // one small range: seg stays "retired" with bounds [0, n-1]
        for (int i = tid; i < n; i += TB) a.seg[i] = SORT_NONE;
    }
    const int C = (n + TB - 1) / TB, c_lo = min(n, tid * C), c_hi = min(n, c_lo + C);
    int cur = 0;
    while (true) {
        const int ns = cnt[cur];
        if (ns == 0) break;
        const int nxt = cur ^ 1;
        // (select the generation's arrays with conditional moves: indexing the pointer arrays with a runtime value would
        // put the descriptor structs into a private segment)
        uint16_t *lo_c = cur ? st.lo[1] : st.lo[0], *hi_c = cur ? st.hi[1] : st.hi[0];
        uint16_t *lo_n = cur ? st.lo[0] : st.lo[1], *hi_n = cur ? st.hi[0] : st.hi[1];
        int16_t *dep_c = cur ? st.dep[1] : st.dep[0], *dep_n = cur ? st.dep[0] : st.dep[1];
        // 1. median of three, pivot parked at hi - 1
        for (int s = tid; s < ns; s += TB) {
            const int pl = lo_c[s], pr = hi_c[s], pm = pl + ((pr - pl) >> 1);
            if (a.key[pm] < a.key[pl]) sort_swap(a, pm, pl);
            if (a.key[pr] < a.key[pm]) sort_swap(a, pr, pm);
            if (a.key[pm] < a.key[pl]) sort_swap(a, pm, pl);
            st.vp[s] = a.key[pm];
            sort_swap(a, pm, pr - 1);
            st.K[s] = 0;
        }
        __syncthreads();
        // 2. stoppers of the two scans, counted by one prefix sum over the array (L in the low half, R in the high half)
        uint32_t acc = 0;
        for (int i = c_lo; i < c_hi; ++i) {
            const uint32_t sg = a.seg[i];
            if (sg != SORT_NONE) {
                const int pl = lo_c[sg], pr = hi_c[sg];
                const uint32_t vp = st.vp[sg], k = a.key[i];
                acc += (uint32_t)(i > pl && i <= pr - 1 && k >= vp) | ((uint32_t)(i <= pr - 2 && k <= vp) << 16);
            }
            a.scan[i] = acc;
        }
        int tot;
        const uint32_t off = (uint32_t)block_excl_scan((int)acc, sh, &tot);
        for (int i = c_lo; i < c_hi; ++i) a.scan[i] += off;
        __syncthreads();
        // 3. L[k] ascending, R[k] descending, into the range's own slice of the position arrays
        for (int i = tid; i < n; i += TB) {
            const uint32_t sg = a.seg[i];
            if (sg == SORT_NONE) continue;
            const int pl = lo_c[sg], pr = hi_c[sg];
            const uint32_t vp = st.vp[sg], k = a.key[i], sc = a.scan[i], base = pl > 0 ? a.scan[pl - 1] : 0u;
            if (i > pl && i <= pr - 1 && k >= vp) a.lpos[pl + (int)((sc & 0xFFFFu) - (base & 0xFFFFu)) - 1] = (uint16_t)i;
            if (i <= pr - 2 && k <= vp) a.rpos[pl + (int)((a.scan[pr - 2] >> 16) - (sc >> 16))] = (uint16_t)i;
        }
        __syncthreads();
        // 4. the swaps of the scan: L[k] <-> R[k] while L[k] < R[k]
        for (int i = tid; i < n; i += TB) {
            const uint32_t sg = a.seg[i];
            if (sg == SORT_NONE) continue;
            const int pl = lo_c[sg], pr = hi_c[sg], k = i - pl;
            const uint32_t base = pl > 0 ? a.scan[pl - 1] : 0u;
            const int nL = (int)((a.scan[pr - 1] & 0xFFFFu) - (base & 0xFFFFu)), nR = (int)((a.scan[pr - 2] >> 16) - (base >> 16));
            if (k < nL && k < nR) {
                const int l = a.lpos[pl + k], r = a.rpos[pl + k];
                if (l < r) { sort_swap(a, l, r); atomicMax(&st.K[sg], (uint32_t)(k + 1)); }
            }
        }
        __syncthreads();
        // 5. where the left pointer ends, the pivot goes there, the two sides become ranges of the next generation
        for (int s = tid; s < ns; s += TB) {
            const int pl = lo_c[s], pr = hi_c[s], K = (int)st.K[s];
            const uint32_t base = pl > 0 ? a.scan[pl - 1] : 0u;
            const int nL = (int)((a.scan[pr - 1] & 0xFFFFu) - (base & 0xFFFFu));
            const int rprev = K >= 1 ? (int)a.rpos[pl + K - 1] : pr;
            int pi = rprev;
            if (K < nL && (int)a.lpos[pl + K] < rprev) pi = a.lpos[pl + K];
            sort_swap(a, pi, pr - 1);
            st.pi[s] = (uint16_t)pi;
            const int d = dep_c[s] - 1;
            const bool push_right = (pi - pl) < (pr - pi);          // the larger side goes on numpy's stack
            uint16_t slot[2];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int lo = side ? pi + 1 : pl, hi = side ? pr : pi - 1;
                const bool popped = side ? push_right : !push_right;
                slot[side] = SORT_NONE;
                if (hi < lo) continue;
                if (popped && d < 0) {                               // depth limit: numpy heap-sorts a popped range
                    sort_heapsort(a, lo, hi - lo + 1);
                    for (int e = lo; e <= hi; ++e) { a.lpos[e] = (uint16_t)e; a.rpos[e] = (uint16_t)e; }
                } else if (hi - lo > SORT_SMALL) {
                    const int q = atomicAdd(&cnt[nxt], 1);
                    lo_n[q] = (uint16_t)lo; hi_n[q] = (uint16_t)hi; dep_n[q] = (int16_t)d;
                    slot[side] = (uint16_t)q;
                } else {
                    for (int e = lo; e <= hi; ++e) { a.lpos[e] = (uint16_t)lo; a.rpos[e] = (uint16_t)hi; }
                }
            }
            a.lpos[pi] = (uint16_t)pi; a.rpos[pi] = (uint16_t)pi;
            st.ls[s] = slot[0]; st.rs[s] = slot[1];
        }
        __syncthreads();
        // 6. every element moves to its side's range
        for (int i = tid; i < n; i += TB) {
            const uint32_t sg = a.seg[i];
            if (sg == SORT_NONE) continue;
            const int pi = st.pi[sg];
            a.seg[i] = i < pi ? st.ls[sg] : (i > pi ? st.rs[sg] : (uint16_t)SORT_NONE);
        }
        if (tid == 0) cnt[cur] = 0;
        cur = nxt;
        __syncthreads();
    }
    // insertion sort of every remaining range = a stable sort of the range: rank of each element
    for (int i = tid; i < n; i += TB) {
        const int lo = a.lpos[i], hi = a.rpos[i];
        const uint32_t k = a.key[i];
        int rank = 0;
        for (int j = lo; j <= hi; ++j) {
            const uint32_t kj = a.key[j];
            rank += (kj < k || (kj == k && j < i)) ? 1 : 0;
        }
        out[lo + rank] = a.idx[i];
    }
    __syncthreads();
}

// carve the working arrays of np_argsort_block: in LDS when they fit, else in the caller's global scratch
__device__ __forceinline__ void sort_carve(uint8_t *lds, uint8_t *glob, int n, SortArrays &a, SortSegs &st, int *&sh) {
    uint8_t *p = lds;
    sh = carve<int>(p, NW16 + 4);
    const int nseg = n / 17 + 2;
    for (int g = 0; g < 2; ++g) { st.lo[g] = carve<uint16_t>(p, nseg); st.hi[g] = carve<uint16_t>(p, nseg); st.dep[g] = carve<int16_t>(p, nseg); }
    st.pi = carve<uint16_t>(p, nseg); st.ls = carve<uint16_t>(p, nseg); st.rs = carve<uint16_t>(p, nseg);
    st.vp = carve<uint32_t>(p, nseg); st.K = carve<uint32_t>(p, nseg);
    uint8_t *q = ((size_t)(p - lds) + (size_t)n * 16 + 6 * 16 <= SORT_LDS_BYTES) ? p : glob;
    a.key = carve<uint32_t>(q, n); a.scan = carve<uint32_t>(q, n);
    a.idx = carve<uint16_t>(q, n); a.seg = carve<uint16_t>(q, n); a.lpos = carve<uint16_t>(q, n); a.rpos = carve<uint16_t>(q, n);
}

__device__ __forceinline__ void sort_body(const TailArgs &A) {
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (!hdr[3]) return;
    const int n = hdr[0] - 1;
    extern __shared__ uint8_t sm_sort[];
    const long long t0 = wall_clock64();
    const hdb::Edge *mst = (const hdb::Edge *)(ws + A.L.mst);
    hdb::Edge *ea = (hdb::Edge *)(ws + A.L.ea);
    SortArrays a; SortSegs st; int *sh;
    sort_carve(sm_sort, ws + A.L.srt, n, a, st, sh);
    for (int i = threadIdx.x; i < n; i += TB) a.key[i] = mst[i].w;
    uint16_t *perm = (uint16_t *)(ws + A.L.eb);            // sorted position -> Prim position
    np_argsort_block(a, st, n, perm, sh);
    for (int i = threadIdx.x; i < n; i += TB) ea[i] = mst[perm[i]];
    if (threadIdx.x == 0) hdr[8] = (int)(wall_clock64() - t0);
}
__global__ __launch_bounds__(TB) void k_sort(TailArgs A) { sort_body(A); }

// test door (svc_debug_argsort_u32): the same routine on arbitrary keys
__global__ __launch_bounds__(TB) void k_argsort_test(const uint32_t *keys, int n, uint16_t *out, uint8_t *scratch) {
    extern __shared__ uint8_t sm_sort[];
    SortArrays a; SortSegs st; int *sh;
    sort_carve(sm_sort, scratch, n, a, st, sh);
    for (int i = threadIdx.x; i < n; i += TB) a.key[i] = keys[i];
    np_argsort_block(a, st, n, out, sh);
}

// k_tree: hierarchy + excess of mass + labels + cluster weights + the kept cluster.  The build is a
// serial union-find pass, so the map gets ONE wavefront (64 threads): the other SIMDs and wave slots of
// the CU stay free for the network kernels of the next batch (which use no LDS).
__global__ __launch_bounds__(64) void k_tree(TailArgs A) {
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (!hdr[3] || hdr[23]) return;
    const int N = hdr[0];
    extern __shared__ uint8_t sm_tree[];
    __shared__ TreeShared S;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    int best = -2;
    if (N <= TREE_LDS_CAP) best = cluster_phase<true>(A, ws, sm_tree, S, hdr, N, t0);
    if (best == -2) best = cluster_phase<false>(A, ws, sm_tree, S, hdr, N, t0);
    if (threadIdx.x == 0) { hdr[14] = (int)(clock64() - c0); hdr[15] = (int)(wall_clock64() - t0); }
}

// --------------------------------------------------------------------------------------
// k_tree_par: the hierarchy as data-parallel passes (tools/sim/tree_path.py is the executable specification, checked
// against oracle/hdbscan_ref: single_linkage -> condense_tree -> select_and_label).
//
// The library's Prim records every edge as (last node added, new node, weight), so the "MST" that its single
// linkage sees is a PATH through the points in Prim order: edge k joins positions k and k + 1, and the union-find
// pass over the edges in sorted order merges ADJACENT INTERVALS.  Hence, with rank = an edge's sorted position:
//   * when edge k is processed its sides reach out to the nearest edges of greater rank: a side (left) = k - PGE(k)
//     points, b side = NGE(k) - k; the dendrogram is the Cartesian tree of the ranks, parent(k) = the lower-ranked of
//     PGE(k), NGE(k);
//   * by its side sizes alone an edge is a small union, the birth of a condensed cluster, a true split or an absorption;
//     clusters are numbered in rank order of their birth / split (= hdb::build's creation order);
//   * the cluster on top of a big side is found by following "big child" links down to the first birth / split
//     (pointer jumping, with the distance: the rows of a cluster in the library's order are its chain top-down);
//   * a point falls out at the first ancestor of its leaf that is not a small union.
// Only the float64 stability sums stay serial per cluster (one wavefront per cluster: terms in parallel, additions in
// the library's order), as in hdb::accumulate.  N - 1 serial union-find steps become ~15 block-wide passes.
// --------------------------------------------------------------------------------------
#ifndef TP_CAP
#define TP_CAP 4352                     // points per map with everything in LDS
#endif
#ifndef TP_CAP_BIG
#define TP_CAP_BIG 8192                 // ... with the jump buffers, weights and order in the frame's workspace
#endif
#define TP_CAP_HUGE 65025               // ... with every per-edge array there (L2-resident); LDS keeps the rank-maxima levels and the cluster tables
#define TP_NONE 0xFFFFu
enum { TP_SMALL = 0, TP_BIRTH = 1, TP_SPLIT = 2, TP_ABS_A = 3, TP_ABS_B = 4 };   // ABS_A: the a side is big, the b side falls out

struct TpCl {            // per condensed cluster
    double *acc;
    uint32_t *birthw, *minw, *weight;
    uint16_t *tp, *left, *right, *spa, *spb, *node, *topk, *off, *len;
    int16_t *rep;
    uint8_t *sel;
};

static size_t tp_lds_bytes(int hw, int mcs, int *cap_clusters, int *cap_clusters_huge = nullptr) {
    const int cap = std::min(hw, TP_CAP), capb = std::min(hw, TP_CAP_BIG);
    auto up = [](size_t b) { return (b + 15) / 16 * 16; };
    const size_t small = up((size_t)cap * 2) * 8 + up((size_t)(cap + 256) * 2) + up((size_t)cap * 4) * 3 + up((size_t)((cap + 256) / 16 + 16) * 2) + 64;
    const size_t big = hw > TP_CAP ? up((size_t)capb * 2) * 6 + up((size_t)(capb + 256) * 2) + up((size_t)((capb + 256) / 16 + 16) * 2) + 64 : 0;
    const size_t huge = hw > TP_CAP_BIG ? up((size_t)((hw + 256) / 16 + 16) * 2) + up(((size_t)hw / 256 + 32) * 2) + up(32 * 2) + 64 : 0;
    const size_t per_edge = std::max({small, big, huge});
    int cc = hdb::max_clusters(std::min(hw, TP_CAP_HUGE), mcs);
#ifndef TP_LDS_BUDGET
#define TP_LDS_BUDGET (160 * 1024 - 4096)     // LDS a tail workgroup may ask for (experiment builds: 78 KB = two workgroups per CU)
#endif
    const size_t budget = TP_LDS_BUDGET;
#ifdef TP_CC_EXACT                                         // (experiment builds with a small budget: every cluster the budget holds, not the next power-of-two fraction)
    if (per_edge + (size_t)cc * 48 + 512 > budget) cc = std::max(8, (int)((budget > per_edge + 512 ? budget - per_edge - 512 : 0) / 48));
#else
    while (cc > 8 && per_edge + (size_t)cc * 48 + 512 > budget) cc /= 2;
#endif
    *cap_clusters = cc;
    const size_t total = per_edge + (size_t)cc * 48 + 512;
    if (cap_clusters_huge) {                               // MODE 2 keeps only the rank-maxima levels in LDS: the rest of the launch's allocation is cluster tables
        const size_t room = total > huge + 512 ? (total - huge - 512) / 48 : 0;
        *cap_clusters_huge = (int)std::min<size_t>(room, (size_t)hdb::max_clusters(std::min(hw, TP_CAP_HUGE), mcs));
    }
    return total;
}

// bit i set when the i-th of the 16 uint16 at p (32-byte aligned) is greater than r
__device__ __forceinline__ uint32_t tp_gt_mask16(const uint16_t *p, uint32_t r) {
    const uint4 a = ((const uint4 *)p)[0], b = ((const uint4 *)p)[1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m |= ((v[i] & 0xFFFFu) > r ? 1u << (2 * i) : 0u) | ((v[i] >> 16) > r ? 2u << (2 * i) : 0u);
    return m;
}

__device__ __forceinline__ int tp_class(int sa, int sb, int mcs) {
    if (sa + sb < mcs) return TP_SMALL;
    if (sa < mcs && sb < mcs) return TP_BIRTH;
    if (sa >= mcs && sb >= mcs) return TP_SPLIT;
    return sa >= mcs ? TP_ABS_A : TP_ABS_B;
}

// MODE 0: maps of up to TP_CAP points, everything in LDS.  MODE 1 (BIG): up to TP_CAP_BIG points -- the weights and
// the sorted order are read from the frame's workspace, the two jump buffers live there, the row lists re-use the
// child links (dead by then), so that the rest still fits in LDS.  MODE 2 (HUGE, round 4): any map the tail takes -- the
// seven per-edge arrays live in the workspace too (14 B per edge: 490 KB at 35 000 points, L2-resident), the passes are the
// same block-wide gathers; the nearest-greater search walks four levels of rank maxima (fan-out 16: 16^4 = 65 536 >= E)
// instead of three.  Replaces the serial k_tree (one wavefront: 4 - 11 ms per map at 10 - 35 k points) on the default path.
template <int MODE>
__device__ __forceinline__ void tp_body(const TailArgs &A, int cap_clusters, uint8_t *ws, int32_t *hdr, int N, uint8_t *sm_tp,
                                        int *lds16, int &sh_nc, int &sh_nsel) {
    const int E = N - 1, mcs = A.mcs;
#define TP_STAMP(i) do { if (tid == 0) hdr[25 + (i)] = (int)(wall_clock64() - t0); } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t0 = wall_clock64();
    constexpr bool BIG = MODE >= 1, HUGE = MODE == 2;
    const int cap = min(A.h * A.w, HUGE ? TP_CAP_HUGE : (BIG ? TP_CAP_BIG : TP_CAP));
    uint8_t *p = sm_tp;
    // HUGE: regions of the frame's workspace that only the serial builder (k_tree) and the finished Prim use
    uint16_t *rho = HUGE ? (uint16_t *)(ws + A.L.absw) : carve<uint16_t>(p, cap + 256);
    uint16_t *sa = HUGE ? (uint16_t *)(ws + A.L.sp) : carve<uint16_t>(p, cap), *sb = HUGE ? (uint16_t *)(ws + A.L.ssz) : carve<uint16_t>(p, cap);
    uint16_t *par = HUGE ? (uint16_t *)(ws + A.L.absc) : carve<uint16_t>(p, cap), *lch = HUGE ? (uint16_t *)(ws + A.L.evc) : carve<uint16_t>(p, cap);
    uint16_t *rch = HUGE ? (uint16_t *)(ws + A.L.evs) : carve<uint16_t>(p, cap), *cid = HUGE ? (uint16_t *)(ws + A.L.sdn) : carve<uint16_t>(p, cap);
    uint16_t *order = BIG ? nullptr : carve<uint16_t>(p, cap);
    uint16_t *rowlist = BIG ? lch : carve<uint16_t>(p, cap);
    uint32_t *w = BIG ? nullptr : carve<uint32_t>(p, cap);
    uint32_t *jA = BIG ? (uint32_t *)(ws + A.L.dparent) : carve<uint32_t>(p, cap);
    uint32_t *jB = BIG ? (uint32_t *)(ws + A.L.dparent) + (size_t)A.h * A.w : carve<uint32_t>(p, cap);
    uint16_t *l1 = carve<uint16_t>(p, (cap + 256) / 16 + 16), *l2 = carve<uint16_t>(p, HUGE ? cap / 256 + 32 : 32);
    uint16_t *l3 = HUGE ? carve<uint16_t>(p, 32) : nullptr;
#define TP_ORDER(s_) (BIG ? (int)perm[s_] : (int)order[s_])
#define TP_W(k_) (BIG ? mst[k_].w : w[k_])
    TpCl C;
    C.acc = carve<double>(p, cap_clusters);
    C.birthw = carve<uint32_t>(p, cap_clusters); C.minw = carve<uint32_t>(p, cap_clusters); C.weight = carve<uint32_t>(p, cap_clusters);
    C.tp = carve<uint16_t>(p, cap_clusters); C.left = carve<uint16_t>(p, cap_clusters); C.right = carve<uint16_t>(p, cap_clusters);
    C.spa = carve<uint16_t>(p, cap_clusters); C.spb = carve<uint16_t>(p, cap_clusters); C.node = carve<uint16_t>(p, cap_clusters);
    C.topk = carve<uint16_t>(p, cap_clusters); C.off = carve<uint16_t>(p, cap_clusters); C.len = carve<uint16_t>(p, cap_clusters);
    C.rep = carve<int16_t>(p, cap_clusters); C.sel = carve<uint8_t>(p, cap_clusters);
    const hdb::Edge *mst = (const hdb::Edge *)(ws + A.L.mst);
    const uint16_t *perm = (const uint16_t *)(ws + A.L.eb);            // k_sort: sorted position -> Prim position
    // ---- ranks, weights, block maxima of the ranks
    for (int s0 = tid; s0 < E; s0 += TB) {
        const uint32_t k = perm[s0];
        if (!BIG) { order[s0] = (uint16_t)k; w[s0] = mst[s0].w; }      // (w: index = Prim position)
        rho[k] = (uint16_t)s0;
        lch[s0] = rch[s0] = TP_NONE;
    }
    for (int i = E + tid; i < ((E + 255) & ~255); i += TB) rho[i] = 0;     // padding: never "greater"
    __syncthreads();
    // maxima over groups of 16 ranks and over groups of 16 groups (the rank array is padded with zeros to whole groups)
    const int n1 = (E + 15) >> 4, n2 = (n1 + 15) >> 4;
    for (int g = tid; g < n2 * 16; g += TB) {
        uint32_t mx = 0;
        if (g < n1) {
            const uint4 a = ((const uint4 *)(rho + 16 * g))[0], b = ((const uint4 *)(rho + 16 * g))[1];
            const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) mx = max(mx, max(v[i] & 0xFFFFu, v[i] >> 16));
        }
        l1[g] = (uint16_t)mx;
    }
    __syncthreads();
    const int n3 = (n2 + 15) >> 4;                                        // HUGE: <= 16 groups of 16 second-level maxima
    for (int g = tid; g < (HUGE ? n3 * 16 : 32); g += TB) {
        uint32_t mx = 0;
        if (g < n2) for (int i = 0; i < 16; ++i) mx = max(mx, (uint32_t)l1[16 * g + i]);
        l2[g] = (uint16_t)mx;
    }
    __syncthreads();
    if (HUGE) {
        for (int g = tid; g < 16; g += TB) {
            uint32_t mx = 0;
            if (g < n3) for (int i = 0; i < 16; ++i) mx = max(mx, (uint32_t)l2[16 * g + i]);
            l3[g] = (uint16_t)mx;
        }
        __syncthreads();
    }
    // ---- nearest greater ranks on both sides -> side sizes, Cartesian-tree parent, children
    for (int k = tid; k < E; k += TB) {
        const uint32_t r = rho[k];
        const int g0 = k >> 4, s0 = g0 >> 4;
        // right: first position > k whose rank is greater
        int nge = E;
        {
            uint32_t m = tp_gt_mask16(rho + 16 * g0, r) & ~((2u << (k & 15)) - 1u);
            if (m) nge = 16 * g0 + __builtin_ctz(m);
            else {
                int g = -1;
                uint32_t m1 = tp_gt_mask16(l1 + 16 * s0, r) & ~((2u << (g0 & 15)) - 1u);
                if (m1) g = 16 * s0 + __builtin_ctz(m1);
                else if (HUGE) {
                    const int t0_ = s0 >> 4;
                    int sx = -1;
                    const uint32_t m2 = tp_gt_mask16(l2 + 16 * t0_, r) & ~((2u << (s0 & 15)) - 1u);
                    if (m2) sx = 16 * t0_ + __builtin_ctz(m2);
                    else {
                        const uint32_t m3 = tp_gt_mask16(l3, r) & ~((2u << t0_) - 1u);
                        if (m3) {
                            const int tx = __builtin_ctz(m3);
                            sx = 16 * tx + __builtin_ctz(tp_gt_mask16(l2 + 16 * tx, r));
                        }
                    }
                    if (sx >= 0) g = 16 * sx + __builtin_ctz(tp_gt_mask16(l1 + 16 * sx, r));
                } else {
                    uint32_t m2 = (tp_gt_mask16(l2, r) | (tp_gt_mask16(l2 + 16, r) << 16)) & ~((2u << s0) - 1u);
                    if (s0 >= 31) m2 = 0;
                    if (m2) {
                        const int sx = __builtin_ctz(m2);
                        g = 16 * sx + __builtin_ctz(tp_gt_mask16(l1 + 16 * sx, r));
                    }
                }
                if (g >= 0) nge = 16 * g + __builtin_ctz(tp_gt_mask16(rho + 16 * g, r));
            }
        }
        // left: last position < k whose rank is greater
        int j = -1;
        {
            uint32_t m = tp_gt_mask16(rho + 16 * g0, r) & ((1u << (k & 15)) - 1u);
            if (m) j = 16 * g0 + 31 - __builtin_clz(m);
            else {
                int g = -1;
                uint32_t m1 = tp_gt_mask16(l1 + 16 * s0, r) & ((1u << (g0 & 15)) - 1u);
                if (m1) g = 16 * s0 + 31 - __builtin_clz(m1);
                else if (HUGE) {
                    const int t0_ = s0 >> 4;
                    int sx = -1;
                    const uint32_t m2 = tp_gt_mask16(l2 + 16 * t0_, r) & ((1u << (s0 & 15)) - 1u);
                    if (m2) sx = 16 * t0_ + 31 - __builtin_clz(m2);
                    else {
                        const uint32_t m3 = tp_gt_mask16(l3, r) & ((1u << t0_) - 1u);
                        if (m3) {
                            const int tx = 31 - __builtin_clz(m3);
                            sx = 16 * tx + 31 - __builtin_clz(tp_gt_mask16(l2 + 16 * tx, r));
                        }
                    }
                    if (sx >= 0) g = 16 * sx + 31 - __builtin_clz(tp_gt_mask16(l1 + 16 * sx, r));
                } else {
                    const uint32_t m2 = (tp_gt_mask16(l2, r) | (tp_gt_mask16(l2 + 16, r) << 16)) & ((1u << s0) - 1u);
                    if (m2) {
                        const int sx = 31 - __builtin_clz(m2);
                        g = 16 * sx + 31 - __builtin_clz(tp_gt_mask16(l1 + 16 * sx, r));
                    }
                }
                if (g >= 0) j = 16 * g + 31 - __builtin_clz(tp_gt_mask16(rho + 16 * g, r));
            }
        }
        const int pge = j < 0 ? -1 : j;
        sa[k] = (uint16_t)(k - pge);
        sb[k] = (uint16_t)(nge - k);
        int pr = -1;
        if (pge >= 0 && (nge >= E || rho[pge] < rho[nge])) pr = pge;
        else if (nge < E) pr = nge;
        par[k] = pr < 0 ? TP_NONE : (uint16_t)pr;
        if (pr >= 0) { if (k < pr) lch[pr] = (uint16_t)k; else rch[pr] = (uint16_t)k; }
    }
    __syncthreads();
    TP_STAMP(1);
    // ---- clusters in creation order (= rank order of the births and splits)
    {
        const int per = (E + TB - 1) / TB, lo = min(E, tid * per), hi = min(E, lo + per);
        int cnt = 0;
        for (int s0 = lo; s0 < hi; ++s0) {
            const int k = TP_ORDER(s0), cl = tp_class(sa[k], sb[k], mcs);
            cnt += cl == TP_BIRTH || cl == TP_SPLIT;
        }
        int tot;
        int id = block_excl_scan(cnt, lds16, &tot);
        if (tid == 0) sh_nc = tot;
        for (int s0 = lo; s0 < hi; ++s0) {
            const int k = TP_ORDER(s0), cl = tp_class(sa[k], sb[k], mcs);
            const bool isc = cl == TP_BIRTH || cl == TP_SPLIT;
            cid[k] = isc ? (uint16_t)id : TP_NONE;
            if (isc && id < cap_clusters) { C.node[id] = (uint16_t)k; C.minw[id] = TP_W(k); C.tp[id] = TP_NONE; C.birthw[id] = 0; C.acc[id] = 0.0; C.weight[id] = 0; }
            id += isc;
            // jump pointer: (next edge down the chain of the big side) | distance << 16; births, splits, small unions end it
            const uint32_t nx = cl == TP_ABS_A ? lch[k] : (cl == TP_ABS_B ? rch[k] : (uint32_t)k);
            jA[k] = nx | ((nx != (uint32_t)k ? 1u : 0u) << 16);
        }
    }
    __syncthreads();
    const int nc = sh_nc;
    if (nc > cap_clusters) return;                                     // tables too small: k_tree redoes the map (hdr[23] stays 0)
    TP_STAMP(2);
    // ---- pointer jumping down the big-child chains: every absorption learns its cluster's first edge and its distance to it
    uint32_t *ja = jA, *jb = jB;
    for (int round = 0; round < 16; ++round) {
        int changed = 0;
        for (int k = tid; k < E; k += TB) {
            const uint32_t a1 = ja[k], n1 = a1 & 0xFFFFu, a2 = ja[n1];
            jb[k] = (a2 & 0xFFFFu) | ((a1 & 0xFFFF0000u) + (a2 & 0xFFFF0000u));
            changed |= (a2 & 0xFFFFu) != n1;
        }
        uint32_t *t_ = ja; ja = jb; jb = t_;
        if (!__syncthreads_or(changed)) break;
    }
    TP_STAMP(3);
    // ja[k] = (edge that created the cluster of k's merged component | rows between them); cluster = cid[that edge]
    // ---- splits: children, tree parents, birth weights; chains: top edge and length of every cluster
    for (int k = tid; k < E; k += TB) {
        const int cl = tp_class(sa[k], sb[k], mcs);
        if (cl == TP_SPLIT) {
            const uint32_t pc = cid[k];
            const uint32_t l = cid[ja[lch[k]] & 0xFFFFu], r = cid[ja[rch[k]] & 0xFFFFu];
            C.left[pc] = (uint16_t)l; C.right[pc] = (uint16_t)r;
            C.tp[l] = C.tp[r] = (uint16_t)pc;
            C.birthw[l] = C.birthw[r] = TP_W(k);
            C.spa[pc] = sa[k]; C.spb[pc] = sb[k];
        } else if (cl == TP_BIRTH) {
            C.left[cid[k]] = C.right[cid[k]] = TP_NONE;
        }
        if (cl != TP_SMALL) {
            const uint32_t pr = par[k];
            if (pr == TP_NONE || tp_class(sa[pr], sb[pr], mcs) == TP_SPLIT) {   // the top of its cluster's chain
                const uint32_t j = ja[k], c = cid[j & 0xFFFFu];
                C.topk[c] = (uint16_t)k;
                C.len[c] = (uint16_t)((j >> 16) + 1);
            }
        }
    }
    __syncthreads();
    TP_STAMP(4);
    if (tid == 0) {                                                     // offsets of the clusters' row lists
        uint32_t o = 0;
        for (int c = 0; c < nc; ++c) { C.off[c] = (uint16_t)o; o += C.len[c]; }
    }
    __syncthreads();
    for (int k = tid; k < E; k += TB) {
        if (tp_class(sa[k], sb[k], mcs) == TP_SMALL) continue;
        const uint32_t j = ja[k], c = cid[j & 0xFFFFu];
        rowlist[C.off[c] + (C.len[c] - 1 - (j >> 16))] = (uint16_t)k;  // top of the chain first = descending rank
    }
    __syncthreads();
    if (tid == 0) hdr[13] = (int)(wall_clock64() - t0);
    // ---- stabilities: one wavefront per cluster, rows in the library's order (hdb::accumulate), terms 64 at a time
    for (int c = wave; c < nc; c += NW16) {
        const uint32_t bw = C.birthw[c];
        const double birth = bw ? 1.0 / (double)bw : 0.0;
        double acc = 0.0;
        const int off = C.off[c], len = C.len[c];
        for (int base = 0; base < len; base += 64) {
            const int i = base + lane;
            uint32_t cnt = 0;
            double term = 0.0, ta = 0.0, tb = 0.0;
            if (i < len) {
                const int k = rowlist[off + i];
                const int cl = tp_class(sa[k], sb[k], mcs);
                const double lam = 1.0 / (double)TP_W(k);
                term = (lam - birth) * 1.0;
                if (cl == TP_SPLIT) { ta = (lam - birth) * (double)C.spa[c]; tb = (lam - birth) * (double)C.spb[c]; cnt = 0; }
                else cnt = cl == TP_BIRTH ? (uint32_t)(sa[k] + sb[k]) : (cl == TP_ABS_A ? sb[k] : sa[k]);
            }
            const int m = min(64, len - base);
            if (m == 64 && __ballot(cnt != 1u) == 0ull) {              // the common batch: 64 single points falling out
#pragma unroll
                for (int j = 0; j < 64; ++j) acc += readlane_f64(term, j);
                continue;
            }
            for (int j = 0; j < m; ++j) {
                const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)cnt, j);
                if (sj == 0) {
                    acc += readlane_f64(ta, j);
                    acc += readlane_f64(tb, j);
                } else {
                    const double tj = readlane_f64(term, j);
                    for (uint32_t q = 0; q < sj; ++q) acc += tj;
                }
            }
        }
        if (lane == 0) C.acc[c] = acc;
    }
    __syncthreads();
    // ---- excess of mass (hdb::choose): children before parents, then parents before children
    if (tid == 0) {
        for (int c = 0; c < nc; ++c) {
            double stab = C.acc[c], sub = 0.0;
            if (C.left[c] != TP_NONE) sub = C.acc[C.left[c]] + C.acc[C.right[c]];
            if (sub > stab) { C.sel[c] = 0; stab = sub; } else C.sel[c] = 1;
            C.acc[c] = stab;
        }
        int nsel = 0;
        for (int c = nc - 1; c >= 0; --c) {
            const uint32_t pp = C.tp[c];
            if (pp != TP_NONE && C.rep[pp] >= 0) { C.sel[c] = 0; C.rep[c] = C.rep[pp]; }
            else if (C.sel[c]) { C.rep[c] = (int16_t)c; ++nsel; }
            else C.rep[c] = (int16_t)hdb::ROOT_NOISE;
        }
        sh_nsel = nsel;
        hdr[4] = nc;
        hdr[9] = (int)(wall_clock64() - t0);
    }
    // ---- (meanwhile) the edge at which every small union's component falls out: first ancestor that is not one
    uint16_t *fo = (uint16_t *)jb;                                      // the spare jump buffer
    for (int k = tid; k < E; k += TB) fo[k] = tp_class(sa[k], sb[k], mcs) == TP_SMALL ? par[k] : (uint16_t)k;
    __syncthreads();
    for (int round = 0; round < 16; ++round) {
        int changed = 0;
        for (int k = tid; k < E; k += TB) {
            const uint32_t f1 = fo[k], f2 = fo[f1];
            if (f2 != f1) { fo[k] = (uint16_t)f2; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // ---- labels and cluster weights (hdb::point_cluster; smartVidCrop.py:1107-1114)
    const int nsel = sh_nsel;
    const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
    int32_t *labels = (int32_t *)(ws + A.L.labels);
    for (int i = tid; i < N; i += TB) {
        int e;
        if (i == 0) e = 0; else if (i == E) e = E - 1; else e = rho[i - 1] < rho[i] ? i - 1 : i;
        e = fo[e];
        const uint32_t c0 = cid[ja[e] & 0xFFFFu];
        const int rep = C.rep[c0];
        int lab = rep;
        if (rep == hdb::ROOT_NOISE) lab = -1;
        else if (rep == nc - 1 && nsel == 1) lab = TP_W(e) <= C.minw[nc - 1] ? rep : -1;
        const uint32_t pid = i == 0 ? mst[0].a : mst[i - 1].b;
        labels[pid] = lab;
        if (lab >= 0) {
            const uint32_t val = pts[pid] >> 16;
            if (A.select_sum == 1) atomicAdd(&C.weight[lab], val); else atomicMax(&C.weight[lab], val);
        }
    }
    __syncthreads();
    if (tid == 0) {
        // first arg-max in the library's cluster numbering (BFS order of the dendrogram; hdb::cluster_before)
        auto depth = [&](uint32_t k) { int d = 0; while (par[k] != TP_NONE) { k = par[k]; ++d; } return d; };
        auto before = [&](int c1, int c2) -> bool {
            const uint32_t p1 = C.tp[c1], p2 = C.tp[c2];
            if (p1 == TP_NONE) return true;
            if (p2 == TP_NONE) return false;
            if (p1 == p2) return C.left[p1] == c1;
            uint32_t s1 = C.node[p1], s2 = C.node[p2];
            const int d1 = depth(s1), d2 = depth(s2);
            if (d1 != d2) return d1 < d2;
            uint32_t side1 = 0, side2 = 0;
            while (s1 != s2) {
                side1 = s1 > par[s1]; side2 = s2 > par[s2];
                s1 = par[s1]; s2 = par[s2];
            }
            return side1 < side2;
        };
        int b = -1;
        for (int c = 0; c < nc; ++c) {
            if (C.rep[c] != c) continue;
            if (b < 0 || C.weight[c] > C.weight[b] || (C.weight[c] == C.weight[b] && before(c, b))) b = c;
        }
        hdr[1] = nsel;
        hdr[2] = b;
        hdr[23] = 1;                                                    // done: k_tree leaves the map alone
        hdr[10] = (int)(wall_clock64() - t0);
        hdr[15] = hdr[10];
    }
}
#undef TP_ORDER
#undef TP_W

__device__ __forceinline__ void tree_par_body(const TailArgs &A, int cap_clusters, int cap_clusters_huge) {
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (!hdr[3]) return;
    const int N = hdr[0];
    if (N > TP_CAP_HUGE) return;                                       // (never: the tail takes maps of up to 255 x 255)
    extern __shared__ uint8_t sm_tp[];
    __shared__ int lds16[NW16];
    __shared__ int sh_nc, sh_nsel;
    if (N <= TP_CAP) tp_body<0>(A, cap_clusters, ws, hdr, N, sm_tp, lds16, sh_nc, sh_nsel);
    else if (N <= TP_CAP_BIG) tp_body<1>(A, cap_clusters, ws, hdr, N, sm_tp, lds16, sh_nc, sh_nsel);
    else tp_body<2>(A, cap_clusters_huge, ws, hdr, N, sm_tp, lds16, sh_nc, sh_nsel);
}
__global__ __launch_bounds__(TB) void k_tree_par(TailArgs A, int cap_clusters, int cap_clusters_huge) { tree_par_body(A, cap_clusters, cap_clusters_huge); }

// k_finish: zero everything outside the kept cluster, CLOSE 5x5, write the map back, centroid
__device__ __forceinline__ void finish_body(const TailArgs &A) {
    const int f = A.order[blockIdx.x];
    uint8_t *ws = A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride;
    int32_t *hdr = (int32_t *)(ws + A.L.hdr);
    if (hdr[30]) return;                                               // k_tail_back has finished this map already
    const int N = hdr[0];
    const int hw = A.h * A.w;
    uint8_t *map = A.maps + (size_t)f * hw;
    extern __shared__ uint8_t sm_fin[];
    __shared__ unsigned long long red[3 * NW16];
    __shared__ uint32_t box[4];
    const int tid = threadIdx.x;
    if (tid < 4) box[tid] = 255u;                          // (published by the barrier behind the map copy)
    const bool clustered = hdr[3] != 0;
    long long t0 = 0;
    if (tid == 0) t0 = wall_clock64();
    const int best = clustered ? hdr[2] : -1;
    // ---- map phase: the LDS buffer now holds the map (the hierarchy state is no longer needed)
    uint8_t *m0 = sm_fin;                            // [hw]
    uint8_t *m1 = sm_fin + (hw + 15) / 16 * 16;      // [hw]
    copy_bytes(m0, map, hw);
    __syncthreads();
    int wr0 = 0, wc0 = 0, wh = A.h, ww = A.w;                // the window the CLOSE and the centroid have to look at
    if (best >= 0) {
        const uint32_t *pts = (const uint32_t *)(ws + A.L.pts);
        const int32_t *labels = (const int32_t *)(ws + A.L.labels);
        // bounding box of the kept cluster (the maxima as minima of 255 - r, 255 - c: one kind of reduction)
        uint32_t rmin = 255, cmin = 255, rmaxc = 255, cmaxc = 255;
        for (int p = tid; p < N; p += TB) {
            const uint32_t v = pts[p], r = v & 255, c = (v >> 8) & 255;
            if (labels[p] != best) m0[r * A.w + c] = 0;
            else { rmin = min(rmin, r); cmin = min(cmin, c); rmaxc = min(rmaxc, 255u - r); cmaxc = min(cmaxc, 255u - c); }
        }
        const uint32_t a = wave_min_u32(rmin), b = wave_min_u32(cmin), c_ = wave_min_u32(rmaxc), d = wave_min_u32(cmaxc);
        if ((tid & 63) == 0) { atomicMin(&box[0], a); atomicMin(&box[1], b); atomicMin(&box[2], c_); atomicMin(&box[3], d); }
        __syncthreads();
        if (box[0] <= 255u - box[2]) {                       // (a kept cluster has points; an empty box keeps the whole image)
            wr0 = max(0, (int)box[0] - 4); wc0 = max(0, (int)box[1] - 4);
            wh = min(A.h, (int)(255u - box[2]) + 5) - wr0; ww = min(A.w, (int)(255u - box[3]) + 5) - wc0;
        }
        if (A.op_close) {
            // grey CLOSE with a 5x5 rectangle = separable max (dilate) then separable min (erode);
            // out-of-image samples are ignored
            close_pass<true, true>(m0, m1, A.w, wr0, wc0, wh, ww);      // dilate: rows then columns
            __syncthreads();
            close_pass<true, false>(m1, m0, A.w, wr0, wc0, wh, ww);
            __syncthreads();
            close_pass<false, true>(m0, m1, A.w, wr0, wc0, wh, ww);     // erode
            __syncthreads();
            close_pass<false, false>(m1, m0, A.w, wr0, wc0, wh, ww);
            __syncthreads();
        }
        copy_bytes(map, m0, hw);
    }
    __syncthreads();
    // centroid of the non-zero pixels of the final map
    unsigned long long cnt = 0, sr = 0, sc = 0;
    for (int i = tid; i < wh * ww; i += TB) {                // (outside the window the map is zero)
        const int r = wr0 + i / ww, c = wc0 + i % ww;
        if (m0[r * A.w + c]) { ++cnt; sr += (uint32_t)r; sc += (uint32_t)c; }
    }
    // (a thread sees at most hw / TB + 1 pixels: the wavefront's sums fit 32 bits)
    cnt = (unsigned long long)(uint32_t)wave_sum_i32((int)cnt);
    sr = (unsigned long long)(uint32_t)wave_sum_i32((int)sr);
    sc = (unsigned long long)(uint32_t)wave_sum_i32((int)sc);
    if ((tid & 63) == 0) { red[tid >> 6] = cnt; red[NW16 + (tid >> 6)] = sr; red[2 * NW16 + (tid >> 6)] = sc; }
    __syncthreads();
    if (tid == 0) {
        cnt = sr = sc = 0;
        for (int i = 0; i < NW16; ++i) { cnt += red[i]; sr += red[NW16 + i]; sc += red[2 * NW16 + i]; }
        if (cnt) {
            A.xy[2 * f] = (double)sc / (double)cnt;
            A.xy[2 * f + 1] = (double)sr / (double)cnt;
        } else {
            A.xy[2 * f] = A.xy[2 * f + 1] = __longlong_as_double(0x7FF8000000000000LL);
        }
        if (A.stats) {
            A.stats[4 * f] = N;
            A.stats[4 * f + 1] = clustered ? hdr[1] : 0;
            A.stats[4 * f + 2] = clustered ? hdr[2] : -1;
            A.stats[4 * f + 3] = (int)cnt;
        }
        hdr[11] = (int)(wall_clock64() - t0);
    }
}
__global__ __launch_bounds__(TB) void k_finish(TailArgs A) { finish_body(A); }

// --------------------------------------------------------------------------------------
// The round's kernels fused at the launch level (same workgroup, same map, LDS re-carved between the stages): two
// launches instead of six, i.e. four kernel boundaries (~4 us each on a stream) less per round.  Maps a stage cannot
// take (more than 8 192 points, cluster tables full) are left to the stand-alone kernels launched behind.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_tail_front(TailArgs A) {           // k_compact -> k_core -> k_prim_lvl
    // (s_setprio 3 for the tail's wavefronts -- a chain of dependent steps sharing its SIMDs with other streams' network wavefronts --
    // was measured in round 4: no difference, 1.2315 / 1.2354 ms per pipelined step; the chain waits for its own latencies, not for
    // issue slots.  The knob is gone.)
    compact_body(A);
    __syncthreads();
    core_body(A);
    __syncthreads();
    prim_lvl_body(A);
}

__global__ __launch_bounds__(TB) void k_tail_back(TailArgs A, int cap_clusters, int cap_clusters_huge) {   // k_sort -> k_tree_par -> k_finish
    int32_t *hdr = (int32_t *)(A.ws + (size_t)(A.slot0 + (int)blockIdx.x) * A.ws_stride + A.L.hdr);
    sort_body(A);
    __syncthreads();
    tree_par_body(A, cap_clusters, cap_clusters_huge);
    __syncthreads();
    if (hdr[3] && !hdr[23]) return;                                        // hierarchy not done here: k_tree, then k_finish
    finish_body(A);
    __syncthreads();
    if (threadIdx.x == 0) hdr[30] = 1;
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
extern "C" int svc_threshold_u8(SvcHandle *h, uint8_t *maps, size_t n_bytes, int t, void *stream) {
    if (!h || (n_bytes > 0 && !maps)) { svc_set_error("svc_threshold_u8: invalid argument"); return SVC_E_INVALID; }
    if (n_bytes == 0) return SVC_OK;
    SVC_HIP(hipSetDevice(h->device));
    ProfScope ps(h, SVC_K_THRESHOLD, (hipStream_t)stream);
    k_threshold<<<(unsigned)((n_bytes + 4095) / 4096), 256, 0, (hipStream_t)stream>>>(maps, n_bytes, t);
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

extern "C" int svc_iou_i32(const int32_t *a, const int32_t *b, size_t n, double *out, void *stream) {
    if (!a || !b || !out) { svc_set_error("svc_iou_i32: invalid argument"); return SVC_E_INVALID; }
    if (n == 0) return SVC_OK;
    k_iou<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const int4 *)a, (const int4 *)b, n, out);
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

// OpenCV INTER_LINEAR tables for one axis with an explicit scale (see svc_net.hip for the frame version)
static void cv_tab_axis(int src, int dst, double scale, bool horizontal, int *ofs, int *a, int *xmax_out) {
    int xmax = dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (horizontal) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s + 1 >= src) {
                xmax = std::min(xmax, d);
                if (s >= src - 1) { f = 0.f; s = src - 1; }
            }
        }
        ofs[d] = s;
        a[2 * d] = (int)std::min(std::max(lrintf((1.f - f) * 2048.f), -32768L), 32767L);
        a[2 * d + 1] = (int)std::min(std::max(lrintf(f * 2048.f), -32768L), 32767L);
    }
    if (xmax_out) *xmax_out = xmax;
}

static int upload_map_tab(DevBuf &buf, int h, int w, int oh, int ow, double sy, double sx) {
    std::vector<int> tab(3 * ow + 3 * oh + 1);
    int xmax = ow;
    cv_tab_axis(w, ow, sx, true, tab.data(), tab.data() + ow, &xmax);
    cv_tab_axis(h, oh, sy, false, tab.data() + 3 * ow, tab.data() + 3 * ow + oh, nullptr);
    tab[3 * ow + 3 * oh] = xmax;
    int rc = buf.ensure(tab.size() * 4);
    if (rc) return rc;
    SVC_HIP(hipMemcpy(buf.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    return SVC_OK;
}

#define RC_TAIL(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int ensure_ring(SvcHandle *h) {
    if (h->tail_n_offsets) return SVC_OK;
    std::vector<uint32_t> v;
    for (int dr = -RING_R; dr <= RING_R; ++dr)
        for (int dc = -RING_R; dc <= RING_R; ++dc) {
            int d2 = dr * dr + dc * dc;
            if (d2 == 0 || d2 > RING_R * RING_R) continue;
            v.push_back((uint32_t)(dr + 128) | ((uint32_t)(dc + 128) << 8) | ((uint32_t)d2 << 16));
        }
    std::stable_sort(v.begin(), v.end(), [](uint32_t x, uint32_t y) { return (x >> 16) < (y >> 16); });
    int rc = h->tail_offsets.ensure(v.size() * 4);
    if (rc) return rc;
    SVC_HIP(hipMemcpy(h->tail_offsets.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    h->tail_offsets_host = v;
    h->tail_n_offsets = (int)v.size();
    h->tail_n_offsets1 = 0;
    for (uint32_t x : v) h->tail_n_offsets1 += (int)((x >> 16) <= RING_R1 * RING_R1);
    std::vector<uint16_t> cnt(RING_R * RING_R + 1, 0);       // offsets with d2 <= m, for k_prim_lvl's discs
    for (uint32_t x : v) ++cnt[x >> 16];
    for (size_t m = 1; m < cnt.size(); ++m) cnt[m] = (uint16_t)(cnt[m] + cnt[m - 1]);
    if ((rc = h->tail_ring_cnt.ensure(cnt.size() * 2))) return rc;
    SVC_HIP(hipMemcpy(h->tail_ring_cnt.p, cnt.data(), cnt.size() * 2, hipMemcpyHostToDevice));
    return SVC_OK;
}

// linear offsets of the ring table for maps of this width (rebuilt only when the width changes)
static int ensure_ring_delta(SvcHandle *h, int width, const int32_t **out) {
    auto it = h->tail_delta.find(width);
    if (it != h->tail_delta.end()) { *out = (const int32_t *)it->second.p; return SVC_OK; }
    std::vector<int32_t> d(h->tail_offsets_host.size());
    for (size_t i = 0; i < d.size(); ++i) {
        const uint32_t o = h->tail_offsets_host[i];
        d[i] = ((int)(o & 255) - 128) * width + ((int)((o >> 8) & 255) - 128);
    }
    DevBuf buf;                     // a fresh buffer: nothing in flight can be reading it during the upload
    int rc = buf.ensure(d.size() * 4);
    if (rc) return rc;
    SVC_HIP(hipMemcpy(buf.p, d.data(), d.size() * 4, hipMemcpyHostToDevice));
    h->tail_delta.emplace(width, buf);
    *out = (const int32_t *)buf.p;
    return SVC_OK;
}

// The round of every map of a call: a map blended from its predecessor runs one round after it.  A HELD map (SVC_MAP_HELD)
// is not processed at all (depth 255) -- it is final already (the tail of a chain carried over from an earlier call) or
// left for a later call -- and a map blended from a held one runs in round 0, after the blend (listed in blend0).
static int plan_rounds(const uint8_t *flags, int n, std::vector<uint8_t> &depth, std::vector<uint16_t> &blend0, int &maxd) {
    const int HELD = 255;
    depth.assign(n, 0);
    blend0.clear();
    maxd = 0;
    for (int i = 0; i < n; ++i) {
        const int fl = flags ? flags[i] : 0, fp = (flags && i > 0) ? flags[i - 1] : 0;
        if (fl & SVC_MAP_HELD) { depth[i] = HELD; continue; }
        if (fp & SVC_BLEND_NEXT) {
            if (depth[i - 1] == HELD) { depth[i] = 0; blend0.push_back((uint16_t)i); }
            else {
                if (depth[i - 1] >= 254) { svc_set_error("svc_cluster_center: blend chain longer than 254"); return SVC_E_INVALID; }
                depth[i] = depth[i - 1] + 1;
                maxd = std::max(maxd, (int)depth[i]);
            }
        }
    }
    return SVC_OK;
}

// Test door (no GPU needed): the plan svc_cluster_center makes for a flag array.  round_out[i] = round of map i, -1 = held;
// blend0_out[i] = 1 when map i first takes a blend from a held predecessor.  Returns the number of rounds.
extern "C" int svc_debug_round_plan(const uint8_t *flags_host, int n, int32_t *round_out, int32_t *blend0_out) {
    if (n < 0 || (n > 0 && (!round_out || !blend0_out))) { svc_set_error("svc_debug_round_plan: invalid argument"); return SVC_E_INVALID; }
    std::vector<uint8_t> depth;
    std::vector<uint16_t> blend0;
    int maxd = 0;
    const int rc = plan_rounds(flags_host, n, depth, blend0, maxd);
    if (rc) return rc;
    bool any = false;
    for (int i = 0; i < n; ++i) { round_out[i] = depth[i] == 255 ? -1 : depth[i]; blend0_out[i] = 0; any = any || depth[i] != 255; }
    for (uint16_t i : blend0) blend0_out[i] = 1;
    return any ? maxd + 1 : 0;
}

static int cluster_center_impl(SvcHandle *h, uint8_t *maps, int n, int height, int width,
                                  const uint8_t *blend_flags_host, const SvcParams *params, double *xy,
                                  int32_t *stats, void *stream);
// SVC_HOST_TIMING=1: the call's host-side duration goes to stderr (every 100 calls; every call above 2 ms) -- a call is
// supposed to enqueue and return; a workspace re-allocation or a full upload ring shows up here
static std::atomic<long long> g_cc_ns{0}, g_cc_calls{0}, g_cc_max{0};
extern "C" int svc_cluster_center(SvcHandle *h, uint8_t *maps, int n, int height, int width,
                                  const uint8_t *blend_flags_host, const SvcParams *params, double *xy,
                                  int32_t *stats, void *stream) {
    static const bool timing = getenv("SVC_HOST_TIMING") != nullptr;
    if (!timing) return cluster_center_impl(h, maps, n, height, width, blend_flags_host, params, xy, stats, stream);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = cluster_center_impl(h, maps, n, height, width, blend_flags_host, params, xy, stats, stream);
    const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    g_cc_ns += ns; const long long c = ++g_cc_calls;
    if (ns > g_cc_max) g_cc_max = ns;
    if (ns > 2000000) fprintf(stderr, "svc_cluster_center call %lld: %.1f ms (n=%d)\n", c, ns / 1e6, n);
    if (c % 100 == 0) fprintf(stderr, "svc_cluster_center host time: %lld calls, mean %.1f us, max %.1f us\n", c, g_cc_ns / 1e3 / c, g_cc_max / 1e3);
    return rc;
}
static int cluster_center_impl(SvcHandle *h, uint8_t *maps, int n, int height, int width,
                                  const uint8_t *blend_flags_host, const SvcParams *params, double *xy,
                                  int32_t *stats, void *stream) {
    if (!h || !params || n < 0 || (n > 0 && (!maps || !xy)) || height < 1 || width < 1) {     // n = 0: a no-op, null buffers allowed
        svc_set_error("svc_cluster_center: invalid argument");
        return SVC_E_INVALID;
    }
    if (height > 255 || width > 255 || height * width > 65535) {
        svc_set_error("svc_cluster_center: map %dx%d exceeds the supported 255x255", height, width);
        return SVC_E_INVALID;
    }
    if (params->struct_size != sizeof(SvcParams)) {
        svc_set_error("svc_cluster_center: SvcParams.struct_size is %u, this library expects %zu (ABI %d): rebuild the binding "
                      "against include/svc.h", params->struct_size, sizeof(SvcParams), SVC_ABI_VERSION);
        return SVC_E_INVALID;
    }
    if (params->hdbscan_min < 2) { svc_set_error("svc_cluster_center: hdbscan_min must be >= 2"); return SVC_E_INVALID; }
    if (params->resize_factor < 1 || params->resize_factor > 16) { svc_set_error("svc_cluster_center: resize_factor must be an integer in 1..16"); return SVC_E_INVALID; }
    if (n == 0) return SVC_OK;
    SVC_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_ring(h);
    if (rc) return rc;
    // best settings: the cluster filter runs on maps shrunk by resize_factor (cvRound sizes, scale = factor)
    const int factor = params->clust_filt ? params->resize_factor : 1;
    const int full_h = height, full_w = width;
    uint8_t *full_maps = maps;
    const int *rs_down = nullptr, *rs_up = nullptr;
    if (factor > 1) {
        height = (int)lrint((double)full_h * (1.0 / factor));
        width = (int)lrint((double)full_w * (1.0 / factor));
        if (height < 1 || width < 1) { svc_set_error("svc_cluster_center: map too small for resize_factor"); return SVC_E_INVALID; }
        auto key = std::make_tuple(full_h, full_w, factor);
        auto it = h->rs_tabs.find(key);
        if (it == h->rs_tabs.end()) {       // tables are keyed by shape and never rewritten (earlier calls may still read theirs)
            std::pair<DevBuf, DevBuf> t;
            if ((rc = upload_map_tab(t.first, full_h, full_w, height, width, (double)factor, (double)factor))) return rc;
            if ((rc = upload_map_tab(t.second, height, width, full_h, full_w, (double)height / full_h, (double)width / full_w))) return rc;
            it = h->rs_tabs.emplace(key, t).first;
        }
        rs_down = (const int *)it->second.first.p;
        rs_up = (const int *)it->second.second.p;
        if ((rc = h->rs_maps.ensure((size_t)n * height * width))) return rc;
        maps = (uint8_t *)h->rs_maps.p;
    }
    const int cap = height * width;
    const int mc = hdb::max_clusters(cap, params->hdbscan_min);
    FrameWS L = make_layout(cap, mc);
    const int HELD = 255;
    std::vector<uint8_t> depth;
    std::vector<uint16_t> blend0;                       // round-0 maps that take a blend from a held predecessor first
    int maxd = 0;
    if ((rc = plan_rounds(blend_flags_host, n, depth, blend0, maxd))) return rc;
    if (n > DEPTH_SLOT) { svc_set_error("svc_cluster_center: more than %d maps per call", DEPTH_SLOT); return SVC_E_INVALID; }
    // A round's kernels are launched over the maps of THAT round only: the list of all maps sorted by round (stable),
    // one slice per round.  (One workgroup per map of the call in every round -- 31 of 32 exiting at once in the
    // follower rounds -- still has each of them claim a whole CU's wave slots / up to 154 KB of LDS before it can exit:
    // in the pipelined run those claims keep the network's workgroups off the CUs.)
    std::vector<uint16_t> order(2 * (size_t)n, 0);      // [0, n): maps sorted by round; [n, n + blend0.size()): the round-0 blend targets
    std::vector<int> round_start(maxd + 2, 0);
    for (int i = 0; i < n; ++i) if (depth[i] != HELD) ++round_start[depth[i] + 1];
    for (int r = 0; r <= maxd; ++r) round_start[r + 1] += round_start[r];
    {
        std::vector<int> fill(round_start.begin(), round_start.end() - 1);
        for (int i = 0; i < n; ++i) if (depth[i] != HELD) order[fill[depth[i]]++] = (uint16_t)i;
    }
    for (size_t i = 0; i < blend0.size(); ++i) order[n + i] = blend0[i];
    // workspace: the round lists (a ring of 8 slots, fixed size) and behind them one slot per map that takes part in a
    // round (a streamed call spans up to twice as many maps as it processes: the held ones need none); grown in steps of
    // 32 slots -- a re-allocation is a hipFree, i.e. a device-wide synchronisation, and used to happen whenever a call
    // was one map longer than any before it
    const size_t lists_bytes = 8 * 4 * (size_t)DEPTH_SLOT;
    const int n_proc = round_start[maxd + 1];
    if ((rc = h->tail_ws.ensure(lists_bytes + (size_t)L.total * std::max<size_t>(32, ((size_t)n_proc + 31) / 32 * 32)))) return rc;
    // the lists travel through a small ring of pinned host slots so that the upload is asynchronous (up to 8 calls may
    // be in flight on the stream before a slot is reused)
    if (!h->depth_pinned) SVC_HIP(hipHostMalloc((void **)&h->depth_pinned, 8 * 4 * (size_t)DEPTH_SLOT, hipHostMallocDefault));
    const int slot = h->depth_slot++ & 7;
    if (h->depth_ev[slot]) SVC_HIP(hipEventSynchronize(h->depth_ev[slot]));      // the upload that last used this slot has run
    else SVC_HIP(hipEventCreateWithFlags(&h->depth_ev[slot], hipEventDisableTiming));
    uint16_t *order_dev = (uint16_t *)((uint8_t *)h->tail_ws.p + (size_t)slot * 4 * DEPTH_SLOT);
    memcpy(h->depth_pinned + (size_t)slot * 4 * DEPTH_SLOT, order.data(), (size_t)n * 4);
    SVC_HIP(hipMemcpyAsync(order_dev, h->depth_pinned + (size_t)slot * 4 * DEPTH_SLOT, (size_t)n * 4, hipMemcpyHostToDevice, s));
    SVC_HIP(hipEventRecord(h->depth_ev[slot], s));
    h->tail_frames = n; h->tail_h = height; h->tail_w = width; h->tail_frame_stride = L.total;
    h->tail_slot_of.assign((size_t)n, -1);                       // (svc_debug_cluster_state: map -> workspace slot)
    for (int i = 0; i < n_proc; ++i) h->tail_slot_of[order[i]] = i;
    TailArgs A;
    A.maps = maps; A.ws = (uint8_t *)h->tail_ws.p + lists_bytes; A.ws_stride = L.total; A.order = order_dev; A.slot0 = 0;
    A.n = n; A.h = height; A.w = width; A.dW = make_fdiv(width);
    A.mcs = params->hdbscan_min; A.min_samples = params->hdbscan_min_samples; A.select_sum = params->select_sum;
    A.op_close = params->op_close; A.clust_filt = params->clust_filt;
    A.prim_pt = h->prim_pt;
    const int32_t *ring_delta = nullptr;
    RC_TAIL(ensure_ring_delta(h, width, &ring_delta));
    A.ring = (const uint32_t *)h->tail_offsets.p; A.n_ring = h->tail_n_offsets; A.n_ring1 = h->tail_n_offsets1;
    A.ring_delta = ring_delta;
    A.ring_cnt = (const uint16_t *)h->tail_ring_cnt.p; A.prim_lvl = h->prim_lvl;
    A.xy = xy; A.stats = stats; A.L = L;
    const int hw = height * width;
    const size_t lds_core = (size_t)(hw + 15) / 16 * 16 + (size_t)h->tail_n_offsets * 4 + (size_t)h->tail_n_offsets1 * 4;
    const size_t lds_prim = 2 * NW16 * 16 + (size_t)hw * 2;
    const size_t lds_fin = 2 * ((size_t)(hw + 15) / 16 * 16);
    if (h->lds_attr_done.insert((const void *)k_core).second) {      // per handle = per device (the attribute is per device)
        SVC_HIP(hipFuncSetAttribute((const void *)k_core, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_pt<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_pt<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_pt<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_lvl, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));      // (255 x 255 maps: 162 896 bytes with eight workers; static LDS 136)
        SVC_HIP(hipFuncSetAttribute((const void *)k_prim_lvl_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_tree, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_tree_par, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_tail_front, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));      // (255 x 255 maps: 162 896 bytes with eight workers; static LDS 136)
        SVC_HIP(hipFuncSetAttribute((const void *)k_tail_back, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        SVC_HIP(hipFuncSetAttribute((const void *)k_sort, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_BYTES));
    }
    for (int r = 0; r <= maxd; ++r) {
        const int m = round_start[r + 1] - round_start[r];   // maps of this round
        const uint16_t *ord = order_dev + round_start[r];
        A.order = ord; A.slot0 = round_start[r];
        if (r == 0 && !blend0.empty()) {                     // blends from held (already final) predecessors
            k_blend<<<dim3(8, (unsigned)blend0.size()), 256, 0, s>>>(full_maps, order_dev + n, full_h * full_w);
            SVC_CHECK_LAUNCH();
        }
        if (m == 0) continue;                                // every map of the call is held
        if (r > 0) {
            k_blend<<<dim3(8, m), 256, 0, s>>>(full_maps, ord, full_h * full_w);
            SVC_CHECK_LAUNCH();
        }
        if (factor > 1) {
            k_map_resize<<<dim3(8, m), 256, 0, s>>>(full_maps, maps, rs_down, ord, full_h, full_w, height, width);
            SVC_CHECK_LAUNCH();
        }
        int cap_cl = 0, cap_cl_huge = 0;
        const size_t lds_tp = tp_lds_bytes(hw, params->hdbscan_min, &cap_cl, &cap_cl_huge);
        const bool tree_fallback = !h->tree_par || hw > TP_CAP_HUGE || hdb::max_clusters(std::min(hw, TP_CAP_BIG), params->hdbscan_min) > cap_cl ||
                                   (hw > TP_CAP_BIG && hdb::max_clusters(std::min(hw, TP_CAP_HUGE), params->hdbscan_min) > cap_cl_huge);
        if (params->clust_filt && h->tail_merge && h->prim_lvl && h->tree_par) {
            // two fused launches per round (k_tail_front, k_tail_back) + the stand-alone kernels for what they leave
            const size_t lds_front = std::max({(size_t)(hw + 15) / 16 * 16, lds_core, lvl_lds_bytes(height, width, h->tail_n_offsets)});
            const size_t lds_back = std::max({(size_t)SORT_LDS_BYTES, lds_tp, lds_fin});
            {
                ProfScope ps(h, SVC_K_PRIM, s);
                k_tail_front<<<m, TB, lds_front, s>>>(A);
                SVC_CHECK_LAUNCH();
                if (hw > LVL_CAP) { k_prim_lvl_big<<<m, TB, lvl_big_lds_bytes(height, width, h->tail_n_offsets), s>>>(A); SVC_CHECK_LAUNCH(); }
            }
            ProfScope ps(h, SVC_K_FINISH, s);
            k_tail_back<<<m, TB, lds_back, s>>>(A, cap_cl, cap_cl_huge);
            SVC_CHECK_LAUNCH();
            if (tree_fallback) {
                k_tree<<<m, 64, FIN_LDS_BYTES, s>>>(A);
                SVC_CHECK_LAUNCH();
                k_finish<<<m, TB, lds_fin, s>>>(A);
                SVC_CHECK_LAUNCH();
            }
        } else {
        {
            ProfScope ps(h, SVC_K_COMPACT, s);
            k_compact<<<m, TB, (size_t)(hw + 15) / 16 * 16, s>>>(A);
            SVC_CHECK_LAUNCH();
        }
        if (params->clust_filt) {
            {
                ProfScope ps(h, SVC_K_CORE, s);
                k_core<<<m, TB, lds_core, s>>>(A);
                SVC_CHECK_LAUNCH();
            }
            ProfScope ps(h, SVC_K_PRIM, s);
            // maps of up to LVL_CAP points: the level-bucketed Prim; larger ones (or all, SVC_PRIM_LVL=0): one node per step
            const int n_min = h->prim_lvl ? LVL_CAP : 0;
            if (h->prim_lvl) {
                k_prim_lvl<<<m, TB, lvl_lds_bytes(height, width, h->tail_n_offsets), s>>>(A);
                SVC_CHECK_LAUNCH();
            } else {
                if (h->prim_pt <= 2) { k_prim_pt<2><<<m, TB, lds_prim, s>>>(A, n_min); SVC_CHECK_LAUNCH(); }
                if (h->prim_pt <= 4 && (hw > 2 * TB || h->prim_pt > 2)) { k_prim_pt<4><<<m, TB, lds_prim, s>>>(A, n_min); SVC_CHECK_LAUNCH(); }
                if (hw > 4 * TB || h->prim_pt > 4) { k_prim_pt<8><<<m, TB, lds_prim, s>>>(A, n_min); SVC_CHECK_LAUNCH(); }
            }
            if (hw > LVL_CAP && h->prim_lvl) { k_prim_lvl_big<<<m, TB, lvl_big_lds_bytes(height, width, h->tail_n_offsets), s>>>(A); SVC_CHECK_LAUNCH(); }
            else if (hw > 8 * TB) { k_prim_big<<<m, TB, lds_prim, s>>>(A, n_min); SVC_CHECK_LAUNCH(); }
        }
        {
            ProfScope ps(h, SVC_K_FINISH, s);
            if (params->clust_filt) {
                k_sort<<<m, TB, SORT_LDS_BYTES, s>>>(A);
                SVC_CHECK_LAUNCH();
                if (h->tree_par) {
                    k_tree_par<<<m, TB, lds_tp, s>>>(A, cap_cl, cap_cl_huge);
                    SVC_CHECK_LAUNCH();
                }
                // the serial builder: maps the parallel one does not hold in LDS (more points or clusters), or all (SVC_TREE_PAR=0)
                if (tree_fallback) {
                    k_tree<<<m, 64, FIN_LDS_BYTES, s>>>(A);
                    SVC_CHECK_LAUNCH();
                }
            }
            k_finish<<<m, TB, lds_fin, s>>>(A);
            SVC_CHECK_LAUNCH();
        }
        }
        if (factor > 1) {
            k_map_resize<<<dim3(16, m), 256, 0, s>>>(maps, full_maps, rs_up, ord, height, width, full_h, full_w);
            SVC_CHECK_LAUNCH();
        }
    }
    if (!params->com_km) {                 // centre = arg-max pixel of the final full-size map (held maps keep what the caller has)
        k_centre_argmax<<<n, 256, 0, s>>>(full_maps, full_h, full_w, make_fdiv(full_w), xy);
        SVC_CHECK_LAUNCH();
    } else if (params->resize_factor > 1) {       // centre of the nearest-neighbour shrunk final map (also when clust_filt is off)
        const int f2 = params->resize_factor;
        k_centre_nearest<<<n, 256, 0, s>>>(full_maps, full_h, full_w, (int)lrint((double)full_h * (1.0 / f2)),
                                           (int)lrint((double)full_w * (1.0 / f2)), f2, xy);
        SVC_CHECK_LAUNCH();
    }
    return SVC_OK;
}

extern "C" int svc_debug_argsort_u32(SvcHandle *h, const uint32_t *keys_host, int n, int32_t *order_host) {
    if (!h || !keys_host || !order_host || n < 1 || n > 65535) { svc_set_error("svc_debug_argsort_u32: invalid argument"); return SVC_E_INVALID; }
    SVC_HIP(hipSetDevice(h->device));
    DevBuf keys, out, scratch;
    int rc;
    if ((rc = keys.ensure((size_t)n * 4)) || (rc = out.ensure((size_t)n * 2)) || (rc = scratch.ensure((size_t)n * 16 + 128))) return rc;
    SVC_HIP(hipMemcpy(keys.p, keys_host, (size_t)n * 4, hipMemcpyHostToDevice));
    if (h->lds_attr_done.insert((const void *)k_argsort_test).second)
        SVC_HIP(hipFuncSetAttribute((const void *)k_argsort_test, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_BYTES));
    k_argsort_test<<<1, TB, SORT_LDS_BYTES, 0>>>((const uint32_t *)keys.p, n, (uint16_t *)out.p, (uint8_t *)scratch.p);
    SVC_CHECK_LAUNCH();
    SVC_HIP(hipDeviceSynchronize());
    std::vector<uint16_t> o(n);
    SVC_HIP(hipMemcpy(o.data(), out.p, (size_t)n * 2, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) order_host[i] = o[i];
    keys.release(); out.release(); scratch.release();
    return n;
}

extern "C" int svc_debug_cluster_state(SvcHandle *h, int frame, int cap, uint32_t *pts_host, uint32_t *core_host,
                                       uint32_t *mst_host, int32_t *labels_host, int32_t *hdr_host) {
    if (!h || !h->tail_ws.p || frame < 0 || frame >= h->tail_frames || h->tail_slot_of[frame] < 0) {
        svc_set_error("svc_debug_cluster_state: no such frame (or a held map: the last call did not process it)");
        return SVC_E_INVALID;
    }
    SVC_HIP(hipSetDevice(h->device));
    SVC_HIP(hipDeviceSynchronize());
    const int fcap = h->tail_h * h->tail_w;
    FrameWS L = make_layout(fcap, 1);      // point-array offsets do not depend on the cluster capacity
    const uint8_t *ws = (const uint8_t *)h->tail_ws.p + 8 * 4 * (size_t)DEPTH_SLOT + (size_t)h->tail_slot_of[frame] * h->tail_frame_stride;
    int32_t hdr[32];
    SVC_HIP(hipMemcpy(hdr, ws + L.hdr, sizeof hdr, hipMemcpyDeviceToHost));
    const int N = hdr[0];
    const int m = std::min(N, cap);
    if (hdr_host) memcpy(hdr_host, hdr, sizeof hdr);
    if (pts_host) SVC_HIP(hipMemcpy(pts_host, ws + L.pts, (size_t)m * 4, hipMemcpyDeviceToHost));
    if (hdr[3]) {
        if (core_host) SVC_HIP(hipMemcpy(core_host, ws + L.core, (size_t)m * 4, hipMemcpyDeviceToHost));
        if (labels_host) SVC_HIP(hipMemcpy(labels_host, ws + L.labels, (size_t)m * 4, hipMemcpyDeviceToHost));
        if (mst_host && m > 1) {
            std::vector<hdb::Edge> e(m - 1);
            SVC_HIP(hipMemcpy(e.data(), ws + L.mst, (size_t)(m - 1) * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < m - 1; ++i) { mst_host[3 * i] = e[i].a; mst_host[3 * i + 1] = e[i].b; mst_host[3 * i + 2] = e[i].w; }
        }
    } else {
        if (core_host) memset(core_host, 0, (size_t)m * 4);
        if (labels_host) for (int i = 0; i < m; ++i) labels_host[i] = -1;
    }
    return N;
}
