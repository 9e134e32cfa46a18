// svc_net.hip — UNISAL static (SALICON) saliency forward for gfx950, plus the ingest
// down-scale.  Hand-written HIP: NHWC fp32 activations, BN folded into the weights,
// 1x1 convolutions on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32),
// depthwise / resampling / softmax-quantise stages as coalesced float4 kernels.
//
// Reference semantics restated per kernel (paths relative to the reference tree):
//   k_cv_resize      cv2.resize(INTER_LINEAR)         smartVidCrop.py:333-335, :633-635
//   k_lanczos_norm   PIL LANCZOS + ToTensor + Normalize 3rd_party_libs/unisal/unisal/data.py:1281-1294
//   k_stem           conv_bn(3,32,stride 2)+ReLU6      unisal/models/MobileNetV2.py:10-15,124
//   k_pw<TN>         1x1 conv (+BN)(+ReLU6)(+residual) MobileNetV2.py:18-23,26-83; unisal/model.py:388-409
//   k_dw<S>          3x3 depthwise (+BN+ReLU6)         MobileNetV2.py:48-51,66-69
//   k_subsample      x[..., ::2, ::2] (omit-stride)    MobileNetV2.py:170-171
//   k_gauss_fill     Gaussian prior maps + concat      unisal/model.py:348-385,446-448
//   k_upsample2x     bilinear x2, align_corners=False  unisal/model.py:304-309,475-478
//   k_adapt          adaptation 1x1 64->1 (+bias)      unisal/model.py:481-483
//   k_smooth_down    nearest x8 + replicate pad 20 + 41x41 smoothing + bilinear to (h,w)
//                                                      unisal/model.py:485-495
//   k_quantise       log-softmax/exp/max/x255/u8 cast  unisal/utils.py:132-136, unisal/train.py:1267-1274
#include <math.h>

#include <algorithm>

#include "svc_internal.h"
#include "svc_x3.h"

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// XCD-aware workgroup order: hardware workgroup ids go round the eight XCDs, so with the logical tile of workgroup b taken as
// (b & 7) * (n / 8) + (b >> 3) XCD x works on the x-th eighth of the tiles -- four whole frames of a 32-frame pass, in every
// kernel alike, so a frame's activations are produced and consumed through one XCD's L2 (measured: a pass 1.236 -> 1.227 ms
// alone, 0.866 -> 0.860 ms with four passes sharing the chip, three alternations; grids whose x extent is no multiple of 8 keep
// the hardware order).
__device__ __forceinline__ unsigned xcd_bx() {
    const unsigned b = blockIdx.x, n = gridDim.x;
    return (n & 7) ? b : (b & 7) * (n >> 3) + (b >> 3);
}

// a += x * w on four channels as two packed fp32 FMAs (v_pk_fma_f32: two lanes' worth of fused multiply-adds per
// instruction; the same fma per element as fmaf, so the depthwise sums do not change)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma4(float4 &a, const float4 x, const float4 w) {
    const f32x2 lo = __builtin_elementwise_fma((f32x2){x.x, x.y}, (f32x2){w.x, w.y}, (f32x2){a.x, a.y});
    const f32x2 hi = __builtin_elementwise_fma((f32x2){x.z, x.w}, (f32x2){w.z, w.w}, (f32x2){a.z, a.w});
    a = make_float4(lo.x, lo.y, hi.x, hi.y);
}

// --------------------------------------------------------------------------------------
// Split-bf16 operands (round 5): the 1x1-convolution GEMMs on the bf16 matrix pipe WITHOUT narrowing the arithmetic.
// x = hi + mid + lo, each a bf16: 8 + 8 + 8 significant bits = the 24 of an f32, so the split by truncation is EXACT
// (hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid has at most 8 bits left).  Every bf16 x bf16
// product is exact in f32 and v_mfma_f32_32x32x16_bf16 accumulates in f32, so  x . w = sum over the plane pairs;
// the three pairs mid.lo, lo.mid, lo.lo are <= 2^-24 of the product -- the size of the f32 rounding the fp32 MFMA
// makes per product -- and are left out ("bf16x6": 6 x 32 = 192 matrix-pipe cycles per 16 k against 8 x 64 = 512 on
// v_mfma_f32_32x32x2_f32; X3_NP = 9 keeps them).  Measured against float64 on the layers' shapes
// (tools/micro/bf16x3_gemm.hip, profiles/r05_bf16x3_micro.txt): rms error 1.14e-7 (x6) / 1.14e-7 (x9) / 1.08e-7 (fp32
// MFMA) of a unit-rms result on ReLU6-like inputs, 1.24 / 1.23 / 1.55e-7 on signed wide-range inputs.
// Small pairs are accumulated first.  Weights are split ONCE per handle (round-to-nearest planes, x3_weights below);
// activations are split in registers where a wave holds them anyway.
// Operand convention: element j (0..7) of the fragment of lane (r, hh) in the 16-deep step q is
//   k = 16 q + 8 (j >> 2) + 4 hh + (j & 3)
// i.e. the two float4 a lane of the fp32 kernels holds for the 8-deep steps 2q and 2q + 1 -- both operands use it, so
// the existing register / LDS layouts carry over (the k order inside a step is a consistent permutation).
// The helpers (x3_split, x3_load, x3_mma) live in svc_x3.h.
// --------------------------------------------------------------------------------------

// --------------------------------------------------------------------------------------
// error string
// --------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void svc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *svc_last_error(void) { return g_err; }
extern "C" int svc_abi_version(void) { return SVC_ABI_VERSION; }

// --------------------------------------------------------------------------------------
// ingest down-scale: OpenCV INTER_LINEAR on u8 (11-bit fixed-point weights)
// tab layout (int32): xofs[ow] | xa[ow][2] | yofs[oh] | ya[oh][2] | xmax
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cv_resize(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                   const int *__restrict__ tab, int n, int h, int w, int oh, int ow) {
    const int *xofs = tab, *xa = tab + ow, *yofs = tab + 3 * ow, *ya = tab + 3 * ow + oh;
    const int xmax = tab[3 * ow + 3 * oh];
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t total = (size_t)n * oh * ow;
    if (gid >= total) return;
    int ox = gid % ow;
    int oy = (gid / ow) % oh;
    int f = gid / ((size_t)ow * oh);
    int sy = yofs[oy];
    int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
    int b0 = ya[2 * oy], b1 = ya[2 * oy + 1];
    int sx = xofs[ox];
    int sx1 = min(sx + 1, w - 1);
    int a0 = xa[2 * ox], a1 = xa[2 * ox + 1];
    const uint8_t *r0 = in + ((size_t)f * h + y0) * w * 3;
    const uint8_t *r1 = in + ((size_t)f * h + y1) * w * 3;
    uint8_t *o = out + gid * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int h0, h1;
        if (ox < xmax) {
            h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        } else {
            h0 = r0[sx * 3 + c] * 2048;
            h1 = r1[sx * 3 + c] * 2048;
        }
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)min(max(v, 0), 255);
    }
}

static void cv_linear_tab(int src, int dst, bool horizontal, int *ofs, int *a, int *xmax_out) {
    double scale = (double)src / dst;
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
        long w0 = lrintf((1.f - f) * 2048.f), w1 = lrintf(f * 2048.f);
        a[2 * d] = (int)std::min(std::max(w0, -32768L), 32767L);
        a[2 * d + 1] = (int)std::min(std::max(w1, -32768L), 32767L);
    }
    if (xmax_out) *xmax_out = xmax;
}

extern "C" int svc_resize_frames_u8(SvcHandle *h, const uint8_t *frames, int n, int height, int width,
                                    uint8_t *out, int sh, int sw, void *stream) {
    if (!h || n < 0 || (n > 0 && (!frames || !out)) || height < 1 || width < 1 || sh < 1 || sw < 1) {     // n = 0: a no-op, null buffers allowed
        svc_set_error("svc_resize_frames_u8: invalid argument");
        return SVC_E_INVALID;
    }
    if (n == 0) return SVC_OK;
    SVC_HIP(hipSetDevice(h->device));
    auto key = std::make_tuple(height, width, sh, sw);
    auto it = h->cvtabs.find(key);
    if (it == h->cvtabs.end()) {
        std::vector<int> tab(3 * sw + 3 * sh + 1);
        int xmax = sw;
        cv_linear_tab(width, sw, true, tab.data(), tab.data() + sw, &xmax);
        cv_linear_tab(height, sh, false, tab.data() + 3 * sw, tab.data() + 3 * sw + sh, nullptr);
        tab[3 * sw + 3 * sh] = xmax;
        DevBuf buf;
        int rc = buf.ensure(tab.size() * 4);
        if (rc) return rc;
        SVC_HIP(hipMemcpy(buf.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        it = h->cvtabs.emplace(key, buf).first;
    }
    size_t total = (size_t)n * sh * sw;
    ProfScope ps(h, SVC_K_RESIZE, (hipStream_t)stream);
    k_cv_resize<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        frames, out, (const int *)it->second.p, n, height, width, sh, sw);
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

// --------------------------------------------------------------------------------------
// K0: Pillow LANCZOS (two 8-bit fixed-point passes) + /255 + normalise (via LUT) -> NHWC fp32
// --------------------------------------------------------------------------------------
#define LZ_PREC 22
// One workgroup = one frame x rows_per_block output rows.  The input rows it needs, the horizontal
// coefficients and the normalisation LUT are staged in LDS first (coalesced), so both passes run
// out of LDS.  LDS layout: in_s[in_cap][w*3] u8 | tile[tile_cap][NW*3] u8 | hk[NW][hks] i32 | lut[768] f32
__global__ __launch_bounds__(256) void k_lanczos_norm(
    const uint8_t *__restrict__ in, float *__restrict__ out, int h, int w, int NH, int NW,
    const int *__restrict__ hb, const int *__restrict__ hk, int hks, const int *__restrict__ vb,
    const int *__restrict__ vk, int vks, const float *__restrict__ lut, int rows_per_block, int tile_cap) {
    extern __shared__ uint8_t sm_lz[];
    const int in_bytes = (tile_cap * w * 3 + 15) / 16 * 16, tile_bytes = (tile_cap * NW * 3 + 15) / 16 * 16;
    uint8_t *in_s = sm_lz;
    uint8_t *tile = sm_lz + in_bytes;
    int *hk_s = (int *)(sm_lz + in_bytes + tile_bytes);
    float *lut_s = (float *)(hk_s + NW * hks);
    const int f = blockIdx.y;
    const int y0 = blockIdx.x * rows_per_block;
    const int y1 = min(NH, y0 + rows_per_block);
    const int r_lo = vb[2 * y0];
    const int r_hi = vb[2 * (y1 - 1)] + vb[2 * (y1 - 1) + 1];
    const int nr = min(r_hi - r_lo, tile_cap);
    const uint8_t *src = in + ((size_t)f * h + r_lo) * w * 3;
    for (int i = threadIdx.x; i < nr * w * 3; i += 256) in_s[i] = src[i];
    for (int i = threadIdx.x; i < NW * hks; i += 256) hk_s[i] = hk[i];
    for (int i = threadIdx.x; i < 768; i += 256) lut_s[i] = lut[i];
    __syncthreads();
    const int rowlen = NW * 3;
    for (int i = threadIdx.x; i < rowlen; i += 256) {           // horizontal pass, one output column per thread
        const int x = i / 3, c = i - 3 * x;
        const int xmin = hb[2 * x], cnt = hb[2 * x + 1];
        const int *k = hk_s + x * hks;
        const uint8_t *p = in_s + xmin * 3 + c;
        for (int r = 0; r < nr; ++r, p += w * 3) {
            int acc = 1 << (LZ_PREC - 1);
            for (int j = 0; j < cnt; ++j) acc += __mul24((int)p[j * 3], k[j]);      // |coefficient| < 2^23: exact, full-rate
            tile[r * rowlen + i] = (uint8_t)min(max(acc >> LZ_PREC, 0), 255);
        }
    }
    __syncthreads();
    float *dst = out + ((size_t)f * NH + y0) * rowlen;
    for (int i = threadIdx.x; i < rowlen; i += 256) {           // vertical pass + normalisation
        const int c = i - 3 * (i / 3);
        const float *lc = lut_s + c * 256;
        for (int y = y0; y < y1; ++y) {
            const int ymin = vb[2 * y], cnt = vb[2 * y + 1];
            const int *k = vk + y * vks;
            const uint8_t *p = tile + (ymin - r_lo) * rowlen + i;
            int acc = 1 << (LZ_PREC - 1);
            for (int j = 0; j < cnt; ++j) acc += __mul24((int)p[j * rowlen], k[j]);
            dst[(size_t)(y - y0) * rowlen + i] = lc[min(max(acc >> LZ_PREC, 0), 255)];
        }
    }
}

static double lz_sinc(double x) {
    if (x == 0.0) return 1.0;
    x *= M_PI;
    return sin(x) / x;
}
static double lz_filter(double x) { return (-3.0 <= x && x < 3.0) ? lz_sinc(x) * lz_sinc(x / 3.0) : 0.0; }

// Pillow's precompute_coeffs + normalize_coeffs_8bpc.  Identity table when sizes match
// (Pillow skips that pass).
static void lanczos_tab(int in_size, int out_size, std::vector<int> &bounds, std::vector<int> &coef, int &ksize) {
    bounds.assign(2 * out_size, 0);
    if (in_size == out_size) {
        ksize = 1;
        coef.assign(out_size, 1 << LZ_PREC);
        for (int i = 0; i < out_size; ++i) { bounds[2 * i] = i; bounds[2 * i + 1] = 1; }
        return;
    }
    double scale = (double)in_size / out_size, filterscale = std::max(scale, 1.0);
    double support = 3.0 * filterscale, ss = 1.0 / filterscale;
    ksize = (int)ceil(support) * 2 + 1;
    coef.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        double center = (xx + 0.5) * scale, ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            k[x] = lz_filter((x + xmin - center + 0.5) * ss);
            ww += k[x];
        }
        for (int x = 0; x < xmax; ++x) {
            double v = (ww != 0.0) ? k[x] / ww : k[x];
            coef[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << LZ_PREC)) : (int)(0.5 + v * (1 << LZ_PREC));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

// features.0 on the matrix cores: one workgroup = 8 x 16 output pixels.  The 17 x 33 x 3 input patch is read once
// (whole rows, coalesced) into LDS; every lane owns one output pixel and gathers its 27 taps (K padded to 32) from
// the patch as the B operand of sixteen 32x32x2 MFMAs against the transposed weights (A operand, four float4 per
// lane, loaded once).  With the operands this way round a lane ends with all 32 channels of its pixel in runs of
// four: bias, ReLU6 and 128 contiguous bytes per pixel as float4 stores.  k_stem (FMA form) issues 27 scalar loads
// and 27 LDS weight reads per 4 outputs and runs at a third of the HBM rate.
#define STEM_TH 8
#define STEM_TW 16
#define STEM_PRS 100      // floats per patch row: (2 * 16 + 1) pixels x 3 channels = 99
__global__ __launch_bounds__(256) void k_stem_mfma(const float *__restrict__ X, const float *__restrict__ Wtr,
                                                   const float *__restrict__ bias, float *__restrict__ Y, int NH,
                                                   int NW, int OH, int OW, int tiles_x, int tiles_y) {
    constexpr int PR = 2 * STEM_TH + 1, PCW = 2 * STEM_TW + 1;
    __shared__ float Pin[PR * STEM_PRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int bid = xcd_bx();
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, f = bid / tiles_y;
    const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW, py0 = 2 * oy0 - 1, px0 = 2 * ox0 - 1;
    const float *xin = X + (size_t)f * NH * NW * 3;
    for (int idx = tid; idx < PR * PCW * 3; idx += 256) {
        const int pr = idx / (PCW * 3), rem = idx - pr * (PCW * 3), pc = rem / 3;
        const int y = py0 + pr, x = px0 + pc;
        float v = 0.f;                                       // the conv's zero padding and the ragged right / bottom edge
        if ((unsigned)y < (unsigned)NH && (unsigned)x < (unsigned)NW) v = xin[((long long)y * NW + px0) * 3 + rem];
        Pin[pr * STEM_PRS + rem] = v;
    }
    // A operand: weights of output channel r, taps 4hh + 8q .. + 3
    float4 wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[q] = *(const float4 *)(Wtr + r * 32 + 8 * q + 4 * hh);
    __syncthreads();
    // B operand: this lane's pixel = 32 * wave + r of the tile
    const int p = 32 * wave + r, py = p / STEM_TW, px = p - py * STEM_TW;
    const float *src = Pin + (2 * py) * STEM_PRS + (2 * px) * 3;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 8 * q + 4 * hh + e;                // tap (ky, kx, ci) = (k / 9, (k % 9) / 3, k % 3): offset k + 91 * ky
            const int ky = (k * 57) >> 9;
            t[e] = k < 27 ? src[k + (STEM_PRS - 9) * ky] : 0.f;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].x, t[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].y, t[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].z, t[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].w, t[3], acc, 0, 0, 0);
    }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= OH || ox >= OW) return;
    float *yp = Y + (((size_t)f * OH + oy) * OW + ox) * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 8 * g + 4 * hh;
        const float4 b = *(const float4 *)(bias + c);
        float4 v;
        v.x = fminf(fmaxf(acc[4 * g] + b.x, 0.f), 6.f);
        v.y = fminf(fmaxf(acc[4 * g + 1] + b.y, 0.f), 6.f);
        v.z = fminf(fmaxf(acc[4 * g + 2] + b.z, 0.f), 6.f);
        v.w = fminf(fmaxf(acc[4 * g + 3] + b.w, 0.f), 6.f);
        *(float4 *)(yp + c) = v;
    }
}

// --------------------------------------------------------------------------------------
// The front of the network in one kernel: Pillow LANCZOS + normalise (k_lanczos_norm) -> features.0 (k_stem_mfma) ->
// features.1 (3x3 depthwise + ReLU6 -> 1x1 project 32 -> 16; the t = 1 case of k_irb) for a tile of 8 x 16 pixels of
// features.1's output.  As three kernels the 1.3 MB normalised input and the 3.4 MB stem output of every frame are
// written to memory and read back (46 + 45 + 69 us per 32 frames, the stem and block 1 at 2.4-3.3 TB/s); here the
// only traffic is the u8 frame in and the 16-channel tensor out.  Per tile:
//   source patch (u8) -> horizontal pass -> vertical pass + LUT -> 21 x 37 x 3 input patch          (integer, in LDS)
//   180 halo pixels of features.0 = 6 MFMA tiles of 32 pixels, K = 27 taps padded to 32             (S, in LDS)
//   depthwise 3x3 on the 128 pixels, one thread = one pixel x 4 channels                            (D, in LDS)
//   project: one 32-pixel tile per wave, the k range in two halves summed afterwards                 (16 channels out)
// Every stage keeps the operation order of the kernel it replaces (the resampling is integer arithmetic; the MFMA
// sequences, the tap order of the depthwise sum and the two-way k split of block 1's project are the same), so the
// result is bit-identical to the three-kernel path.
// Measured on MI355X, 32 frames (6656 workgroups): 116 us with three workgroups per CU (rounds 2-3), 100 us with four
// (round 4: the LDS layout below), against 160 for the three kernels.  The first form took 154: a tile's time was mostly
// round trips to memory in series (table look-ups for the patch bounds, then the source bytes, then per-row coefficient loads
// inside the vertical pass, 17 weight float4 per thread).  Now the patch bounds arrive with the kernel arguments, the source
// patch (aligned 32-bit words), coefficients and LUT are requested before the first wait, each layer's weights one phase
// before their use, and the passes read coefficients from registers / LDS.  Phase shares of the three-per-CU form (skipping
// one phase at a time): features.0 29 us (six 32-pixel tiles on four waves), vertical pass 17, depthwise 14, project 14 (16 of
// the 32 MFMA columns are padding), horizontal pass 10, source words 6, workgroup start-up 23.  A persistent variant (one
// workgroup walking a run of tiles with the next tile's loads in flight) was slower (143 us): fewer independent workgroups
// hide less.
// --------------------------------------------------------------------------------------
#define FR_TH 8
#define FR_TW 16
#define FR_HH (FR_TH + 2)           // stem rows of a tile incl. the depthwise halo
#define FR_HW (FR_TW + 2)
#define FR_NPX (FR_HH * FR_HW)      // 180
#define FR_PR (2 * FR_HH + 1)       // 21 input rows
#define FR_PC (2 * FR_HW + 1)       // 37 input columns
#define FR_PRS 112                  // floats (bytes for the u8 intermediate) per patch row: 37 x 3 = 111
#define FR_ES 36                    // floats per pixel of S / D (32 channels + 4: conflict-free float4 rows)

#define FR_MAXT 32                  // tiles per frame side (NH, NW <= 416: at most 26 x 13)

struct FrontArgs {
    const uint8_t *frames;          // [n][h][w][3]
    int h, w, NH, NW, OH, OW, tiles_x, tiles_y;
    const int *hb, *hk, *vb, *vk;   // LANCZOS tables (lanczos_tab); coefficient rows are zero-padded to hks / vks
    int hks, vks;
    const float *lut;               // [3][256]
    const float *Wstem, *bstem;     // [32 out][32 taps], [32]
    const float *Wstem_l, *Wp_l;    // lane-order copies of Wstem and Wp (lane_weights): a wave's weight load is one contiguous KB
    const float *Wd, *bd;           // [9][32], [32]
    const float *Wp, *bp;           // [16][32], [16]
    float *Y;                       // [n][OH][OW][16]
    float *in_dbg;                  // optional: the normalised network input [n][NH][NW][3] (debug tap)
    int nr_cap, nc_cap;             // source rows / columns a tile needs at most
    // first source row / column and their number, per tile row / tile column (in the kernel arguments: the first
    // global loads of a workgroup then depend on nothing but its block index)
    short row_lo[FR_MAXT], row_n[FR_MAXT], col_lo[FR_MAXT], col_n[FR_MAXT];
};

// bytes per source-patch row in LDS: whole aligned 32-bit words around the nc * 3 bytes of a row
__host__ __device__ static inline int front_src_stride(int nc_cap) { return ((nc_cap * 3 + 6) >> 2) << 2; }
__host__ __device__ static inline int front_in_offset(int nr_cap, int nc_cap) {
    return (nr_cap * (front_src_stride(nc_cap) + FR_PRS) + 15) & ~15;
}
// LDS (round 4): 35 KB and <= 128 registers = FOUR workgroups per CU (rounds 2-3: 52 KB, three).  [input patch 9.4 KB][S 25.9 KB]:
// the source patch, the horizontal-pass rows, the vertical coefficients and the LUT are dead when features.0 starts writing S,
// so they live INSIDE S's array; D is written over S behind a barrier (the depthwise outputs wait in registers); the three
// layers' weights are not parked in LDS but requested from L2 one phase before their use.
#define FR_IN_BYTES (FR_PR * FR_PRS * 4)
#define FR_SONLY_BYTES (FR_NPX * FR_ES * 4)
static inline int front_small_bytes(int nr_cap, int nc_cap) { return front_in_offset(nr_cap, nc_cap) + (FR_PR * 8 + 24 + 768) * 4; }
static inline int front_lds_bytes() { return FR_IN_BYTES + FR_SONLY_BYTES; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_front(const FrontArgs A) {       // <= 128 registers: four workgroups per CU
    extern __shared__ uint8_t sm_fr[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int bid = xcd_bx();
    const int tx = bid % A.tiles_x;
    bid /= A.tiles_x;
    const int ty = bid % A.tiles_y, f = bid / A.tiles_y;
    const int oy0 = ty * FR_TH, ox0 = tx * FR_TW, py0 = 2 * oy0 - 3, px0 = 2 * ox0 - 3;
    const int srs = front_src_stride(A.nc_cap);
    float *IN_s = (float *)sm_fr;                               // [FR_PR][FR_PRS]  normalised input patch, 0 outside the image
    float *S = (float *)(sm_fr + FR_IN_BYTES);                  // [180][FR_ES]
    uint8_t *src_s = (uint8_t *)S;                              // [nr_cap][srs]      source patch (rows start at an aligned word): inside S's array, like the next four
    uint8_t *tile_s = src_s + A.nr_cap * srs;                   // [nr_cap][FR_PRS]   after the horizontal pass
    int *vk_s = (int *)(src_s + front_in_offset(A.nr_cap, A.nc_cap));   // [FR_PR][8]  vertical coefficient rows
    int *vb_s = vk_s + FR_PR * 8;                               // [FR_PR]            first source row of every patch row
    float *lut_s = (float *)(vb_s + 24);                        // [768]
    float *D = S;                                               // [128][FR_ES]  over S once every depthwise tap has been read
    // 1. the source patch (aligned words), coefficients and LUT are requested here, before anything waits
    const int c4 = tid & 7;
    const int ya = max(py0, 0), yb = min(py0 + FR_PR, A.NH), xa = max(px0, 0), xb = min(px0 + FR_PC, A.NW);
    const int r_lo = A.row_lo[ty], nr = A.row_n[ty], c_lo = A.col_lo[tx], nc3 = A.col_n[tx] * 3, ncol3 = (xb - xa) * 3;
    const uintptr_t g0 = (uintptr_t)A.frames + (((size_t)f * A.h + r_lo) * A.w + c_lo) * 3;     // first byte of the patch
    const int rstep = A.w * 3, gm0 = (int)(g0 & 3), rs3 = rstep & 3;
    const int srow = tid >> 5, swi = tid & 31;                  // source words: 8 rows x 32 words per round
    uint32_t sv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = srow + 8 * it;
        const uintptr_t ga = g0 + (size_t)row * rstep;
        const int nwords = ((int)(ga & 3) + nc3 + 3) >> 2;
        sv[it] = (row < nr && swi < nwords) ? *(const uint32_t *)((ga & ~(uintptr_t)3) + 4 * swi) : 0u;
    }
    // a thread owns one (column, channel) of the patch in both passes, and every second row
    const int ci = tid & 127, half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const bool col_ok = ci < ncol3;
    const int xi = ci / 3, cc = ci - 3 * xi, x = min(xa + xi, A.NW - 1), pcol = (x - px0) * 3 + cc;
    const int xmin = A.hb[2 * x];
    int kh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kh[j] = j < A.hks ? A.hk[x * A.hks + j] : 0;
    int vkv = 0, vbv = 0;
    if (tid < FR_PR * 8) {
        const int y = ya + (tid >> 3), j = tid & 7;
        if (y < yb && j < A.vks) vkv = A.vk[y * A.vks + j];
    } else if (tid < FR_PR * 8 + FR_PR) {
        const int y = ya + tid - FR_PR * 8;
        if (y < yb) vbv = A.vb[2 * y];
    }
    float lv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) lv[i] = A.lut[tid + 256 * i];
    for (int i = tid; i < FR_PR * FR_PRS / 4; i += 256) ((float4 *)IN_s)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = srow + 8 * it;
        if (row < nr && swi < (srs >> 2)) ((uint32_t *)src_s)[row * (srs >> 2) + swi] = sv[it];
    }
    if (tid < FR_PR * 8) vk_s[tid] = vkv;
    else if (tid < FR_PR * 8 + FR_PR) vb_s[tid - FR_PR * 8] = vbv;
#pragma unroll
    for (int i = 0; i < 3; ++i) lut_s[tid + 256 * i] = lv[i];
    __syncthreads();
    // 2. horizontal pass: the column's (at most eight) coefficients are in registers; taps past the end of the row of
    //    coefficients carry 0 (their bytes are whatever follows in LDS), so the eight reads of a row are independent
    if (col_ok) {
        const uint8_t *p = src_s + (xmin - c_lo) * 3 + cc;
        for (int row = half; row < nr; row += 2) {
            const uint8_t *q = p + row * srs + ((gm0 + row * rs3) & 3);
            int acc = 1 << (LZ_PREC - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __mul24((int)q[3 * j], kh[j]);     // |coefficient| < 2^23: the 24-bit multiply is exact and full-rate
            tile_s[row * FR_PRS + pcol] = (uint8_t)min(max(acc >> LZ_PREC, 0), 255);
        }
    }
    float4 wv[4];                                               // features.0 weights of this lane's output channel: requested two phases ahead
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[q] = *(const float4 *)(A.Wstem_l + (q * 64 + lane) * 4);
    __syncthreads();
    // 3. vertical pass + normalisation LUT
    if (col_ok) {
        const float *lc = lut_s + cc * 256;
        const bool dbg_col = A.in_dbg && x >= 2 * ox0 && x < 2 * ox0 + 2 * FR_TW;
        for (int y = ya + half; y < yb; y += 2) {
            const int vrow = y - ya;
            const int4 k0 = *(const int4 *)(vk_s + vrow * 8), k1 = *(const int4 *)(vk_s + vrow * 8 + 4);
            const uint8_t *q = tile_s + (vb_s[vrow] - r_lo) * FR_PRS + pcol;
            int acc = 1 << (LZ_PREC - 1);
            acc += __mul24((int)q[0], k0.x) + __mul24((int)q[FR_PRS], k0.y) + __mul24((int)q[2 * FR_PRS], k0.z) + __mul24((int)q[3 * FR_PRS], k0.w);
            acc += __mul24((int)q[4 * FR_PRS], k1.x) + __mul24((int)q[5 * FR_PRS], k1.y) + __mul24((int)q[6 * FR_PRS], k1.z) + __mul24((int)q[7 * FR_PRS], k1.w);
            const float v = lc[min(max(acc >> LZ_PREC, 0), 255)];
            IN_s[(y - py0) * FR_PRS + pcol] = v;
            if (dbg_col && y >= 2 * oy0 && y < 2 * oy0 + 2 * FR_TH) A.in_dbg[(((size_t)f * A.NH + y) * A.NW + x) * 3 + cc] = v;
        }
    }
    __syncthreads();
    // 4. features.0 on the 180 halo pixels (k_stem_mfma's operand roles and tap order); zero outside the image
    float4 wd[9];                                               // depthwise weights: requested now, used behind the MFMA phase
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t] = *(const float4 *)(A.Wd + t * 32 + c4 * 4);
    for (int t = wave; t < (FR_NPX + 31) / 32; t += 4) {
        const int hp = t * 32 + r, hq = min(hp, FR_NPX - 1);
        const int hy = hq / FR_HW, hx = hq - hy * FR_HW;
        const float *src = IN_s + (2 * hy) * FR_PRS + (2 * hx) * 3;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float tp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * q + 4 * hh + e;                // tap (ky, kx, ci) = (k / 9, (k % 9) / 3, k % 3)
                const int ky = (k * 57) >> 9;
                tp[e] = k < 27 ? src[k + (FR_PRS - 9) * ky] : 0.f;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].x, tp[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].y, tp[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].z, tp[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q].w, tp[3], acc, 0, 0, 0);
        }
        if (hp < FR_NPX) {
            const bool in = (unsigned)(oy0 - 1 + hy) < (unsigned)A.OH && (unsigned)(ox0 - 1 + hx) < (unsigned)A.OW;
            float *sp = S + hp * FR_ES + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) {
                    const float4 b = *(const float4 *)(A.bstem + 8 * g + 4 * hh);
                    v.x = fminf(fmaxf(acc[4 * g] + b.x, 0.f), 6.f);
                    v.y = fminf(fmaxf(acc[4 * g + 1] + b.y, 0.f), 6.f);
                    v.z = fminf(fmaxf(acc[4 * g + 2] + b.z, 0.f), 6.f);
                    v.w = fminf(fmaxf(acc[4 * g + 3] + b.w, 0.f), 6.f);
                }
                *(float4 *)(sp + 8 * g) = v;
            }
        }
    }
    __syncthreads();
    // 5. depthwise 3x3 + bias + ReLU6 (k_irb's tap order): thread = channels 4 c4 .. + 3 of four adjacent pixels of a row,
    //    from six columns per halo row (18 LDS reads for four outputs instead of 36: the phase is bound by LDS bandwidth)
    float4 wq[4];                                               // project weights: requested now, used behind the depthwise phase
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wq[q] = *(const float4 *)(A.Wp_l + (q * 64 + lane) * 4);
        if (r >= 16) wq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const float4 b = *(const float4 *)(A.bd + c4 * 4);
        const int oy = tid >> 5, ox = ((tid >> 3) & 3) * 4;
        float4 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float *sp = S + ((oy + ky) * FR_HW + ox) * FR_ES + c4 * 4;
            float4 xv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) xv[j] = *(const float4 *)(sp + j * FR_ES);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 w = wd[ky * 3 + kx];
#pragma unroll
                for (int i = 0; i < 4; ++i) fma4(a4[i], xv[i + kx], w);
            }
        }
        __syncthreads();                                        // D lies over S: every tap has been read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 o;
            o.x = fminf(fmaxf(a4[i].x + b.x, 0.f), 6.f);
            o.y = fminf(fmaxf(a4[i].y + b.y, 0.f), 6.f);
            o.z = fminf(fmaxf(a4[i].z + b.z, 0.f), 6.f);
            o.w = fminf(fmaxf(a4[i].w + b.w, 0.f), 6.f);
            *(float4 *)(D + (oy * FR_TW + ox + i) * FR_ES + c4 * 4) = o;
        }
    }
    __syncthreads();
    // 6. project 32 -> 16: wave = pixels 32 wave .. + 31; k = 0..15 and 16..31 accumulate apart and are added afterwards,
    //    like the two k-split partials of k_irb on this block
    {
        f32x16 a0, a1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        const int px = 32 * wave + r;
        const float *dp = D + px * FR_ES + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *(const float4 *)(dp + 8 * q);
            if (q < 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].x, a.x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].y, a.y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].z, a.z, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].w, a.w, a0, 0, 0, 0);
            } else {
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].x, a.x, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].y, a.y, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].z, a.z, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q].w, a.w, a1, 0, 0, 0);
            }
        }
        const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
        if (oy < A.OH && ox < A.OW) {
            float *yp = A.Y + (((size_t)f * A.OH + oy) * A.OW + ox) * 16;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int col = 8 * g + 4 * hh;
                const float4 b = *(const float4 *)(A.bp + col);
                float4 v = make_float4(a0[4 * g] + a1[4 * g], a0[4 * g + 1] + a1[4 * g + 1], a0[4 * g + 2] + a1[4 * g + 2],
                                       a0[4 * g + 3] + a1[4 * g + 3]);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                *(float4 *)(yp + col) = v;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// pointwise conv as a GEMM on the f32 MFMA:  Y[M,N] = act(X[M,K] * W[N,K]^T + b) (+ R)
// Block = 4 waves; wave v owns rows m0+32v..+31 and TN column tiles of 32.
// Lane l (r = l&31, hh = l>>5) feeds A[r][k] and B[k][r] with k = kk + 4*hh + s for the
// s-th MFMA of an 8-deep K chunk: both operands come from one float4 per lane, read
// straight from global memory (the k order inside a chunk is a consistent permutation).
// --------------------------------------------------------------------------------------
// Optional epilogue term of k_pw: a low-resolution tensor U [n][UH][UW][ldu] added after 2x bilinear
// up-sampling (align_corners = False, the arithmetic of k_upsample2x).  A 1x1 convolution commutes with the
// up-sampling, so the decoder's "up-sample, concatenate with the skip, expand" (model.py:463-483) is evaluated
// as conv(skip part) + up-sample(conv(low-resolution part)): the up-sampled half of the concatenation is
// contracted at a quarter of the pixels and never materialised.
struct UpsAdd {
    const float *U;
    int UH, UW, ldu;
    FDiv dOW, dOH;
};

// TR = true: the MFMA operands are swapped (weights as A, activations as B), so a lane ends up with ONE pixel and
// 16 output channels in runs of four -- the epilogue then moves float4 (4 stores per tile instead of 16).
template <int TN, int PF, bool TR = false>
__global__ __launch_bounds__(256) SVC_NO_PK void k_pw(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                            const float *__restrict__ bias, const float *__restrict__ R, int ldr,
                                            float *__restrict__ Y, int ldy, int M, int N, int Npad, int K,
                                            int relu6, UpsAdd ups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = xcd_bx() * 128 + wave * 32;
    const int n0 = blockIdx.y * (32 * TN);
    if (m0 >= M) return;
    const int row = min(m0 + r, M - 1);
    const float *xp = X + (size_t)row * ldx + 4 * hh;
    const float *wp[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        int col = min(n0 + t * 32 + r, Npad - 1);
        wp[t] = Wt + (size_t)col * ldw + 4 * hh;
    }
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // the operands of PF k-steps (8 deep each) are in flight ahead of the MFMAs (PF = 1 measured
    // fastest on MI355X: these layers are bound by the L1 line rate, not by load latency)
    const int nsteps = K >> 3;
    float4 A[PF], B[PF][TN];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < nsteps) {
            A[p] = *(const float4 *)(xp + 8 * p);
#pragma unroll
            for (int t = 0; t < TN; ++t) B[p][t] = *(const float4 *)(wp[t] + 8 * p);
        }
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int st = s0 + p;
            if (st < nsteps) {
                const float4 a = A[p];
                float4 b[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) b[t] = B[p][t];
                if (st + PF < nsteps) {
                    A[p] = *(const float4 *)(xp + 8 * (st + PF));
#pragma unroll
                    for (int t = 0; t < TN; ++t) B[p][t] = *(const float4 *)(wp[t] + 8 * (st + PF));
                }
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    if (TR) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].x, a.x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].y, a.y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].z, a.z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].w, a.w, acc[t], 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (TR) {
        // lane = pixel m0 + r; accumulator i = channel n0 + 32t + 8(i>>2) + 4hh + (i&3)
        const int rr = m0 + r;
        if (rr >= M) return;
        const float *u00 = nullptr, *u01 = nullptr, *u10 = nullptr, *u11 = nullptr;
        float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
        if (ups.U) {
            uint32_t ox, oy;
            const uint32_t f = fdivmod(fdivmod((uint32_t)rr, ups.dOW, ox), ups.dOH, oy);
            const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < ups.UH - 1 ? 1 : 0), x1 = x0 + (x0 < ups.UW - 1 ? 1 : 0);
            const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float *uf = ups.U + (size_t)f * ups.UH * ups.UW * ups.ldu;
            u00 = uf + ((size_t)y0 * ups.UW + x0) * ups.ldu; u01 = uf + ((size_t)y0 * ups.UW + x1) * ups.ldu;
            u10 = uf + ((size_t)y1 * ups.UW + x0) * ups.ldu; u11 = uf + ((size_t)y1 * ups.UW + x1) * ups.ldu;
            w00 = lx0; w01 = lx1; w10 = ly0; w11 = ly1;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + t * 32 + 8 * g + 4 * hh;
                if (col >= N) continue;
                float4 v = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                if (bias) { const float4 bv = *(const float4 *)(bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                if (ups.U) {
                    const float4 a0 = *(const float4 *)(u00 + col), a1 = *(const float4 *)(u01 + col);
                    const float4 c0 = *(const float4 *)(u10 + col), c1 = *(const float4 *)(u11 + col);
                    v.x += w10 * (w00 * a0.x + w01 * a1.x) + w11 * (w00 * c0.x + w01 * c1.x);
                    v.y += w10 * (w00 * a0.y + w01 * a1.y) + w11 * (w00 * c0.y + w01 * c1.y);
                    v.z += w10 * (w00 * a0.z + w01 * a1.z) + w11 * (w00 * c0.z + w01 * c1.z);
                    v.w += w10 * (w00 * a0.w + w01 * a1.w) + w11 * (w00 * c0.w + w01 * c1.w);
                }
                if (R) { const float4 rv = *(const float4 *)(R + (size_t)rr * ldr + col); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                if (relu6) {
                    v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
                    v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
                }
                *(float4 *)(Y + (size_t)rr * ldy + col) = v;
            }
        return;
    }
    if (ups.U) {
        // rows first: the four taps and weights of an output pixel are shared by all of the wave's column tiles
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (rr >= M) continue;
            uint32_t ox, oy;
            const uint32_t f = fdivmod(fdivmod((uint32_t)rr, ups.dOW, ox), ups.dOH, oy);
            const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < ups.UH - 1 ? 1 : 0), x1 = x0 + (x0 < ups.UW - 1 ? 1 : 0);
            const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float *uf = ups.U + (size_t)f * ups.UH * ups.UW * ups.ldu;
            const float *u00 = uf + ((size_t)y0 * ups.UW + x0) * ups.ldu, *u01 = uf + ((size_t)y0 * ups.UW + x1) * ups.ldu;
            const float *u10 = uf + ((size_t)y1 * ups.UW + x0) * ups.ldu, *u11 = uf + ((size_t)y1 * ups.UW + x1) * ups.ldu;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int col = n0 + t * 32 + r;
                if (col >= N) continue;
                float v = acc[t][i] + (bias ? bias[col] : 0.f);
                v += ly0 * (lx0 * u00[col] + lx1 * u01[col]) + ly1 * (lx0 * u10[col] + lx1 * u11[col]);
                if (relu6) v = fminf(fmaxf(v, 0.f), 6.f);
                Y[(size_t)rr * ldy + col] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + t * 32 + r;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (rr < M) {
                float v = acc[t][i] + bv;
                if (R) v += R[(size_t)rr * ldr + col];
                if (relu6) v = fminf(fmaxf(v, 0.f), 6.f);
                Y[(size_t)rr * ldy + col] = v;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// Short-K pointwise layers (K <= 160: the 6x expansions of blocks 8-17, the skips, the decoder expansions) with the
// activations RESIDENT IN REGISTERS.  k_pw feeds both MFMA operands from global memory per k-step: every column tile
// re-reads the activation rows, the four waves of a workgroup each re-read the same weight rows, one load pair is in
// flight per wave, and a load instruction touches 32 cache lines for 1 KB -- the CU's vector-memory path is as busy as
// its matrix pipes and the waves sit in s_waitcnt (SQ_WAIT_ANY 59 %).  Here a wave loads its 32 rows x K ONCE, all
// k-steps in flight together (KS float4 per lane), the workgroup's weight chunk (NTW column tiles x K) is staged once
// in LDS and shared by the four waves, and the wave then walks the NTW column tiles with the activations in registers:
// per tile K/2 MFMAs fed by one ds_read_b128 per four MFMAs, and a float4 epilogue (operands swapped: a lane owns one
// pixel).  Global loads per MFMA drop by 2 NTW x; the k order of every sum is k_pw's, so results are bit-identical.
// --------------------------------------------------------------------------------------
// The epilogue goes through a per-wave LDS slab: the accumulator layout gives a lane 16 B pieces of ONE pixel's row, so a
// direct float4 store instruction touches 32 cache lines for 1 KB and the CU's store path, not HBM, limits the kernel
// (tools/micro/pw_phases.hip: the stores of a 64 -> 384 layer alone take 6.6 us = 5 B/clk/CU, as long as its MFMAs and
// loads together, and nothing overlaps them).  Transposed through the slab, a store instruction writes 8 rows x 128 B:
// eight whole lines.
#define PWR_SLAB 36            // floats per slab row: 32 channels + 4 pad
// MX (round 5): the split-bf16 form (x3_split above).  Wt is then the X3_ROWS copy of the weight slice (rows of Q * 6 uint4), the
// chunk in LDS keeps that form (row stride Q * 6 + 1 uint4: an odd number of 16-B slots, conflict-free b128 reads), the wave's
// activations are split once, in the prologue, and a tile is Q * X3_NP MFMAs of 32 cycles instead of 4 KS of 64.
template <int KS, bool UPS, bool MX = false>     // K = 8 * KS; UPS: the up-sample-add term (its taps are requested in front of every tile's MFMAs)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_pwr(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                             const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N,
                                             int Npad, int ntw, int relu6, UpsAdd ups) {
    constexpr int K = 8 * KS, WS = K + 4;                  // LDS row stride of the weight chunk (floats): conflict-free b128 rows
    constexpr int Q = KS / 2, RS = Q * 6 + 1;              // MX: 16-deep steps, row stride in uint4
    extern __shared__ float sm_pwr[];                      // [4 waves][32][PWR_SLAB] epilogue slabs | [128] bias chunk | [ntw * 32][WS] weight chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    float *slab = sm_pwr + wave * (32 * PWR_SLAB);
    float *bch = sm_pwr + 4 * 32 * PWR_SLAB;
    float *wch = bch + 128;
    const int m0 = xcd_bx() * 128 + wave * 32;
    const int n0 = blockIdx.y * (32 * ntw);
    const int ncols = min(32 * ntw, Npad - n0);            // multiple of 32, at most 128
    // activations of the wave's 32 pixels: every load in flight at once -- requested in WHOLE LINES (round 4: eight lanes per
    // row and 128-byte block, 8 cache lines per load; a lane asking for its own MFMA operands touches 32 lines per load, a
    // quarter of each, and these loads were three quarters of the lines the kernel touches) and handed to the MFMA layout
    // (a lane = one pixel, k = 8p + 4hh ..) through the wave's epilogue slab, once, behind the weight staging
    constexpr int NB = KS / 4;                              // blocks of 32 input channels (K is 64, 96, 128 or 160)
    static_assert(KS % 4 == 0, "K must be a multiple of 32");
    const int lrow = lane >> 3, lc4 = (lane & 7) * 4;
    float4 G[NB][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float *xr = X + (size_t)min(m0 + lrow + 8 * j, M - 1) * ldx + lc4;
#pragma unroll
        for (int b = 0; b < NB; ++b) G[b][j] = *(const float4 *)(xr + 32 * b);
    }
    // weight chunk and bias chunk -> LDS (coalesced float4 rows)
    constexpr int K4 = K / 4;
    uint4 *wq = (uint4 *)wch;
    if constexpr (MX) {
        const uint4 *W3 = (const uint4 *)Wt;
        for (int i = tid; i < ncols * (Q * 6); i += 256) {
            const int row = i / (Q * 6), c = i - row * (Q * 6);
            wq[row * RS + c] = W3[(size_t)(n0 + row) * (Q * 6) + c];
        }
    } else {
        for (int i = tid; i < ncols * K4; i += 256) {
            const int row = i / K4, c4 = i - row * K4;
            *(float4 *)(wch + row * WS + c4 * 4) = *(const float4 *)(Wt + (size_t)(n0 + row) * ldw + c4 * 4);
        }
    }
    if (tid < ncols) bch[tid] = (bias && n0 + tid < N) ? bias[n0 + tid] : 0.f;
    float4 A[MX ? 1 : KS];
    X3 A3[MX ? Q : 1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4 *)(slab + (lrow + 8 * j) * PWR_SLAB + lc4) = G[b][j];
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the slab is private to the wave
        __builtin_amdgcn_wave_barrier();
        if constexpr (MX) {
            float4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *(const float4 *)(slab + r * PWR_SLAB + 8 * q + 4 * hh);
            A3[MX ? 2 * b : 0] = x3_split(a[0], a[1]);
            A3[MX ? 2 * b + 1 : 0] = x3_split(a[2], a[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) A[MX ? 0 : 4 * b + q] = *(const float4 *)(slab + r * PWR_SLAB + 8 * q + 4 * hh);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();                    // the next block overwrites the slab
    }
    __syncthreads();
    if (m0 >= M) return;
    // Every load of this wave has landed (the barrier above waited for the weight chunk, requested after the
    // activations).  Saying so explicitly empties the compiler's vector-memory scoreboard in front of the tile loop, and
    // the loop itself issues no global load (the bias comes from LDS): otherwise hipcc puts an s_waitcnt vmcnt(0) at the
    // loop head -- the activation registers count as "possibly pending" on the back edge -- or in front of the first use
    // of a bias value, and because loads and stores retire in order every tile then waits for the STORES of the tile
    // before it: the chip alternates between an MFMA phase and a store phase (tools/micro/pw_phases.hip).
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    // store role of the lane: rows (lane >> 3) + 8 it, channels 4 (lane & 7) .. + 3 of the tile
    const int srow = lane >> 3, sc = (lane & 7) * 4;
    const int nt = ncols >> 5;
    // up-sample-add: the four taps and weights of each of the lane's four rows do not depend on the tile
    uint32_t uo[UPS ? 4 : 1][4];
    float uw[UPS ? 4 : 1][4];
    if (UPS) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = min(m0 + srow + 8 * it, M - 1);
            uint32_t ox, oy;
            const uint32_t f = fdivmod(fdivmod((uint32_t)rr, ups.dOW, ox), ups.dOH, oy);
            const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < ups.UH - 1 ? 1 : 0), x1 = x0 + (x0 < ups.UW - 1 ? 1 : 0);
            const float ly1 = sy - y0, lx1 = sx - x0;
            const uint32_t fb = f * (uint32_t)(ups.UH * ups.UW);
            uo[UPS ? it : 0][0] = (fb + y0 * ups.UW + x0) * (uint32_t)ups.ldu; uo[UPS ? it : 0][1] = (fb + y0 * ups.UW + x1) * (uint32_t)ups.ldu;
            uo[UPS ? it : 0][2] = (fb + y1 * ups.UW + x0) * (uint32_t)ups.ldu; uo[UPS ? it : 0][3] = (fb + y1 * ups.UW + x1) * (uint32_t)ups.ldu;
            uw[UPS ? it : 0][0] = 1.f - lx1; uw[UPS ? it : 0][1] = lx1; uw[UPS ? it : 0][2] = 1.f - ly1; uw[UPS ? it : 0][3] = ly1;
        }
    }
    for (int t = 0; t < nt; ++t) {
        const int col = n0 + t * 32 + sc;
        float4 tap[UPS ? 4 : 1][4];
        if (UPS) {                                          // requested now, used behind the tile's MFMAs
            const int cc = min(col, N - 4);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) tap[UPS ? it : 0][q] = *(const float4 *)(ups.U + uo[UPS ? it : 0][q] + cc);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if constexpr (MX) {
            const uint4 *bq = wq + (t * 32 + r) * RS + 3 * hh;
#pragma unroll
            for (int q = 0; q < Q; ++q) x3_mma(acc, x3_load(bq + 6 * q), A3[MX ? q : 0]);                        // weights as A: lane = pixel
        } else {
            const float *bq = wch + (t * 32 + r) * WS + 4 * hh;
#pragma unroll
            for (int p = 0; p < KS; ++p) {
                const float4 b = *(const float4 *)(bq + 8 * p);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, A[MX ? 0 : p].x, acc, 0, 0, 0);      // swapped: lane = pixel
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, A[MX ? 0 : p].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, A[MX ? 0 : p].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, A[MX ? 0 : p].w, acc, 0, 0, 0);
            }
        }
        // accumulator i = channel 32t + 8(i>>2) + 4hh + (i&3) of pixel r -> slab[r][channel]
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4 *)(slab + r * PWR_SLAB + 8 * g + 4 * hh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the slab is private to the wave
        __builtin_amdgcn_wave_barrier();
        const float4 bv = *(const float4 *)(bch + t * 32 + sc);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it, rr = m0 + row;
            float4 v = *(const float4 *)(slab + row * PWR_SLAB + sc);
            if (rr >= M || col >= N) continue;
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (UPS) {                                      // + the 2x bilinear up-sampling of the low-resolution product (see UpsAdd)
                const float lx0 = uw[UPS ? it : 0][0], lx1 = uw[UPS ? it : 0][1], ly0 = uw[UPS ? it : 0][2], ly1 = uw[UPS ? it : 0][3];
                const float4 a0 = tap[UPS ? it : 0][0], a1 = tap[UPS ? it : 0][1], c0 = tap[UPS ? it : 0][2], c1 = tap[UPS ? it : 0][3];
                v.x += ly0 * (lx0 * a0.x + lx1 * a1.x) + ly1 * (lx0 * c0.x + lx1 * c1.x);
                v.y += ly0 * (lx0 * a0.y + lx1 * a1.y) + ly1 * (lx0 * c0.y + lx1 * c1.y);
                v.z += ly0 * (lx0 * a0.z + lx1 * a1.z) + ly1 * (lx0 * c0.z + lx1 * c1.z);
                v.w += ly0 * (lx0 * a0.w + lx1 * a1.w) + ly1 * (lx0 * c0.w + lx1 * c1.w);
            }
            if (relu6) {
                v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
                v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
            }
            *(float4 *)(Y + (size_t)rr * ldy + col) = v;
        }
        __builtin_amdgcn_wave_barrier();                    // the next tile overwrites the slab
    }
}

// --------------------------------------------------------------------------------------
// Pointwise conv on v_mfma_f32_16x16x4_f32 (K % 16 == 0): the four 16-lane groups of a wave
// carry four k slots, so one float4 load instruction covers 16 rows x 64 contiguous bytes
// (16 cache lines) instead of 32 rows x 32 bytes (32 lines) — these layers are bound by the
// L1 line rate (SQ_WAIT_ANY ~ 68 % in k_pw).  Wave tile 32 x (32*TN) = 2 x 2TN accumulators
// of 16x16; lane l (r16 = l&15, q = l>>4) feeds A[r16][k0+4q+j], B[k0+4q+j][r16] in MFMA j.
// --------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MT, int NT>     // wave tile = (16*MT) rows x (16*NT) columns; workgroup = 4 waves stacked along M
__global__ __launch_bounds__(256) void k_pw16(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                              const float *__restrict__ bias, const float *__restrict__ R, int ldr,
                                              float *__restrict__ Y, int ldy, int M, int N, int Npad, int K,
                                              int relu6) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = xcd_bx() * (64 * MT) + wave * (16 * MT);
    const int n0 = blockIdx.y * (16 * NT);
    if (m0 >= M) return;
    const float *xa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xa[i] = X + (size_t)min(m0 + 16 * i + r16, M - 1) * ldx + 4 * q;
    const float *wb[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) wb[c] = Wt + (size_t)min(n0 + 16 * c + r16, Npad - 1) * ldw + 4 * q;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][c][e] = 0.f;
#pragma unroll 2
    for (int k = 0; k < K; k += 16) {
        float4 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *(const float4 *)(xa[i] + k);
#pragma unroll
        for (int c = 0; c < NT; ++c) b[c] = *(const float4 *)(wb[c] + k);
#define PW16_STEP(EL)                                                                                       \
    _Pragma("unroll") for (int c = 0; c < NT; ++c) _Pragma("unroll") for (int i = 0; i < MT; ++i)           \
        acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].EL, b[c].EL, acc[i][c], 0, 0, 0);
        PW16_STEP(x) PW16_STEP(y) PW16_STEP(z) PW16_STEP(w)
#undef PW16_STEP
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int col = n0 + 16 * c + r16;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = m0 + 16 * i + 4 * q + e;
                if (rr < M) {
                    float v = acc[i][c][e] + bv;
                    if (R) v += R[(size_t)rr * ldr + col];
                    if (relu6) v = fminf(fmaxf(v, 0.f), 6.f);
                    Y[(size_t)rr * ldy + col] = v;
                }
            }
    }
}

// --------------------------------------------------------------------------------------
// Split-K variant for the long-K, small-M layers (8x13 / 16x26 levels at B = 32), where k_pw runs
// one wave per SIMD through a serial K loop: the workgroup owns 32 rows x (32*TN) columns and its
// four waves each take a quarter of K; the partial tiles meet in LDS and are summed in a fixed
// order (deterministic), every wave finishing four of the sixteen accumulator rows.
// --------------------------------------------------------------------------------------
#ifndef SK_DEPTH
#define SK_DEPTH 2     // k-steps k_pw_sk keeps in flight per wave (round 4, [N][K] weights: 2 / 4 / 6 -> pw class 1.442 / 1.437 / 1.476 ms; lane-order weights: 2 / 3 / 4 -> the pass 1.383 / 1.385 / 1.388 ms alone, 1.020 / 1.026 / 1.028 shared)
#endif
// LW (round 4): the weights come from a LANE-ORDER copy of the matrix (lane_weights below): element ((k-step, column tile), lane)
// = the float4 that lane feeds into the tile's four MFMAs of the step, so a wave's load is one contiguous KB (8 cache lines,
// each used whole) instead of 32 rows x 32 B (32 lines, a quarter of each) -- the texture-address units are busy 48 % of this
// kernel and stalled by the L1 37 % of it (profiles/r04_pmc_mem_pipes.txt).  ldw then carries the number of column tiles.
// MX (round 5): split-bf16 operands (x3_split).  Wt is the X3_LANES copy (three uint4 per lane and 16-deep step: a wave's weight load
// is 3 KB contiguous), the activations of a step -- the lane's two float4 -- are split in registers behind their load, K % 16 == 0.
template <int TN, bool LW = false, bool MX = false>
__global__ __launch_bounds__(256) void k_pw_sk(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                               const float *__restrict__ bias, const float *__restrict__ R, int ldr,
                                               float *__restrict__ Y, int ldy, int M, int N, int Npad, int K,
                                               int relu6) {
    __shared__ float red[4][TN][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = xcd_bx() * 32, n0 = blockIdx.y * (32 * TN);
    const float *xp = X + (size_t)min(m0 + r, M - 1) * ldx + 4 * hh;
    const float *wp[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
        wp[t] = LW ? Wt + ((size_t)(blockIdx.y * TN + t) * 64 + lane) * 4 : Wt + (size_t)min(n0 + t * 32 + r, Npad - 1) * ldw + 4 * hh;
    const size_t wstep = LW ? (size_t)ldw * 256 : 8;       // floats from one k-step's float4 to the next
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nsteps = MX ? K >> 4 : K >> 3;
    const int s_lo = (wave * nsteps) >> 2, s_hi = ((wave + 1) * nsteps) >> 2;
    if constexpr (MX) {
        // two 16-deep steps in flight, straight-line as below (prefetch index clamped to the wave's last step)
        const uint4 *w3[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) w3[t] = (const uint4 *)Wt + ((size_t)(blockIdx.y * TN + t) * 64 + lane) * 3;
        const size_t w3step = (size_t)ldw * 64 * 3;         // uint4 from one step to the next (ldw = column tiles of the copy)
#define SK3_LOAD(A_, B_, st_)                                                                    \
    {                                                                                            \
        A_[0] = *(const float4 *)(xp + 16 * (st_)); A_[1] = *(const float4 *)(xp + 16 * (st_) + 8); \
        _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                         \
            const uint4 *q_ = w3[t] + w3step * (st_);                                            \
            B_[t][0] = q_[0]; B_[t][1] = q_[1]; B_[t][2] = q_[2];                                \
        }                                                                                        \
    }
#define SK3_MFMA(A_, B_)                                                                         \
    {                                                                                            \
        const X3 a_ = x3_split(A_[0], A_[1]);                                                    \
        _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                         \
            X3Q h_, m_, l_;                                                                      \
            h_.q = B_[t][0]; m_.q = B_[t][1]; l_.q = B_[t][2];                                   \
            x3_mma(acc[t], X3{h_.v, m_.v, l_.v}, a_);                                            \
        }                                                                                        \
    }
        if (s_hi > s_lo) {
            constexpr int D = 2;
            const int last = s_hi - 1, groups = (s_hi - s_lo) / D, rem = (s_hi - s_lo) - groups * D;
            float4 A[D][2];
            uint4 B[D][TN][3];
#pragma unroll
            for (int d = 0; d < D; ++d) SK3_LOAD(A[d], B[d], min(s_lo + d, last))
            int st = s_lo;
            for (int q = 0; q < groups; ++q, st += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    float4 a[2] = {A[d][0], A[d][1]};
                    uint4 b[TN][3];
#pragma unroll
                    for (int t = 0; t < TN; ++t) { b[t][0] = B[d][t][0]; b[t][1] = B[d][t][1]; b[t][2] = B[d][t][2]; }
                    SK3_LOAD(A[d], B[d], min(st + D + d, last))
                    SK3_MFMA(a, b)
                }
            }
            if (rem) SK3_MFMA(A[0], B[0])
        }
#undef SK3_LOAD
#undef SK3_MFMA
    } else {
    // The grid of these layers puts two waves on a SIMD, not enough to hide an L2 round trip per k-step (the plain loop
    // compiles to: three loads, s_waitcnt vmcnt(0), eight MFMAs).  Software pipeline, two k-steps deep, written as
    // straight-line code (no condition inside the loop: the prefetch index is clamped to the wave's last step, so the
    // compiler keeps counted waits instead of draining the queue on every branch); an odd last step runs on its own.
#define SK_LOAD(A_, B_, st_)                                                                     \
    {                                                                                            \
        A_ = *(const float4 *)(xp + 8 * (st_));                                                  \
        _Pragma("unroll") for (int t = 0; t < TN; ++t) B_[t] = *(const float4 *)(wp[t] + wstep * (st_)); \
    }
#define SK_MFMA(A_, B_)                                                                          \
    _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                             \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(B_[t].x, A_.x, acc[t], 0, 0, 0);           \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(B_[t].y, A_.y, acc[t], 0, 0, 0);           \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(B_[t].z, A_.z, acc[t], 0, 0, 0);           \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(B_[t].w, A_.w, acc[t], 0, 0, 0);           \
    }
    if (s_hi > s_lo) {
        // SK_DEPTH k-steps in flight, the same ascending k order per wave whatever the depth
        constexpr int D = SK_DEPTH;
        const int last = s_hi - 1, groups = (s_hi - s_lo) / D, rem = (s_hi - s_lo) - groups * D;
        float4 A[D], B[D][TN];
#pragma unroll
        for (int d = 0; d < D; ++d) SK_LOAD(A[d], B[d], min(s_lo + d, last))
        int st = s_lo;
        for (int q = 0; q < groups; ++q, st += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float4 a = A[d];
                float4 b[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) b[t] = B[d][t];
                SK_LOAD(A[d], B[d], min(st + D + d, last))
                SK_MFMA(a, b)
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < rem) SK_MFMA(A[d], B[d])            // A[d] / B[d] hold step st + d by now
    }
#undef SK_LOAD
#undef SK_MFMA
    }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][t][i][lane] = acc[t][i];
    __syncthreads();
    // a lane holds row m0 + r and accumulator i = column 8(i>>2) + 4hh + (i&3) of the tile: wave w finishes the
    // column run g = w with one float4
    const int rr = m0 + r;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int g = wave;
        const int col = n0 + t * 32 + 8 * g + 4 * hh;
        if (col >= N || rr >= M) continue;
        float4 v;
        v.x = ((red[0][t][4 * g][lane] + red[1][t][4 * g][lane]) + red[2][t][4 * g][lane]) + red[3][t][4 * g][lane];
        v.y = ((red[0][t][4 * g + 1][lane] + red[1][t][4 * g + 1][lane]) + red[2][t][4 * g + 1][lane]) + red[3][t][4 * g + 1][lane];
        v.z = ((red[0][t][4 * g + 2][lane] + red[1][t][4 * g + 2][lane]) + red[2][t][4 * g + 2][lane]) + red[3][t][4 * g + 2][lane];
        v.w = ((red[0][t][4 * g + 3][lane] + red[1][t][4 * g + 3][lane]) + red[2][t][4 * g + 3][lane]) + red[3][t][4 * g + 3][lane];
        if (bias) { const float4 bv = *(const float4 *)(bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
        if (R) { const float4 rv = *(const float4 *)(R + (size_t)rr * ldr + col); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
        if (relu6) {
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
        }
        *(float4 *)(Y + (size_t)rr * ldy + col) = v;
    }
}

// --------------------------------------------------------------------------------------
// depthwise 3x3 pad 1, stride S, + bias, ReLU6.  NHWC, one thread = one pixel x 4 channels.
// w layout [9][C]
// --------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_dw(const float *__restrict__ X, const float *__restrict__ Wt,
                                            const float *__restrict__ bias, float *__restrict__ Y, int n, int H,
                                            int W, int C, int OH, int OW, FDiv dC4, FDiv dOW, FDiv dOH) {
    const int C4 = C >> 2;
    const uint32_t gid = xcd_bx() * 256u + threadIdx.x;
    const uint32_t total = (uint32_t)n * OH * OW * C4;
    if (gid >= total) return;
    uint32_t c4, ox, oy;
    const uint32_t pix = fdivmod(gid, dC4, c4);
    const uint32_t f = fdivmod(fdivmod(pix, dOW, ox), dOH, oy);
    const float *xf = X + (size_t)f * H * W * C + c4 * 4;
    const float *wf = Wt + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        int iy = oy * S - 1 + ky;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int ix = ox * S - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const float4 x = *(const float4 *)(xf + ((size_t)iy * W + ix) * C);
            const float4 w = *(const float4 *)(wf + (ky * 3 + kx) * C);
            acc.x = fmaf(x.x, w.x, acc.x);
            acc.y = fmaf(x.y, w.y, acc.y);
            acc.z = fmaf(x.z, w.z, acc.z);
            acc.w = fmaf(x.w, w.w, acc.w);
        }
    }
    const float4 b = *(const float4 *)(bias + c4 * 4);
    acc.x = fminf(fmaxf(acc.x + b.x, 0.f), 6.f);
    acc.y = fminf(fmaxf(acc.y + b.y, 0.f), 6.f);
    acc.z = fminf(fmaxf(acc.z + b.z, 0.f), 6.f);
    acc.w = fminf(fmaxf(acc.w + b.w, 0.f), 6.f);
    *(float4 *)(Y + (size_t)gid * 4) = acc;
}

// Stride-1 depthwise with a TX x TY register tile of outputs per thread (x 4 channels).  k_dw issues 18
// vector loads per output (9 taps + 9 weights) and is bound by the CU's vector-memory issue rate, not by
// HBM; here a thread walks its TY+2 input rows once, keeps one row of TX+2 taps in registers and feeds
// every output row that uses it: (TY+2)(TX+2)+9 loads for TX*TY outputs (4.1 per output at 4x2).  Per
// output the taps are still accumulated in the order ky, kx with out-of-image taps contributing nothing,
// so the result equals k_dw<1>'s.
template <int TX, int TY>
__global__ __launch_bounds__(256) void k_dw_tile(const float *__restrict__ X, const float *__restrict__ Wt,
                                                 const float *__restrict__ bias, float *__restrict__ Y, int n,
                                                 int H, int W, int C, FDiv dC4, FDiv dGX, FDiv dGY, uint32_t total) {
    const uint32_t gid = xcd_bx() * 256u + threadIdx.x;
    if (gid >= total) return;
    uint32_t c4, gx, gy;
    const uint32_t cell = fdivmod(gid, dC4, c4);
    const uint32_t f = fdivmod(fdivmod(cell, dGX, gx), dGY, gy);
    const int ox0 = gx * TX, oy0 = gy * TY;
    const float *xf = X + (size_t)f * H * W * C + c4 * 4;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *(const float4 *)(Wt + t * C + c4 * 4);
    float4 acc[TY][TX];
#pragma unroll
    for (int ty = 0; ty < TY; ++ty)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) acc[ty][tx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < TY + 2; ++r) {
        const int iy = oy0 - 1 + r;
        if (iy < 0 || iy >= H) continue;
        float4 row[TX + 2];
#pragma unroll
        for (int j = 0; j < TX + 2; ++j) {
            const int ix = ox0 - 1 + j;
            row[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ix >= 0 && ix < W) row[j] = *(const float4 *)(xf + ((size_t)iy * W + ix) * C);
        }
#pragma unroll
        for (int ty = 0; ty < TY; ++ty) {
            const int ky = r - ty;
            if (ky < 0 || ky > 2) continue;
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 x = row[tx + kx], ww = w[ky * 3 + kx];
                    acc[ty][tx].x = fmaf(x.x, ww.x, acc[ty][tx].x);
                    acc[ty][tx].y = fmaf(x.y, ww.y, acc[ty][tx].y);
                    acc[ty][tx].z = fmaf(x.z, ww.z, acc[ty][tx].z);
                    acc[ty][tx].w = fmaf(x.w, ww.w, acc[ty][tx].w);
                }
        }
    }
    const float4 b = *(const float4 *)(bias + c4 * 4);
    float *yf = Y + (size_t)f * H * W * C + c4 * 4;
#pragma unroll
    for (int ty = 0; ty < TY; ++ty)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) {
            if (oy0 + ty >= H || ox0 + tx >= W) continue;
            float4 v = acc[ty][tx];
            v.x = fminf(fmaxf(v.x + b.x, 0.f), 6.f);
            v.y = fminf(fmaxf(v.y + b.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z + b.z, 0.f), 6.f);
            v.w = fminf(fmaxf(v.w + b.w, 0.f), 6.f);
            *(float4 *)(yf + ((size_t)(oy0 + ty) * W + ox0 + tx) * C) = v;
        }
}

__global__ __launch_bounds__(256) void k_subsample(const float *__restrict__ X, float *__restrict__ Y, int n, int H,
                                                   int W, int C) {
    const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t total = (size_t)n * OH * OW * C4;
    if (gid >= total) return;
    int c4 = gid % C4;
    size_t pix = gid / C4;
    int ox = pix % OW, oy = (pix / OW) % OH, f = pix / ((size_t)OW * OH);
    *(float4 *)(Y + gid * 4) = *(const float4 *)(X + (((size_t)f * H + 2 * oy) * W + 2 * ox) * C + c4 * 4);
}

// Y[f][pos][c_off + j] = G[pos][j], j < 16
__global__ __launch_bounds__(256) void k_gauss_fill(const float *__restrict__ G, float *__restrict__ Y, int n,
                                                    int npos, int ldy, int c_off) {
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t total = (size_t)n * npos * 16;
    if (gid >= total) return;
    int j = gid & 15;
    size_t fp = gid >> 4;
    int pos = fp % npos;
    Y[fp * ldy + c_off + j] = G[pos * 16 + j];
}

// bilinear x2 (align_corners=False): X[n][H][W][C] -> Y[n][2H][2W][ldy] channels 0..C-1
__global__ __launch_bounds__(256) SVC_NO_PK void k_upsample2x(const float *__restrict__ X, float *__restrict__ Y, int n,
                                                    int H, int W, int C, int ldy, FDiv dC4, FDiv dOW, FDiv dOH) {
    const int C4 = C >> 2, OH = 2 * H, OW = 2 * W;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = (uint32_t)n * OH * OW * C4;
    if (gid >= total) return;
    uint32_t c4, ox, oy;
    const uint32_t pix = fdivmod(gid, dC4, c4);
    const uint32_t f = fdivmod(fdivmod(pix, dOW, ox), dOH, oy);
    float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float *xf = X + (size_t)f * H * W * C + c4 * 4;
    const float4 v00 = *(const float4 *)(xf + ((size_t)y0 * W + x0) * C);
    const float4 v01 = *(const float4 *)(xf + ((size_t)y0 * W + x1) * C);
    const float4 v10 = *(const float4 *)(xf + ((size_t)y1 * W + x0) * C);
    const float4 v11 = *(const float4 *)(xf + ((size_t)y1 * W + x1) * C);
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    *(float4 *)(Y + (size_t)pix * ldy + c4 * 4) = o;
}

// adaptation: logit[p] = sum_c X[p][c] * w[c] + b, C = 64
__global__ __launch_bounds__(256) void k_adapt(const float *__restrict__ X, const float *__restrict__ w,
                                               const float *__restrict__ b, float *__restrict__ Y, size_t npix,
                                               unsigned *__restrict__ fmax, int n, unsigned long long *__restrict__ census, int ncen) {
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < (size_t)n) fmax[gid] = 0;                     // per-frame running maximum of k_smooth_down (encoded floats)
    // threshold census (svc_threshold_census): the previous pass's per-frame counts are folded into the handle's totals and
    // cleared before this pass's k_quantise counts again (census = [4] u64 totals, then [chunk][4] u32 per-frame counts)
    if (census && gid < (size_t)ncen) {
        unsigned *pf = (unsigned *)(census + 4) + gid * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (pf[j]) { atomicAdd(census + j, (unsigned long long)pf[j]); pf[j] = 0; }
    }
    if (gid >= npix) return;
    const float4 *x = (const float4 *)(X + gid * 64);
    const float4 *w4 = (const float4 *)w;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float4 a = x[i], c = w4[i];
        s = fmaf(a.x, c.x, s);
        s = fmaf(a.y, c.y, s);
        s = fmaf(a.z, c.z, s);
        s = fmaf(a.w, c.w, s);
    }
    Y[gid] = s + b[0];
}

__device__ __forceinline__ unsigned enc_f32(float v) {
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// nearest x8 -> replicate pad 20 -> 41x41 conv, evaluated as 64 phase kernels of 7x7
// low-res taps, then bilinear (align_corners=False) down to (h, w).  One block = one
// frame x rows_per_block output rows; the needed rows of the NHxNW map live in LDS.
// The same on the matrix cores.  For one low-resolution cell (cy, cx) the 64 outputs of its 8 x 8 block are
// 64 different 7x7 kernels ("phases") applied to the same 49 logits: a GEMM [cells x 49] . [49 x 64 phases].  A lane
// owns one cell (B operand: its 49 logits gathered from LDS once, taps padded to 56), the phase kernels are the A
// operand (padded copy in LDS, float4 reads); the accumulator then holds, per lane, 16 phases in runs of four
// consecutive px -- float4 writes into the smoothed-row tile.  The bilinear down-scale and the running maximum are
// those of k_smooth_down.  The sum over the 49 taps is grouped by the MFMA (pairs of taps) instead of one chain of
// FMAs, a difference of a few ulp.
// The bilinear blend  ly0 (lx0 a00 + lx1 a01) + ly1 (lx0 a10 + lx1 a11)  as nine SCALAR VALU instructions (the IEEE operations of
// the C expression, in its order: results unchanged).  Written in C the compiler packs it --
//     v_pk_mul_f32 (lx0 a10, lx1 a01) ; v_pk_mul_f32 (lx0 a00, lx1 a11) ; v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]
// -- and in that form, with workgroups of the bf16-pipe kernels (k_pwr above all) sharing the CU, 1 - 6 % of the passes had the
// product lx1 * a11 MISSING from the sum in lanes 48..63 of one wavefront: the self-check build (-DSD_DEBUG -DSD_PACKED: second
// evaluation + read-back) logged the operands in the registers as correct and the stored value as the sum without that term,
// 17 - 151 events per 1 600 - 2 400 passes in four code shapes (waits and s_nop 7 in front of the packed instructions change
// nothing); with the products or the additions as scalar instructions: 0 of 3 200 passes, twice.  The bare instruction sequence
// in a kernel of its own beside the same co-runners is NOT hit (tools/micro/pk_opsel_victim.hip), so the trigger needs more of
// this kernel's context than the three instructions; what is established is where the value is lost and what removes it
// (profiles/r05_mx_reproducibility.txt, item 7).
// The smoothing kernels carry SVC_NO_PK (no packed f32 instruction at all) -- except in the diagnostic build that is meant to
// reproduce the loss (-DSD_PACKED), which needs the compiler's packed form.
#ifdef SD_PACKED
#define SD_NO_PK
#else
#define SD_NO_PK SVC_NO_PK
#endif
__device__ __forceinline__ float sd_bilinear(float lx0, float lx1, float ly0, float ly1, float a00, float a01, float a10, float a11) {
#ifdef SD_PACKED
    return ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11);      // the compiler's packed form: reproduces the loss
#else
    float p00, p01, p10, p11, r0, r1, q0, q1, v;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p00) : "v"(lx0), "v"(a00));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p01) : "v"(lx1), "v"(a01));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p10) : "v"(lx0), "v"(a10));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p11) : "v"(lx1), "v"(a11));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(p00), "v"(p01));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p10), "v"(p11));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(ly0), "v"(r0));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(ly1), "v"(r1));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(v) : "v"(q0), "v"(q1));
    return v;
#endif
}

#define SD_KP 56          // taps per phase, padded to a multiple of 8
#ifdef SD_DEBUG
// Diagnostic build (make EXTRA="-DSD_DEBUG -DSD_PACKED" OUT=../libsvc_hip_sddbg.so): the bilinear stage evaluates every pixel TWICE
// from LDS and reads its store back; a disagreement is logged (svc_debug_sd_log; tools/soak_network_concurrent.py prints the records).
__device__ unsigned g_sd_count[4];
__device__ float g_sd_log[64][16];
extern "C" int svc_debug_sd_log(unsigned *count4, float *rec64x16) {
    if (hipMemcpyFromSymbol(count4, HIP_SYMBOL(g_sd_count), sizeof(unsigned) * 4) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(rec64x16, HIP_SYMBOL(g_sd_log), sizeof(float) * 64 * 16) != hipSuccess) return -1;
    return 0;
}
#endif
__global__ __launch_bounds__(256) SD_NO_PK void k_smooth_down_mfma(const float *__restrict__ logit, const float *__restrict__ phase,
                                                          float *__restrict__ pre, unsigned *__restrict__ fmax, int LH,
                                                          int LW, int NH, int NW, int h, int w, int rows_per_block,
                                                          int tile_cap, FDiv dw) {
    extern __shared__ float sm[];
    float *L = sm, *ph = sm + LH * LW, *tile = ph + 64 * SD_KP;
    __shared__ unsigned wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int f = blockIdx.y;
    const int oy0 = blockIdx.x * rows_per_block, oy1 = min(h, oy0 + rows_per_block);
    const float scy = (float)NH / (float)h, scx = (float)NW / (float)w;
    const int ylo = (int)fmaxf(scy * (oy0 + 0.5f) - 0.5f, 0.f);
    const int yhi = min((int)fmaxf(scy * ((oy1 - 1) + 0.5f) - 0.5f, 0.f) + 1, NH - 1);
    const int nrows = min(yhi - ylo + 1, tile_cap);
    for (int i = tid; i < LH * LW; i += 256) L[i] = logit[(size_t)f * LH * LW + i];
    for (int i = tid; i < 64 * SD_KP; i += 256) {
        const int p = i / SD_KP, k = i - p * SD_KP;
        ph[i] = k < 49 ? phase[p * 49 + k] : 0.f;
    }
    __syncthreads();
    const int cy_lo = ylo >> 3, cy_hi = (ylo + nrows - 1) >> 3;
    const int ncell = (cy_hi - cy_lo + 1) * LW;
    const int mtiles = (ncell + 31) >> 5;
    for (int mt = wave; mt < mtiles; mt += 4) {
        // this lane's cell and its logits for the taps k = 8q + 4hh + e (clamped at the borders: replicate padding)
        const int cell = min(mt * 32 + r, ncell - 1);
        const int cy = cy_lo + cell / LW, cx = cell % LW;
        float bv[7][4];
#pragma unroll
        for (int q = 0; q < 7; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * q + 4 * hh + e;
                const int a = (k * 37) >> 8, b = k - 7 * a;          // k / 7 for k < 56
                const int yy = min(max(cy + a - 3, 0), LH - 1), xx = min(max(cx + b - 3, 0), LW - 1);
                bv[q][e] = k < 49 ? L[yy * LW + xx] : 0.f;
            }
        const bool live = mt * 32 + r < ncell;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {                             // phases 32 nt .. 32 nt + 31 = py 4 nt .. 4 nt + 3
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const float *ap = ph + (nt * 32 + r) * SD_KP + 4 * hh;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const float4 av = *(const float4 *)(ap + 8 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[q][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[q][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[q][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[q][3], acc, 0, 0, 0);
            }
            if (!live) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {                            // accumulator 4g + j = phase (py = 4 nt + g, px = 4 hh + j)
                const int y = cy * 8 + 4 * nt + g - ylo;
                if (y < 0 || y >= nrows) continue;
                *(float4 *)(tile + y * NW + cx * 8 + 4 * hh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
        }
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int idx = tid; idx < (oy1 - oy0) * w; idx += 256) {
        uint32_t ox;
        const int oy = oy0 + (int)fdivmod((uint32_t)idx, dw, ox);
        float sy = fmaxf(scy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(scx * (ox + 0.5f) - 0.5f, 0.f);
        int y0 = (int)sy, x0 = (int)sx;
        int y1 = y0 + (y0 < NH - 1 ? 1 : 0), x1 = x0 + (x0 < NW - 1 ? 1 : 0);
        float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float *t0 = tile + (y0 - ylo) * NW, *t1 = tile + (y1 - ylo) * NW;
        const float v = sd_bilinear(lx0, lx1, ly0, ly1, t0[x0], t0[x1], t1[x0], t1[x1]);
        pre[((size_t)f * h + oy) * w + ox] = v;
        lmax = fmaxf(lmax, v);
    }
#ifdef SD_DEBUG
    __threadfence();
    __syncthreads();
    for (int idx = tid; idx < (oy1 - oy0) * w; idx += 256) {
        uint32_t ox;
        const int oy = oy0 + (int)fdivmod((uint32_t)idx, dw, ox);
        float sy = fmaxf(scy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(scx * (ox + 0.5f) - 0.5f, 0.f);
        int y0 = (int)sy, x0 = (int)sx;
        int y1 = y0 + (y0 < NH - 1 ? 1 : 0), x1 = x0 + (x0 < NW - 1 ? 1 : 0);
        float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const volatile float *t0 = tile + (y0 - ylo) * NW, *t1 = tile + (y1 - ylo) * NW;
        const float a00 = t0[x0], a01 = t0[x1], a10 = t1[x0], a11 = t1[x1];
        const float v2 = ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11);
        const float rb = __builtin_nontemporal_load(pre + ((size_t)f * h + oy) * w + ox);      // what memory holds now
        atomicAdd(&g_sd_count[3], 1u);
        if (rb != v2) {
            // a third evaluation tells a wrong first evaluation (v3 == v2 != rb) from an unstable one
            const float v3 = ly0 * (lx0 * t0[x0] + lx1 * t0[x1]) + ly1 * (lx0 * t1[x0] + lx1 * t1[x1]);
            const unsigned slot = atomicAdd(&g_sd_count[0], 1u);
            if (slot < 64) {
                float *o = g_sd_log[slot];
                o[0] = (float)f; o[1] = (float)oy; o[2] = (float)ox; o[3] = (float)tid; o[4] = rb; o[5] = v2; o[6] = v3;
                o[7] = a00; o[8] = a01; o[9] = a10; o[10] = a11; o[11] = lx1; o[12] = ly1; o[13] = (float)blockIdx.x;
                o[14] = (float)(__builtin_amdgcn_s_getreg((4 << 0) | (8 << 6) | (3 << 11)));       // HW_ID: cu id bits 8..11
                o[15] = (float)__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11));          // HW_ID: simd id bits 4..5
            }
        }
    }
#endif
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = enc_f32(lmax);
    __syncthreads();
    if (tid == 0) atomicMax(fmax + f, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

__global__ __launch_bounds__(256) SD_NO_PK void k_smooth_down(const float *__restrict__ logit, const float *__restrict__ phase,
                                                     float *__restrict__ pre, unsigned *__restrict__ fmax, int LH,
                                                     int LW, int NH, int NW, int h, int w, int rows_per_block,
                                                     int tile_cap, FDiv dNW, FDiv dw) {
    extern __shared__ float sm[];
    float *L = sm, *ph = sm + LH * LW, *tile = ph + 64 * 49;
    __shared__ unsigned wmax[4];
    const int f = blockIdx.y;
    const int oy0 = blockIdx.x * rows_per_block, oy1 = min(h, oy0 + rows_per_block);
    const float scy = (float)NH / (float)h, scx = (float)NW / (float)w;
    const int ylo = (int)fmaxf(scy * (oy0 + 0.5f) - 0.5f, 0.f);
    const int yhi = min((int)fmaxf(scy * ((oy1 - 1) + 0.5f) - 0.5f, 0.f) + 1, NH - 1);
    const int nrows = min(yhi - ylo + 1, tile_cap);
    for (int i = threadIdx.x; i < LH * LW; i += 256) L[i] = logit[(size_t)f * LH * LW + i];
    for (int i = threadIdx.x; i < 64 * 49; i += 256) ph[i] = phase[i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < nrows * NW; idx += 256) {
        uint32_t x;
        const int ry = (int)fdivmod((uint32_t)idx, dNW, x);
        int y = ylo + ry;
        int cy = y >> 3, py = y & 7, cx = x >> 3, px = x & 7;
        const float *p = ph + (py * 8 + px) * 49;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            int yy = min(max(cy + a - 3, 0), LH - 1);
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                int xx = min(max(cx + b - 3, 0), LW - 1);
                s = fmaf(p[a * 7 + b], L[yy * LW + xx], s);
            }
        }
        tile[idx] = s;
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int idx = threadIdx.x; idx < (oy1 - oy0) * w; idx += 256) {
        uint32_t ox;
        const int oy = oy0 + (int)fdivmod((uint32_t)idx, dw, ox);
        float sy = fmaxf(scy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(scx * (ox + 0.5f) - 0.5f, 0.f);
        int y0 = (int)sy, x0 = (int)sx;
        int y1 = y0 + (y0 < NH - 1 ? 1 : 0), x1 = x0 + (x0 < NW - 1 ? 1 : 0);
        float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float *t0 = tile + (y0 - ylo) * NW, *t1 = tile + (y1 - ylo) * NW;
        const float v = sd_bilinear(lx0, lx1, ly0, ly1, t0[x0], t0[x1], t1[x0], t1[x1]);
        pre[((size_t)f * h + oy) * w + ox] = v;
        lmax = fmaxf(lmax, v);
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = enc_f32(lmax);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(fmax + f, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// u8 = trunc(255 * exp(x - max x)): the softmax normaliser cancels in p / max p.
// census (thr > 0 only): per frame, how many pixels of the UN-thresholded map sit at thr - 1, thr and thr + 1 -- the density of
// the grey-level histogram at the threshold, i.e. how many pixels one grey level of network noise moves across it
// (svc_threshold_census; DESIGN.md 2: 7 per level = windows identical between two correct fp32 implementations, 500 = 21 % differ).
__global__ __launch_bounds__(256) void k_quantise(const float *__restrict__ pre, const unsigned *__restrict__ fmax,
                                                  uint8_t *__restrict__ out, int n, int hw, FDiv dhw, int thr,
                                                  unsigned long long *__restrict__ census, unsigned *__restrict__ rows) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = (uint32_t)n * hw;
    if (gid >= total) return;
    const uint32_t f = fdiv(gid, dhw);
    float m = dec_f32(fmax[f]);
    float e = expf(pre[gid] - m);
    const uint8_t v = (uint8_t)(e * 255.0f);
    out[gid] = (int)v < thr ? (uint8_t)0 : v;              // thr = 0: the plain map; > 0: sc_threshold fused in (svc_saliency_thresholded_u8)
    if (census && thr > 0) {
        const int d = (int)v - (thr - 1);
        if (d >= 0 && d <= 2) {
            atomicAdd((unsigned *)(census + 4) + f * 4 + d, 1u);                           // (a few hundred pixels per map at most)
            if (rows) atomicAdd(rows + f * 4 + d, 1u);                                      // the caller's per-frame rows (svc_saliency_census_u8)
        }
    }
}

// --------------------------------------------------------------------------------------
// plan / workspace
// --------------------------------------------------------------------------------------
enum Buf { B_IN, B_P0, B_P1, B_E0, B_E1, B_F4X, B_F2X, B_S4E, B_S2E, B_CAT1, B_PCD, B_PC, B_CAT2, B_U2E, B_U2D,
           B_U2, B_CAT3, B_P3E, B_P3D, B_DEC, B_LOGIT, B_PRE, B_T1, B_T2, B_COUNT };

struct NetPlan {
    int h = 0, w = 0, NH = 0, NW = 0, nb = 0;
    size_t off[B_COUNT + 1];      // per-frame float offsets
    DevBuf ws, fmax, lut, gauss;
    DevBuf hb, hk, vb, vk;
    int hks = 0, vks = 0, lz_rows = 8, lz_tile_cap = 0;       // output rows per workgroup of k_lanczos_norm (SVC_LZ_ROWS)
    int fr_nr = 0, fr_nc = 0;                                 // source rows / columns one tile of k_front needs at most
    bool fr_ok = false;                                       // the geometry can run k_front
    short fr_row_lo[32], fr_row_n[32], fr_col_lo[32], fr_col_n[32];
    int sd_rows = 7, sd_tile_cap = 0;                         // output rows per workgroup of k_smooth_down (SVC_SD_ROWS)
    int last_n = 0;
    bool last_front = false;       // the last pass ran k_front (the network input is then kept only under SVC_KEEP_INPUT=1)
    int gauss_filled = 0;          // frames of the workspace whose Gaussian-prior channels of CAT1 are already written
    float *buf(int b) const { return (float *)ws.p + off[b] * (size_t)nb; }
    size_t per_frame(int b) const { return off[b + 1] - off[b]; }
};

static void optimal_out_size(int h, int w, int &NH, int &NW) {      // data.py:1086-1103
    double ar = (double)h / w, best = -1.0;
    int b1 = 8, b2 = 13;
    for (int n1 = 7; n1 < 14; ++n1)
        for (int n2 = 7; n2 < 14; ++n2)
            if (n1 * n2 >= 100 && n1 * n2 <= 120) {
                double t = (double)n1 / n2;
                double ratio = std::min(ar, t) / std::max(ar, t);
                if (ratio > best) { best = ratio; b1 = n1; b2 = n2; }
            }
    NH = b1 * 32;
    NW = b2 * 32;
}

// torch.linspace(0, 1, steps) in float32 (symmetric evaluation around the midpoint)
static void linspace01(int steps, std::vector<float> &v) {
    v.resize(steps);
    if (steps == 1) { v[0] = 0.f; return; }
    float step = 1.0f / (float)(steps - 1);
    int half = steps / 2;
    for (int i = 0; i < steps; ++i) v[i] = (i < half) ? (0.f + step * i) : (1.f - step * (steps - 1 - i));
}

static int build_plan(SvcHandle *h, int height, int width, int nb) {
    NetPlan *p = h->plan;
    if (!p) p = h->plan = new NetPlan();
    if (p->h == height && p->w == width && p->nb >= nb) return SVC_OK;
    p->gauss_filled = 0;            // the workspace layout (and possibly the maps) change below
    int NH, NW;
    optimal_out_size(height, width, NH, NW);
    const bool same_size = (p->h == height && p->w == width);
    if (const char *e = getenv("SVC_LZ_ROWS")) if (atoi(e) > 0) p->lz_rows = atoi(e);
    if (const char *e = getenv("SVC_SD_ROWS")) if (atoi(e) > 0) p->sd_rows = atoi(e);
    p->h = height; p->w = width; p->NH = NH; p->NW = NW; p->nb = nb;
    const size_t H1 = NH / 2, W1 = NW / 2, H2 = NH / 4, W2 = NW / 4, H3 = NH / 8, W3 = NW / 8, H4 = NH / 16,
                 W4 = NW / 16, H5 = NH / 32, W5 = NW / 32;
    size_t sz[B_COUNT];
    sz[B_IN] = (size_t)NH * NW * 3;
    sz[B_P0] = sz[B_P1] = H1 * W1 * 32;
    // expanded tensors: f2 expand at H1 (96 ch) is the largest
    sz[B_E0] = std::max(std::max(H1 * W1 * 96, H2 * W2 * 144), std::max(H3 * W3 * 192, std::max(H4 * W4 * 576, H5 * W5 * 960)));
    sz[B_E1] = std::max(std::max(H1 * W1 * 32, H2 * W2 * 144), std::max(H3 * W3 * 192, std::max(H4 * W4 * 576, H5 * W5 * 960)));
    sz[B_F4X] = H3 * W3 * 64;  sz[B_F2X] = H4 * W4 * 160;
    sz[B_S4E] = H3 * W3 * 128; sz[B_S2E] = H4 * W4 * 320;
    sz[B_CAT1] = H5 * W5 * 1296; sz[B_PCD] = H5 * W5 * 1296; sz[B_PC] = H5 * W5 * 256;
    sz[B_CAT2] = H4 * W4 * 384; sz[B_U2E] = H4 * W4 * 768; sz[B_U2D] = H4 * W4 * 768; sz[B_U2] = H4 * W4 * 128;
    sz[B_CAT3] = H3 * W3 * 192; sz[B_P3E] = H3 * W3 * 384; sz[B_P3D] = H3 * W3 * 384; sz[B_DEC] = H3 * W3 * 64;
    sz[B_T1] = H5 * W5 * 768; sz[B_T2] = H4 * W4 * 384;   // low-resolution halves of the two decoder expansions
    sz[B_LOGIT] = (H3 * W3 + 3) / 4 * 4;
    sz[B_PRE] = ((size_t)height * width + 3) / 4 * 4;
    p->off[0] = 0;
    for (int b = 0; b < B_COUNT; ++b) p->off[b + 1] = p->off[b] + (sz[b] + 3) / 4 * 4;
    int rc = p->ws.ensure(p->off[B_COUNT] * (size_t)nb * sizeof(float));
    if (rc) return rc;
    rc = p->fmax.ensure((size_t)nb * sizeof(unsigned));
    if (rc) return rc;
    if (same_size) return SVC_OK;
    // resampling tables
    std::vector<int> hb, hk, vb, vk;
    lanczos_tab(width, NW, hb, hk, p->hks);
    lanczos_tab(height, NH, vb, vk, p->vks);
    int cap = 0;
    for (int y0 = 0; y0 < NH; y0 += p->lz_rows) {
        int y1 = std::min(NH, y0 + p->lz_rows);
        cap = std::max(cap, vb[2 * (y1 - 1)] + vb[2 * (y1 - 1) + 1] - vb[2 * y0]);
    }
    p->lz_tile_cap = cap;
    // k_front: source rows / columns of every tile row / column.  The kernel serves up-scaling geometries (at most
    // eight taps per pass) whose source patch is at most 32 rows of 32 words; anything else runs the three kernels.
    p->fr_nr = p->fr_nc = 0;
    const int fty = ceil_div(NH / 2, FR_TH), ftx = ceil_div(NW / 2, FR_TW);
    p->fr_ok = p->hks <= 8 && p->vks <= 8 && fty <= FR_MAXT && ftx <= FR_MAXT;
    for (int t = 0; t < FR_MAXT; ++t) p->fr_row_lo[t] = p->fr_row_n[t] = p->fr_col_lo[t] = p->fr_col_n[t] = 0;
    for (int t = 0; t < fty && p->fr_ok; ++t) {
        const int ya = std::max(2 * t * FR_TH - 3, 0), yb = std::min(2 * t * FR_TH - 3 + FR_PR, NH);
        p->fr_row_lo[t] = (short)vb[2 * ya];
        p->fr_row_n[t] = (short)(vb[2 * (yb - 1)] + vb[2 * (yb - 1) + 1] - vb[2 * ya]);
        p->fr_nr = std::max(p->fr_nr, (int)p->fr_row_n[t]);
    }
    for (int t = 0; t < ftx && p->fr_ok; ++t) {
        const int xa = std::max(2 * t * FR_TW - 3, 0), xb = std::min(2 * t * FR_TW - 3 + FR_PC, NW);
        p->fr_col_lo[t] = (short)hb[2 * xa];
        p->fr_col_n[t] = (short)(hb[2 * (xb - 1)] + hb[2 * (xb - 1) + 1] - hb[2 * xa]);
        p->fr_nc = std::max(p->fr_nc, (int)p->fr_col_n[t]);
    }
    p->fr_ok = p->fr_ok && p->fr_nr <= 32 && front_src_stride(p->fr_nc) <= 128 &&
               front_small_bytes(p->fr_nr, p->fr_nc) <= FR_SONLY_BYTES;
    if ((rc = p->hb.ensure(hb.size() * 4)) || (rc = p->hk.ensure(hk.size() * 4)) || (rc = p->vb.ensure(vb.size() * 4)) ||
        (rc = p->vk.ensure(vk.size() * 4)))
        return rc;
    SVC_HIP(hipMemcpy(p->hb.p, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    SVC_HIP(hipMemcpy(p->hk.p, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    SVC_HIP(hipMemcpy(p->vb.p, vb.data(), vb.size() * 4, hipMemcpyHostToDevice));
    SVC_HIP(hipMemcpy(p->vk.p, vk.data(), vk.size() * 4, hipMemcpyHostToDevice));
    // ToTensor (/255) + Normalize LUT, float32 arithmetic like torchvision
    std::vector<float> lut(3 * 256);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float t = (float)v / 255.0f;
            volatile float d = t - mean[c];
            lut[c * 256 + v] = d / sd[c];
        }
    if ((rc = p->lut.ensure(lut.size() * 4))) return rc;
    SVC_HIP(hipMemcpy(p->lut.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice));
    // Gaussian prior maps [H5][W5][16]  (model.py:348-378, float32)
    std::vector<float> ys, xs, g(H5 * W5 * 16);
    linspace01((int)H5, ys);
    linspace01((int)W5, xs);
    for (int i = 0; i < 16; ++i) {
        const float *gp = h->gauss_params.data() + i * 4;      // [y/x][mu/logstd]
        float sy = expf(gp[1]), sx = expf(gp[3]);
        for (size_t y = 0; y < H5; ++y)
            for (size_t x = 0; x < W5; ++x) {
                float dy = (ys[y] - gp[0]) / sy, dx = (xs[x] - gp[2]) / sx;
                float m = 1.0f;
                m *= expf(-(dy * dy) / 2.0f);
                m *= expf(-(dx * dx) / 2.0f);
                g[(y * W5 + x) * 16 + i] = m * 6.0f;
            }
    }
    if ((rc = p->gauss.ensure(g.size() * 4))) return rc;
    SVC_HIP(hipMemcpy(p->gauss.p, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    // tile rows for k_smooth_down
    float scy = (float)NH / (float)height;
    cap = 0;
    for (int oy0 = 0; oy0 < height; oy0 += p->sd_rows) {
        int oy1 = std::min(height, oy0 + p->sd_rows);
        int ylo = (int)std::max(scy * (oy0 + 0.5f) - 0.5f, 0.f);
        int yhi = std::min((int)std::max(scy * ((oy1 - 1) + 0.5f) - 0.5f, 0.f) + 1, NH - 1);
        cap = std::max(cap, yhi - ylo + 1);
    }
    p->sd_tile_cap = cap;
    return SVC_OK;
}

int svc_net_release(SvcHandle *h) {
    if (h->plan) {
        NetPlan *p = h->plan;
        p->ws.release(); p->fmax.release(); p->lut.release(); p->gauss.release();
        p->hb.release(); p->hk.release(); p->vb.release(); p->vk.release();
        delete p;
        h->plan = nullptr;
    }
    for (auto &kv : h->cvtabs) kv.second.release();
    h->cvtabs.clear();
    return SVC_OK;
}

// --------------------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------------------
static inline unsigned blocks256(size_t total) { return (unsigned)((total + 255) / 256); }

// Lane-order copy of a weight matrix for k_pw_sk<.., true>: out[((st * tiles + tile) * 64 + lane)] (float4) =
// W[tile * 32 + (lane & 31)][8 st + 4 (lane >> 5) .. + 3] -- what the lane loads in k-step st for column tile `tile`.
__global__ void k_lane_weights(const float *__restrict__ Wt, int ldw, int nsteps, int tiles, int nrows, float4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nsteps * tiles * 64) return;
    const int lane = (int)(i & 63), tile = (int)((i >> 6) % tiles), st = (int)((i >> 6) / tiles);
    out[i] = *(const float4 *)(Wt + (size_t)min(tile * 32 + (lane & 31), nrows - 1) * ldw + 8 * st + 4 * (lane >> 5));     // rows beyond the matrix repeat its last one
}

// the copy is made on the stream of the first launch that needs it (ordered in front of that launch, and waited for) and kept with the handle
static int lane_weights(SvcHandle *h, hipStream_t s, const float *Wt, int ldw, int K, int Npad, const float **out) {
    const auto key = std::make_tuple((const void *)Wt, ldw, K, Npad);
    auto it = h->lane_w.find(key);
    if (it == h->lane_w.end()) {
        DevBuf b;
        const int nsteps = K >> 3, tiles = (Npad + 31) >> 5;
        int rc = b.ensure((size_t)nsteps * tiles * 64 * sizeof(float4));
        if (rc) return rc;
        k_lane_weights<<<blocks256((size_t)nsteps * tiles * 64), 256, 0, s>>>(Wt, ldw, nsteps, tiles, Npad, (float4 *)b.p);
        SVC_CHECK_LAUNCH();
        // once per matrix and handle, on the first pass that needs it:
        // finished before any other stream can be handed the copy
        SVC_HIP(hipStreamSynchronize(s));
        it = h->lane_w.emplace(key, b).first;
    }
    *out = (const float *)it->second.p;
    return SVC_OK;
}

// Split-bf16 copies of a weight matrix (see x3_split): three round-to-nearest bf16 planes per element, 48 B per (row, 16-deep
// step, lane half).  Two orders:
//   X3_ROWS   out[((n * Q + q) * 2 + hh) * 3 + plane]                 rows of Q * 6 uint4 -- staged to LDS by k_pwr / k_irb, read
//                                                                     straight from L2 by k_pwpw's first GEMM
//   X3_LANES  out[(((q * tiles + tile) * 64) + lane) * 3 + plane]     what lane (r, hh) of column tile `tile` feeds into step q:
//                                                                     a wave's load is 3 KB contiguous (k_pw_sk, k_dwpw, k_pwpw)
// k beyond K reads as zero (K = 24: the second step's upper half); rows beyond nrows repeat the last one (as k_lane_weights).
enum { X3_ROWS = 0, X3_LANES = 1 };
__global__ void k_x3_weights(const float *__restrict__ Wt, int ldw, int K, int Q, int tiles, int nrows, int order, uint4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one (row, step, half)
    if (i >= (size_t)tiles * 32 * Q * 2) return;
    int n, q, hh;
    if (order == X3_ROWS) { hh = (int)(i & 1); q = (int)((i >> 1) % Q); n = (int)((i >> 1) / Q); }
    else { const int lane = (int)(i & 63); hh = lane >> 5; const size_t t = i >> 6; n = (int)(t % tiles) * 32 + (lane & 31); q = (int)(t / tiles); }
    const float *w = Wt + (size_t)min(n, nrows - 1) * ldw;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * q + 8 * (j >> 2) + 4 * hh + (j & 3);
        v[j] = k < K ? w[k] : 0.f;
    }
    const X3 s = x3_split<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    X3Q H, Mi, L;
    H.v = s.h; Mi.v = s.m; L.v = s.l;
    out[i * 3] = H.q; out[i * 3 + 1] = Mi.q; out[i * 3 + 2] = L.q;
}

// made on the stream of the first launch that needs it, waited for, kept with the handle (like lane_weights)
static int x3_weights(SvcHandle *h, hipStream_t s, const float *Wt, int ldw, int K, int Npad, int order, const uint4 **out) {
    const auto key = std::make_tuple((const void *)Wt, ldw, K, Npad * 2 + order);
    auto it = h->x3_w.find(key);
    if (it == h->x3_w.end()) {
        DevBuf b;
        const int Q = (K + 15) >> 4, tiles = (Npad + 31) >> 5;
        const size_t cells = (size_t)tiles * 32 * Q * 2;
        int rc = b.ensure(cells * 3 * sizeof(uint4));
        if (rc) return rc;
        k_x3_weights<<<blocks256(cells), 256, 0, s>>>(Wt, ldw, K, Q, tiles, Npad, order, (uint4 *)b.p);
        SVC_CHECK_LAUNCH();
        SVC_HIP(hipStreamSynchronize(s));
        it = h->x3_w.emplace(key, b).first;
    }
    *out = (const uint4 *)it->second.p;
    return SVC_OK;
}

// One pointwise layer, or a column slice of one: K of the ldw input channels of the weight rows, starting at Wt
// (bias may be null).  ups != null adds the up-sampled low-resolution product (see UpsAdd).
static int launch_pw_ex(SvcHandle *h, hipStream_t s, const float *X, int ldx, const float *Wt, int ldw, int K,
                        const float *bias, int relu6v, int N, const float *R, int ldr, float *Y, int ldy, int M, int n,
                        const UpsAdd *ups) {
    if (h->seg_off >> h->seg_cur & 1u) return SVC_OK;
    ProfScope ps(h, SVC_K_PW, s);
    const int Npad = (N + 31) / 32 * 32, tiles = Npad / 32;
    UpsAdd ua = ups ? *ups : UpsAdd{nullptr, 0, 0, 0, make_fdiv(1), make_fdiv(1)};
    const int rb = ceil_div(M, 128);
    int TN = 4;
    while (TN > 1 && (TN > tiles || rb * ceil_div(tiles, TN) < h->pw_min_wg)) --TN;
    dim3 grid(rb, ceil_div(tiles, TN));
    // Which kernel FAMILY runs (and with it the order of the K sum) depends only on the layer shape,
    // judged at a nominal batch of 32 frames, never on the batch actually passed: a frame's map
    // must not depend on its batch.  The tile shape (TN) may follow the real M, it does not change
    // any sum order.
    const int rb_nom = ceil_div((M / n) * 32, 128);
#define PW16_ARGS X, ldx, Wt, ldw, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v
    // short K, no residual: activations resident in registers, weight chunk shared through LDS (k_pwr)
    if (h->pwr && !R && (K == 64 || K == 96 || K == 128 || K == 160) && (N % 4) == 0) {
        // column tiles per workgroup: enough workgroups (at the nominal batch) to put two or three on every CU
        int ntw = h->pwr_nt;
        if (ntw <= 0) {
            ntw = 4;
            while (ntw > 1 && rb_nom * ceil_div(tiles, ntw) < h->pwr_min_wg) --ntw;
        }
        ntw = std::min(ntw, tiles);
        const dim3 g(rb, ceil_div(tiles, ntw));
        const bool mx = h->mx != 0 && (h->mx_mask & 1);
        const float *Wk = Wt;                                // MX: the split-bf16 copy of the slice, rows of (K / 16) * 6 uint4
        if (mx) {
            const uint4 *W3 = nullptr;
            int rc = x3_weights(h, s, Wt, ldw, K, Npad, X3_ROWS, &W3);
            if (rc) return rc;
            Wk = (const float *)W3;
        }
        const size_t lds = mx ? (size_t)ntw * 32 * ((K / 16) * 6 + 1) * sizeof(uint4) + ((size_t)4 * 32 * PWR_SLAB + 128) * sizeof(float)
                              : ((size_t)ntw * 32 * (K + 4) + 4 * 32 * PWR_SLAB + 128) * sizeof(float);
#define PWR_ARGS X, ldx, Wk, ldw, bias, Y, ldy, M, N, Npad, ntw, relu6v, ua
#define PWR_CASE(KSv)                                                                                                   \
    {                                                                                                                   \
        auto kfn = mx ? (ups ? k_pwr<KSv, true, true> : k_pwr<KSv, false, true>)                                        \
                      : (ups ? k_pwr<KSv, true> : k_pwr<KSv, false>);                                                   \
        if (h->lds_attr_done.insert((const void *)kfn).second)                                                          \
            SVC_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));    \
        kfn<<<g, 256, lds, s>>>(PWR_ARGS);                                                                              \
    }
        switch (K) {
            case 64: PWR_CASE(8) break;
            case 96: PWR_CASE(12) break;
            case 128: PWR_CASE(16) break;
            default: PWR_CASE(20) break;
        }
#undef PWR_CASE
#undef PWR_ARGS
        SVC_CHECK_LAUNCH();
        return SVC_OK;
    }
    if (!ups && h->pw_sk && K >= 256 && rb_nom * tiles <= h->pw_sk_max) {   // long K, few workgroups: split K over the four waves
        const int tn = (tiles % 2 == 0) ? 2 : 1;
        dim3 g(ceil_div(M, 32), ceil_div(tiles, tn));
        if (h->mx && (h->mx_mask & 8) && (K & 15) == 0) {
            const uint4 *W3 = nullptr;
            int rc = x3_weights(h, s, Wt, ldw, K, Npad, X3_LANES, &W3);
            if (rc) return rc;
            if (tn == 2) k_pw_sk<2, true, true><<<g, 256, 0, s>>>(X, ldx, (const float *)W3, tiles, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v);
            else k_pw_sk<1, true, true><<<g, 256, 0, s>>>(X, ldx, (const float *)W3, tiles, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v);
        } else if (h->sk_lane && (K & 7) == 0) {
            const float *Wl = nullptr;
            int rc = lane_weights(h, s, Wt, ldw, K, Npad, &Wl);
            if (rc) return rc;
            if (tn == 2) k_pw_sk<2, true><<<g, 256, 0, s>>>(X, ldx, Wl, tiles, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v);
            else k_pw_sk<1, true><<<g, 256, 0, s>>>(X, ldx, Wl, tiles, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v);
        } else if (tn == 2) k_pw_sk<2><<<g, 256, 0, s>>>(PW16_ARGS);
        else k_pw_sk<1><<<g, 256, 0, s>>>(PW16_ARGS);
        SVC_CHECK_LAUNCH();
        return SVC_OK;
    }
    // small-M layers (8x13 and 16x26 levels at B = 32) leave most SIMDs idle with 32x32 wave tiles:
    // 16x32 wave tiles (SVC_PW_SMALL selects the shape) give 2-4x more waves
    if (!ups && h->pw_small && K % 16 == 0 && K >= 64 && rb_nom * tiles < 1024) {
        if (h->pw_small == 1) {
            k_pw16<1, 2><<<dim3(ceil_div(M, 64), tiles), 256, 0, s>>>(PW16_ARGS);
        } else if (h->pw_small == 2) {
            k_pw16<1, 4><<<dim3(ceil_div(M, 64), ceil_div(tiles, 2)), 256, 0, s>>>(PW16_ARGS);
        } else {
            k_pw16<2, 2><<<dim3(ceil_div(M, 128), tiles), 256, 0, s>>>(PW16_ARGS);
        }
        SVC_CHECK_LAUNCH();
        return SVC_OK;
    }
    // measured on MI355X: the 16x16x4 form wins for single-N-tile layers with a short K (the
    // high-resolution project layers), the 32x32x2 form everywhere else
    if (!ups && h->pw16 && tiles == 1 && K % 16 == 0 && K >= 64 && K <= 192) {
        k_pw16<2, 2><<<dim3(rb, 1), 256, 0, s>>>(PW16_ARGS);
        SVC_CHECK_LAUNCH();
        return SVC_OK;
    }
#undef PW16_ARGS
#define PW_ARGS X, ldx, Wt, ldw, bias, R, ldr, Y, ldy, M, N, Npad, K, relu6v, ua
    if (h->pw_tr == 1 || (h->pw_tr == 2 && (TN >= 2 || ups))) {
        switch (TN) {
            case 4: k_pw<4, 1, true><<<grid, 256, 0, s>>>(PW_ARGS); break;
            case 3: k_pw<3, 1, true><<<grid, 256, 0, s>>>(PW_ARGS); break;
            case 2: k_pw<2, 1, true><<<grid, 256, 0, s>>>(PW_ARGS); break;
            default: k_pw<1, 1, true><<<grid, 256, 0, s>>>(PW_ARGS); break;
        }
    } else {
        switch (TN) {
            case 4: k_pw<4, 1><<<grid, 256, 0, s>>>(PW_ARGS); break;
            case 3: k_pw<3, 1><<<grid, 256, 0, s>>>(PW_ARGS); break;
            case 2: k_pw<2, 1><<<grid, 256, 0, s>>>(PW_ARGS); break;
            default: k_pw<1, 1><<<grid, 256, 0, s>>>(PW_ARGS); break;
        }
    }
#undef PW_ARGS
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

static int launch_pw(SvcHandle *h, hipStream_t s, const float *X, int ldx, const SvcLayer &L, const float *R, int ldr,
                     float *Y, int ldy, int M, int n) {
    return launch_pw_ex(h, s, X, ldx, L.w.dev, L.cin, L.cin, L.b.dev, L.relu6, L.cout, R, ldr, Y, ldy, M, n, nullptr);
}

static int launch_dw(SvcHandle *h, hipStream_t s, const float *X, const SvcLayer &L, float *Y, int n, int H, int W,
                     int stride) {
    if (h->seg_off >> h->seg_cur & 1u) return SVC_OK;
    ProfScope ps(h, SVC_K_DW, s);
    const int C = L.cout, OH = stride == 2 ? H / 2 : H, OW = stride == 2 ? W / 2 : W;
    size_t total = (size_t)n * OH * OW * (C / 4);
    if (stride == 2)
        k_dw<2><<<blocks256(total), 256, 0, s>>>(X, L.w.dev, L.b.dev, Y, n, H, W, C, OH, OW, make_fdiv(C / 4), make_fdiv(OW),
                                                 make_fdiv(OH));
    else if (h->dw_tile) {
        const int tx = h->dw_tile / 10, ty = h->dw_tile % 10;
        const int GX = (W + tx - 1) / tx, GY = (H + ty - 1) / ty;
        const size_t cells = (size_t)n * GY * GX * (C / 4);
#define DW_TILE(TXv, TYv)                                                                                             \
    k_dw_tile<TXv, TYv><<<blocks256(cells), 256, 0, s>>>(X, L.w.dev, L.b.dev, Y, n, H, W, C, make_fdiv(C / 4),        \
                                                         make_fdiv(GX), make_fdiv(GY), (uint32_t)cells)
        switch (h->dw_tile) {
            case 21: DW_TILE(2, 1); break;
            case 22: DW_TILE(2, 2); break;
            case 41: DW_TILE(4, 1); break;
            case 44: DW_TILE(4, 4); break;
            default: DW_TILE(4, 2); break;
        }
#undef DW_TILE
    } else
        k_dw<1><<<blocks256(total), 256, 0, s>>>(X, L.w.dev, L.b.dev, Y, n, H, W, C, OH, OW, make_fdiv(C / 4), make_fdiv(OW),
                                                 make_fdiv(OH));
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

// --------------------------------------------------------------------------------------
// Depthwise 3x3 (+BN+ReLU6) fused into the 1x1 project that follows it (the second half of an
// inverted-residual block, MobileNetV2.py:48-55,66-73, and of the three decoder blocks).  The
// depthwise output -- as large as the 6x expanded tensor -- never reaches memory.
//
// One workgroup = one patch of 32 pixels (PW x 32/PW) of one frame x NT*32 output channels.  Its four
// waves split the K (channel) range in 32-channel chunks, chunk c -> wave c % 4, and each wave works
// alone, without barriers: it computes the depthwise output of its chunk for the 32 pixels (a lane owns
// 4 consecutive pixels x 4 channels: 18 tap loads + 9 weights for 4 outputs, like k_dw_tile), parks the
// 32 x 32 tile in its private LDS slab in MFMA A layout, and multiplies it with the chunk's slice of the
// project weights (B straight from global/L2) into NT accumulators.  The four partial sums meet in LDS at
// the end and are added in wave order, so the result is deterministic and independent of the batch.
// --------------------------------------------------------------------------------------
#define IRB_ES 36      // LDS row stride (floats) of 32-channel tiles: 32 + 4 pad (conflict-free float4 rows)
#ifndef IRB_XREG
#define IRB_XREG 1     // 0: the pixel operand of k_irb's expand GEMM staged in LDS (rounds 1-3), for A/B runs of tools/micro/irb_time.hip
#endif
#ifndef IRB_WAVES
#define IRB_WAVES 4    // minimum waves per SIMD the register allocation of k_irb has to allow (4: <= 128 VGPRs, four workgroups per CU)
#endif
#ifndef IRB_STAMP
#define IRB_STAMP(i)   // phase stamps of k_irb: defined by tools/micro/irb_phases.hip only (no code in the product)
#endif

// MX (round 5): the project GEMM on split-bf16 operands (x3_split): Wp is the X3_LANES copy (lw_tiles column tiles), the wave's
// depthwise tile is split as it is read back from the slab (each element once per workgroup), a 32-channel chunk = two 16-deep steps.
template <int NT, int PW, int NWV, bool MX = false>     // NWV = waves per workgroup = ways of the K split (8 was measured: no gain); two waves per SIMD: the NT = 5 instances sit at 256 registers
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_dwpw(const float *__restrict__ X, int H, int W, int C,
                                              const float *__restrict__ Wd, const float *__restrict__ bd,
                                              const float *__restrict__ Wp, const float *__restrict__ bp, int N,
                                              int Npad, const float *__restrict__ R, int ldr, float *__restrict__ Y,
                                              int ldy, int relu6, int tiles_x, int tiles_y, int lw_tiles) {
    // lw_tiles > 0: Wp is the LANE-ORDER copy of the project weights (lane_weights: one contiguous KB per wave load instead of
    // 32 rows x 32 B; lw_tiles = its column tiles) -- the project-weight loads were more than half of the cache lines this
    // kernel touches
    constexpr int PH = 32 / PW;
    // one LDS block: the waves' depthwise slabs [NWV][32 x IRB_ES], then the K partials [NWV][16][64]
    __shared__ float smem_dwpw[NWV * 32 * IRB_ES + NWV * 16 * 64];
    float (*Dw)[32 * IRB_ES] = (float (*)[32 * IRB_ES])smem_dwpw;
    float (*red)[16][64] = (float (*)[16][64])(smem_dwpw + NWV * 32 * IRB_ES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int bid = xcd_bx();
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, f = bid / tiles_y;
    const int x0 = tx * PW, y0 = ty * PH, n0 = blockIdx.y * (NT * 32);
    // depthwise role of the lane: channel group c4 of the chunk, pixels p0 .. p0+3 of the patch (one row)
    const int c4 = lane & 7, p0 = (lane >> 3) * 4;
    const int py = p0 / PW, pxs = p0 % PW;
    const int oy = y0 + py, ox0 = x0 + pxs;
    const float *xf = X + (size_t)f * H * W * C;
    float *D = Dw[wave];
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nchunks = (C + 31) >> 5;
    // Latency, not bandwidth, bounds this kernel (round-1 form: every tap load in its own bounds-check branch followed by
    // s_waitcnt vmcnt(0), and the project weights of a k-step requested only when its MFMAs were next: a full L2 round
    // trip per 8-deep k-step).  So: (1) the chunk's slice of the project weights is requested FIRST and parked in
    // registers through the depthwise phase (NT <= 2: the whole 32-deep slice; wider tiles keep one k-step in flight
    // ahead of the MFMAs); (2) the 18 tap loads are unconditional -- coordinates clamped into the image, out-of-image
    // taps zeroed by a select -- so they are issued back to back.  A zeroed tap adds 0 * w: the sums are unchanged.
    constexpr bool PRE = false;        // (the whole 32-deep slice in registers costs the second wave per SIMD: 22 -> 33 us at NT = 2)
    const float *wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        wrow[t] = lw_tiles ? Wp + ((size_t)min((int)blockIdx.y * NT + t, lw_tiles - 1) * 64 + lane) * 4
                           : Wp + (size_t)min(n0 + t * 32 + r, Npad - 1) * C + 4 * hh;
    const size_t wstep = lw_tiles ? (size_t)lw_tiles * 256 : 8;      // floats from one k-step's float4 to the next
    for (int ch = wave; ch < nchunks; ch += NWV) {
        const int kend = min(32, C - ch * 32);
        float4 Bw[PRE ? NT : 1][4];
        if (PRE) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    Bw[PRE ? t : 0][q] = *(const float4 *)(wrow[t] + wstep * (ch * 4 + min(q, (kend >> 3) - 1)));      // (a short last chunk re-reads its last k-step: unused)
        }
        const int c = ch * 32 + c4 * 4;
        const bool cok = c < C;
        const int cc = cok ? c : 0;
        float4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            float4 w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *(const float4 *)(Wd + (size_t)t * C + cc);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy - 1 + ky, iyc = min(max(iy, 0), H - 1);
                const bool yok = iy >= 0 && iy < H;
                float4 row[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int ix = ox0 - 1 + j, ixc = min(max(ix, 0), W - 1);
                    if (NT >= 5) {                          // five accumulator tiles: keep the loads conditional (fewer live registers: 108 instead of 187 VGPRs)
                        row[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (yok && ix >= 0 && ix < W) row[j] = *(const float4 *)(xf + ((size_t)iy * W + ix) * C + cc);
                    } else {
                        row[j] = *(const float4 *)(xf + ((size_t)iyc * W + ixc) * C + cc);
                    }
                }
                if (NT < 5) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int ix = ox0 - 1 + j;
                        if (!(yok && ix >= 0 && ix < W)) row[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        fma4(o[j], row[j + kx], w[ky * 3 + kx]);
                    }
            }
            const float4 b = *(const float4 *)(bd + cc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j].x = cok ? fminf(fmaxf(o[j].x + b.x, 0.f), 6.f) : 0.f;
                o[j].y = cok ? fminf(fmaxf(o[j].y + b.y, 0.f), 6.f) : 0.f;
                o[j].z = cok ? fminf(fmaxf(o[j].z + b.z, 0.f), 6.f) : 0.f;
                o[j].w = cok ? fminf(fmaxf(o[j].w + b.w, 0.f), 6.f) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4 *)(D + (p0 + j) * IRB_ES + c4 * 4) = o[j];
        // the slab is private to the wave: its own LDS writes are visible to it once they have completed
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const float *ap = D + r * IRB_ES + 4 * hh;
        if constexpr (MX) {
            const uint4 *w3 = (const uint4 *)Wp + (size_t)lane * 3;
            const size_t w3step = (size_t)lw_tiles * 64 * 3;             // uint4 from one step to the next
            for (int q = 0; q < (kend >> 4); ++q) {
                X3 w[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    w[t] = x3_load(w3 + w3step * (ch * 2 + q) + (size_t)min((int)blockIdx.y * NT + t, lw_tiles - 1) * (64 * 3));
                const X3 a = x3_split(*(const float4 *)(ap + 16 * q), *(const float4 *)(ap + 16 * q + 8));
#pragma unroll
                for (int t = 0; t < NT; ++t) x3_mma(acc[t], w[t], a);
            }
        } else if (PRE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (8 * q < kend) {
                    const float4 a = *(const float4 *)(ap + 8 * q);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 b = Bw[PRE ? t : 0][q];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[t], 0, 0, 0);   // swapped: lane = pixel
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[t], 0, 0, 0);
                    }
                }
            }
        } else if (NT >= 5) {                               // five tiles: weights requested per k-step (no second register set)
            for (int k = 0; k < kend; k += 8) {
                const float4 a = *(const float4 *)(ap + k);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 b = *(const float4 *)(wrow[t] + wstep * (ch * 4 + (k >> 3)));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[t], 0, 0, 0);
                }
            }
        } else {
            float4 nb[NT];                                  // the next k-step's weights, requested before this step's MFMAs
#pragma unroll
            for (int t = 0; t < NT; ++t) nb[t] = *(const float4 *)(wrow[t] + wstep * (ch * 4));
            for (int k = 0; k < kend; k += 8) {
                const float4 a = *(const float4 *)(ap + k);
                float4 b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = nb[t];
                if (k + 8 < kend) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) nb[t] = *(const float4 *)(wrow[t] + wstep * (ch * 4 + (k >> 3) + 1));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].x, a.x, acc[t], 0, 0, 0);   // swapped: lane = pixel
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].y, a.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].z, a.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].w, a.w, acc[t], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                    // the next chunk overwrites the slab
    }
    // sum of the NWV K partials (wave order), bias, residual, store.  With the operands swapped a lane holds pixel r
    // of the patch and accumulator i = channel 8(i>>2) + 4hh + (i&3) of the tile: wave w finishes the channel
    // runs g = w (and w + 4 ... when NWV < 4 does not apply: NWV is 4 or 8), one float4 per lane.
    const int yy = y0 + r / PW, xx = x0 + r % PW;
    const bool live = yy < H && xx < W;
    const size_t pix = ((size_t)f * H + yy) * W + xx;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[t][i];
        __syncthreads();
        if (wave >= 4) continue;                            // (8-wave form: the upper waves only contribute partials)
        const int g = wave;
        const int col = n0 + t * 32 + 8 * g + 4 * hh;
        if (col >= N || !live) continue;
        float4 v = make_float4(red[0][4 * g][lane], red[0][4 * g + 1][lane], red[0][4 * g + 2][lane], red[0][4 * g + 3][lane]);
#pragma unroll
        for (int q = 1; q < NWV; ++q) {
            v.x += red[q][4 * g][lane]; v.y += red[q][4 * g + 1][lane];
            v.z += red[q][4 * g + 2][lane]; v.w += red[q][4 * g + 3][lane];
        }
        const float4 bv = *(const float4 *)(bp + col);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (R) {
            const float4 rv = *(const float4 *)(R + pix * ldr + col);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (relu6) {
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
        }
        *(float4 *)(Y + pix * ldy + col) = v;
    }
}

// --------------------------------------------------------------------------------------
// Two 1x1 convolutions in a row with a ReLU6 between them (the skip branches, model.py:443-444: 160 -> 320 -> 128 on the
// 16x26 level, 64 -> 128 -> 64 on the 32x52 level) as ONE launch: the intermediate tensor (17 + 27 MB per 32 frames, written
// by one kernel and read back by the next) never exists.  k_dwpw's structure with an MFMA producer in place of the depthwise
// one: a workgroup = 32 pixels, its four waves split the INTERMEDIATE channels in chunks of 32 (chunk c -> wave (c + block) % 4);
// per chunk a wave computes the 32-channel x 32-pixel tile of the first convolution (weights as the A operand, the pixel's
// K1 inputs resident in registers, k_pwr's k order), adds the bias, clamps -- and the accumulator IS the B operand the
// second convolution wants (a lane owns one pixel and channels 8g + 4hh + j of the chunk: the k pairs of k_pwr's steps), so
// the tile goes straight into NT2 x 16 MFMAs against the chunk's columns of the second weight matrix: no LDS between the
// two GEMMs.  The four waves' partial sums of the second convolution meet in LDS and are added in chunk-group order
// (deterministic, independent of the batch; the order differs from k_pwr's / k_pw_sk's sequential sum: fp32 rounding only).
// --------------------------------------------------------------------------------------
// MX (round 5): both GEMMs on split-bf16 operands (x3_split): W1 is the X3_ROWS copy ([Cm][K1 / 16][2][3] uint4: a lane's step is 48
// contiguous bytes), W2 the X3_LANES copy; the pixel's inputs are split once in the prologue, the intermediate tile -- the first
// GEMM's accumulator after bias and ReLU6 -- is split in registers and is the second GEMM's B operand as before.
template <int KS1, int NT2, bool MX = false>     // K1 = 8 KS1 input channels, N2 = 32 NT2 output channels
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_pwpw(const float *__restrict__ X, int ldx, const float *__restrict__ W1,
                                              const float *__restrict__ b1, int Cm, const float *__restrict__ W2,
                                              const float *__restrict__ b2, float *__restrict__ Y, int ldy, int M, int lw) {
    // lw: W2 is a LANE-ORDER copy (lane_weights: a wave's weight load is one contiguous KB instead of 32 rows x 32 B)
    constexpr int K1 = 8 * KS1;
    __shared__ float red_pp[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int m0 = xcd_bx() * 32;
    const float *xp = X + (size_t)min(m0 + r, M - 1) * ldx + 4 * hh;
    constexpr int Q1 = KS1 / 2;
    static_assert(KS1 % 2 == 0, "K1 must be a multiple of 16");
    float4 A1[MX ? 1 : KS1];
    X3 A13[MX ? Q1 : 1];
    if constexpr (MX) {
#pragma unroll
        for (int q = 0; q < Q1; ++q) A13[q] = x3_split(*(const float4 *)(xp + 16 * q), *(const float4 *)(xp + 16 * q + 8));
    } else {
#pragma unroll
        for (int p = 0; p < KS1; ++p) A1[MX ? 0 : p] = *(const float4 *)(xp + 8 * p);
    }
    f32x16 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
    const float *w2row[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) w2row[t] = lw ? W2 + ((size_t)t * 64 + lane) * 4 : W2 + (size_t)(t * 32 + r) * Cm + 4 * hh;
    const int nchunks = Cm >> 5;
    const size_t step2 = lw ? (size_t)NT2 * 256 : 8;       // floats from one k-step's float4 to the next
    if constexpr (MX) {
        const uint4 *W13 = (const uint4 *)W1, *W23 = (const uint4 *)W2 + (size_t)lane * 3;
        for (int ch = (wave + 4 - (blockIdx.x & 3)) & 3; ch < nchunks; ch += 4) {
            // second-convolution weights of the chunk's first step: requested in front of the first convolution's MFMAs
            X3 nb[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) nb[t] = x3_load(W23 + ((size_t)(ch * 2) * NT2 + t) * (64 * 3));
            const uint4 *w1p = W13 + ((size_t)(ch * 32 + r) * Q1) * 6 + 3 * hh;
            f32x16 e;
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = 0.f;
            X3 wn = x3_load(w1p);
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                const X3 b = wn;
                if (q + 1 < Q1) wn = x3_load(w1p + 6 * (q + 1));
                x3_mma(e, b, A13[MX ? q : 0]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4 *)(b1 + ch * 32 + 8 * g + 4 * hh);
                e[4 * g] = fminf(fmaxf(e[4 * g] + bv.x, 0.f), 6.f);
                e[4 * g + 1] = fminf(fmaxf(e[4 * g + 1] + bv.y, 0.f), 6.f);
                e[4 * g + 2] = fminf(fmaxf(e[4 * g + 2] + bv.z, 0.f), 6.f);
                e[4 * g + 3] = fminf(fmaxf(e[4 * g + 3] + bv.w, 0.f), 6.f);
            }
            // e[8q .. 8q + 7] = intermediate channels ch*32 + 16q + 8(j>>2) + 4hh + (j&3): the B fragment of the second GEMM's step q
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                X3 b[NT2];
#pragma unroll
                for (int t = 0; t < NT2; ++t) b[t] = nb[t];
                if (q == 0) {
#pragma unroll
                    for (int t = 0; t < NT2; ++t) nb[t] = x3_load(W23 + ((size_t)(ch * 2 + 1) * NT2 + t) * (64 * 3));
                }
                const X3 a = x3_split(make_float4(e[8 * q], e[8 * q + 1], e[8 * q + 2], e[8 * q + 3]),
                                      make_float4(e[8 * q + 4], e[8 * q + 5], e[8 * q + 6], e[8 * q + 7]));
#pragma unroll
                for (int t = 0; t < NT2; ++t) x3_mma(acc2[t], b[t], a);
            }
        }
    } else
    for (int ch = (wave + 4 - (blockIdx.x & 3)) & 3; ch < nchunks; ch += 4) {
        // second-convolution weights of the chunk's first k-steps: requested in front of the first convolution's MFMAs
        float4 nb[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t) nb[t] = *(const float4 *)(w2row[t] + step2 * (ch * 4));
        const float *w1p = W1 + (size_t)(ch * 32 + r) * K1 + 4 * hh;
        f32x16 e;
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = 0.f;
        float4 wn = *(const float4 *)w1p;
#pragma unroll
        for (int p = 0; p < KS1; ++p) {
            const float4 b = wn;
            if (p + 1 < KS1) wn = *(const float4 *)(w1p + 8 * (p + 1));
            e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, A1[MX ? 0 : p].x, e, 0, 0, 0);      // swapped: lane = pixel (k_pwr's order)
            e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, A1[MX ? 0 : p].y, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, A1[MX ? 0 : p].z, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, A1[MX ? 0 : p].w, e, 0, 0, 0);
        }
        // e[4g + j] = intermediate channel ch*32 + 8g + 4hh + j of pixel r: bias, ReLU6 -- and it is the B operand of the
        // second convolution's k-step g
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *(const float4 *)(b1 + ch * 32 + 8 * g + 4 * hh);
            e[4 * g] = fminf(fmaxf(e[4 * g] + bv.x, 0.f), 6.f);
            e[4 * g + 1] = fminf(fmaxf(e[4 * g + 1] + bv.y, 0.f), 6.f);
            e[4 * g + 2] = fminf(fmaxf(e[4 * g + 2] + bv.z, 0.f), 6.f);
            e[4 * g + 3] = fminf(fmaxf(e[4 * g + 3] + bv.w, 0.f), 6.f);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) b[t] = nb[t];
            if (g < 3) {
#pragma unroll
                for (int t = 0; t < NT2; ++t) nb[t] = *(const float4 *)(w2row[t] + step2 * (ch * 4 + g + 1));
            }
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].x, e[4 * g], acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].y, e[4 * g + 1], acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].z, e[4 * g + 2], acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t].w, e[4 * g + 3], acc2[t], 0, 0, 0);
            }
        }
    }
    // sum of the four K partials (wave order), bias, store: wave g finishes the channel runs 8g + 4hh .. + 3 of every tile
    const bool live = m0 + r < M;
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red_pp[wave][i][lane] = acc2[t][i];
        __syncthreads();
        const int g = wave, col = t * 32 + 8 * g + 4 * hh;
        if (!live) continue;
        // (the partials are added in the order of the chunk groups 0, 1, 2, 3 -- group k sits with wave (k + block) % 4 -- so the
        // rotation that balances the SIMDs does not reach the result: a pixel's sum does not depend on its place in the batch)
        const int w0 = blockIdx.x & 3;
        float4 v = make_float4(red_pp[w0][4 * g][lane], red_pp[w0][4 * g + 1][lane], red_pp[w0][4 * g + 2][lane], red_pp[w0][4 * g + 3][lane]);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const int wq = (w0 + q) & 3;
            v.x += red_pp[wq][4 * g][lane]; v.y += red_pp[wq][4 * g + 1][lane];
            v.z += red_pp[wq][4 * g + 2][lane]; v.w += red_pp[wq][4 * g + 3][lane];
        }
        const float4 bv = *(const float4 *)(b2 + col);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        *(float4 *)(Y + (size_t)(m0 + r) * ldy + col) = v;
    }
}

// the pair qualifies: K1 = 64 or 160, intermediate channels a multiple of 32, 64 or 128 outputs, ReLU6 between, none after
static bool pwpw_takes(const SvcHandle *h, const SvcLayer &L1, const SvcLayer &L2) {
    return h->pwpw && (L1.cin == 64 || L1.cin == 160) && (L1.cout % 32) == 0 && L2.cin == L1.cout && (L2.cout == 64 || L2.cout == 128) &&
           L1.relu6 && !L2.relu6;
}

static int launch_pwpw(SvcHandle *h, hipStream_t s, const float *X, int ldx, const SvcLayer &L1, const SvcLayer &L2, float *Y, int ldy,
                       int M) {
    if (h->seg_off >> h->seg_cur & 1u) return SVC_OK;
    ProfScope ps(h, SVC_K_PW, s);
    const dim3 grid((unsigned)ceil_div(M, 32));
    const float *W1 = L1.w.dev, *W2 = L2.w.dev;
    int lw = 0;
    const bool mx = h->mx && (h->mx_mask & 16);
    if (mx) {
        const uint4 *a = nullptr, *b = nullptr;
        int rc = x3_weights(h, s, L1.w.dev, L1.cin, L1.cin, L1.cout, X3_ROWS, &a);
        if (!rc) rc = x3_weights(h, s, L2.w.dev, L2.cin, L2.cin, L2.cout, X3_LANES, &b);
        if (rc) return rc;
        W1 = (const float *)a; W2 = (const float *)b;
        lw = 1;
    } else if (h->sk_lane) {
        // (the SECOND matrix only: the first one's loads run one step ahead of their MFMAs, and in [Cm][K1] order three of
        // four hit the line the one before them fetched -- in lane order every one is an L2 round trip: 44 -> 57 us at K1 = 160)
        int rc = lane_weights(h, s, L2.w.dev, L2.cin, L2.cin, L2.cout, &W2);
        if (rc) return rc;
        lw = 1;
    }
#define PWPW_ARGS X, ldx, W1, L1.b.dev, L1.cout, W2, L2.b.dev, Y, ldy, M, lw
    if (mx) {
        if (L1.cin == 64) {
            if (L2.cout == 64) k_pwpw<8, 2, true><<<grid, 256, 0, s>>>(PWPW_ARGS);
            else k_pwpw<8, 4, true><<<grid, 256, 0, s>>>(PWPW_ARGS);
        } else {
            if (L2.cout == 64) k_pwpw<20, 2, true><<<grid, 256, 0, s>>>(PWPW_ARGS);
            else k_pwpw<20, 4, true><<<grid, 256, 0, s>>>(PWPW_ARGS);
        }
    } else if (L1.cin == 64) {
        if (L2.cout == 64) k_pwpw<8, 2><<<grid, 256, 0, s>>>(PWPW_ARGS);
        else k_pwpw<8, 4><<<grid, 256, 0, s>>>(PWPW_ARGS);
    } else {
        if (L2.cout == 64) k_pwpw<20, 2><<<grid, 256, 0, s>>>(PWPW_ARGS);
        else k_pwpw<20, 4><<<grid, 256, 0, s>>>(PWPW_ARGS);
    }
#undef PWPW_ARGS
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

static int launch_dwpw(SvcHandle *h, hipStream_t s, const float *X, const SvcLayer &Ld, const SvcLayer &Lp, const float *R,
                       int ldr, float *Y, int ldy, int n, int H, int W) {
    if (h->seg_off >> h->seg_cur & 1u) return SVC_OK;
    ProfScope ps(h, SVC_K_PW, s);
    const int C = Ld.cout, N = Lp.cout, Npad = (N + 31) / 32 * 32, tiles = Npad / 32;
    // output-channel groups of at most 5 tiles, as even as possible (the depthwise part is redone per group)
    const int groups = ceil_div(tiles, std::min(5, std::max(1, h->dwpw_max_nt))), nt = ceil_div(tiles, groups);
    const int pw = (W % 8 == 0 || W > 16) ? 8 : 16;          // 8x4 patches; 16x2 on the narrow 13-wide level
    const int tx = ceil_div(W, pw), ty = ceil_div(H, 32 / pw);
    dim3 grid((unsigned)(n * tx * ty), groups);
    const float *Wpl = Lp.w.dev;                             // the project weights: lane-order copy where the knob allows
    int lw_tiles = 0;
    const bool mx = h->mx && (h->mx_mask & 4) && (C & 15) == 0;
    if (mx) {
        const uint4 *W3 = nullptr;
        int rc = x3_weights(h, s, Lp.w.dev, C, C, Npad, X3_LANES, &W3);
        if (rc) return rc;
        Wpl = (const float *)W3;
        lw_tiles = tiles;
    } else if (h->sk_lane && (C & 7) == 0) {
        int rc = lane_weights(h, s, Lp.w.dev, C, C, Npad, &Wpl);
        if (rc) return rc;
        lw_tiles = tiles;
    }
#define DWPW_ARGS X, H, W, C, Ld.w.dev, Ld.b.dev, Wpl, Lp.b.dev, N, Npad, R, ldr, Y, ldy, Lp.relu6, tx, ty, lw_tiles
#define DWPW_CASE(NTv)                                                                    \
    case NTv:                                                                             \
        if (mx) {                                                                         \
            if (pw == 8) k_dwpw<NTv, 8, 4, true><<<grid, 256, 0, s>>>(DWPW_ARGS);         \
            else k_dwpw<NTv, 16, 4, true><<<grid, 256, 0, s>>>(DWPW_ARGS);                \
        } else if (pw == 8) k_dwpw<NTv, 8, 4><<<grid, 256, 0, s>>>(DWPW_ARGS);            \
        else k_dwpw<NTv, 16, 4><<<grid, 256, 0, s>>>(DWPW_ARGS);                          \
        break
    switch (nt) {
        DWPW_CASE(1); DWPW_CASE(2); DWPW_CASE(3); DWPW_CASE(4);
        default: DWPW_CASE(5);
    }
#undef DWPW_CASE
#undef DWPW_ARGS
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

// --------------------------------------------------------------------------------------
// Fused inverted-residual block (MobileNetV2.py:26-83): 1x1 expand + ReLU6 -> 3x3 depthwise
// + ReLU6 -> 1x1 project (+ residual) for one tile of TOH x TOW output pixels.  The 6x
// expanded tensor and the depthwise output never leave the CU: per 32-channel chunk of the
// expansion the workgroup (4 waves) computes
//   E  = relu6(Xs . We^T + be)   for the (TOH-1)S+3 x (TOW-1)S+3 input halo   (f32 MFMA, A and B from LDS)
//   D  = relu6(dw3x3(E) + bd)                                                 (VALU, float4 over channels)
//   acc += D . Wp^T                                                           (f32 MFMA, A and B from LDS)
// with the same k order and tap order as k_pw / k_dw, so results are bit-identical to the
// un-fused kernels.  Out-of-image halo pixels hold E = 0 (the depthwise conv pads E, not X).
// EXPAND = false is the t=1 block (features.1): E is the input itself.
//
// Weights: the three weight slices of a chunk are a few KB.  They are fetched from global memory (L2)
// with one coalesced float4 per thread a full phase before they are needed, parked in registers across
// that phase, and written to LDS behind the barrier that retires their previous readers:
//   We[ch+1], Wp[ch]  requested before expand(ch), stored after barrier 1 (readers: expand(ch) / project(ch-1))
//   Wd[ch+1]          requested before expand(ch), stored after barrier 2 (readers: depthwise(ch))
// so no MFMA k step or depthwise tap ever waits on an L2 round trip, and the register cost is 4 float4.
// --------------------------------------------------------------------------------------

template <int S, int TOH, int TOW>
struct IrbGeom {
    static constexpr int IH = (TOH - 1) * S + 3, IW = (TOW - 1) * S + 3, NPX = IH * IW, MT = (NPX + 31) / 32;
    static constexpr int NOUT = TOH * TOW;
    // floats of LDS; rows NPX..MT*32-1 of Xs are never written: the expand MFMA reads whatever lies behind them
    // (the E array, so still inside the allocation) into accumulator rows that are discarded.  Every byte counts:
    // three workgroups per CU need <= 53 KB each.
    static size_t lds_floats(int Cin, int CoutP, bool expand, int Ce = 0, bool xreg = false, bool mx = false) {
        const size_t XS = Cin + 4;
        size_t n = (xreg ? 0 : (size_t)NPX * XS) + (expand ? (size_t)NPX * IRB_ES : 0) + (size_t)NOUT * IRB_ES;
        // mx: the expand slice as split-bf16 rows (uint4, one pad); the project slice too where the output is four tiles (CoutP = 64)
        if (mx) n += 32 * (size_t)(((Cin + 15) / 16) * 6 + 1) * 4 + (CoutP == 64 ? (size_t)CoutP * 13 * 4 : (size_t)CoutP * IRB_ES) + 9 * 32;
        else n += (expand ? 32 * XS : 0) + (size_t)CoutP * IRB_ES + 9 * 32;
        n += 2 * (size_t)((Ce + 31) / 32 * 32);             // expand / depthwise biases of every chunk
        return n;
    }
};

// CIN / CE / COUT > 0 fix the channel counts at compile time (the MobileNetV2 blocks this kernel serves have six
// distinct shapes): strides, trip counts and the slice bookkeeping fold to constants, which matters because the
// kernel is bound by instruction issue.  0 = take them from the arguments (any other shape).
// MX (round 5; the fixed-shape expanding instances): the EXPAND GEMM on the bf16 matrix pipe with split operands (x3_split).  We is
// then the X3_ROWS copy of the matrix (K = Cin), a chunk's slice is staged in LDS in that form (rows of 6 uint4 per 16-deep step +
// one pad), and the lane's halo pixel is split once in the prologue: a tile's K = Cin costs Q1 * 6 MFMAs of 32 cycles instead of
// Cin / 2 of 64.  The PROJECT GEMM takes that form (MXP) only where the output is four 32 x 32 tiles (block 7: Cout = 64): its
// pixel operand is the depthwise output, which would have to be split (44 VALU instructions per 16-deep step and wave) on the
// phase's critical path, and where the output is one or two tiles the waves share a chunk's k range -- a 16-deep step cannot be
// shared four ways.  Measured (profiles/r05_mx_knobs.txt): both GEMMs split made blocks 2-6 SLOWER than the fp32 form (+28 us on
// block 2), block 7 slightly faster.
template <int S, int TOH, int TOW, bool EXPAND, int CIN = 0, int CE = 0, int COUT = 0, bool MX = false>
// (the split slices of block 7 leave room for three workgroups per CU, not four: its register budget follows)
// (the other split-bf16 instances spill 2 / 5 registers at four waves per SIMD; at three, without spills, the pass is 1.265 ms alone /
// 0.877 shared against 1.255 / 0.878: kept at four)
#define IRB_MX3(S_, CIN_, COUT_, MX_) ((MX_) && (COUT_) == 64)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IRB_MX3(S, CIN, COUT, MX) ? 3 : IRB_WAVES, 8))) void k_irb(const float *__restrict__ X, int H, int W, int Cin_,
                                             const float *__restrict__ We, const float *__restrict__ be, int Ce_,
                                             const float *__restrict__ Wd, const float *__restrict__ bd,
                                             const float *__restrict__ Wp, const float *__restrict__ bp, int Cout_,
                                             int CoutP_, const float *__restrict__ R, float *__restrict__ Y, int ldy_,
                                             int OH, int OW, int tiles_x, int tiles_y) {
    const int Cin = CIN > 0 ? CIN : Cin_, Ce = CE > 0 ? CE : Ce_, Cout = COUT > 0 ? COUT : Cout_;
    const int CoutP = COUT > 0 ? (COUT + 31) / 32 * 32 : CoutP_, ldy = COUT > 0 ? COUT : ldy_;
    using G = IrbGeom<S, TOH, TOW>;
    constexpr int IW = G::IW, NPX = G::NPX, MT = G::MT;
    constexpr int NOUT = G::NOUT, MP = NOUT / 32;
    extern __shared__ float sm_irb[];
    const int XS = Cin + 4;
    // XREG (round 4; the fixed-shape expanding instances): the pixel operand of the expand GEMM -- the lane's halo pixel, Cin
    // floats -- does not depend on the chunk, so it is loaded ONCE from global memory into registers (Cin / 8 float4 per
    // expand tile of the wave) instead of being staged in LDS and re-read for every chunk: no Xs array (11 - 14 KB less LDS:
    // four workgroups per CU instead of three where the rest fits in 40 KB), no prologue fill, 3 - 4 ds_read_b128 less per
    // chunk and wave.  Same values, same k order: bit-identical.
    constexpr bool XREG = IRB_XREG && EXPAND && CIN > 0 && CIN <= 32 && (CIN % 8) == 0;
    static_assert(!MX || XREG, "the split-bf16 form exists for the fixed-shape expanding instances");
    constexpr bool MXP = MX && COUT == 64;                              // the project GEMM on split operands too (Wp = its X3_ROWS copy)
    constexpr int Q1 = MX ? (CIN + 15) / 16 : 1, RS1 = Q1 * 6 + 1;      // MX: 16-deep steps of the expand GEMM, LDS row stride (uint4) of its slice
    constexpr int QP = MXP ? (CE + 15) / 16 : 1;                        // ... steps in a row of the project matrix's copy
    float *Xs = sm_irb;                                     // [NPX][XS]   (not with XREG)
    float *E = EXPAND ? Xs + (XREG ? 0 : NPX * XS) : Xs;    // [NPX][IRB_ES]   (t=1: XS == IRB_ES, E is Xs)
    float *D = E + NPX * IRB_ES;                            // [NOUT][IRB_ES]
    float *Wes = D + NOUT * IRB_ES;                         // [32][XS]        expand weights of the chunk        (MX: uint4 [32][RS1])
    float *Wps = Wes + (MX ? 32 * RS1 * 4 : (EXPAND ? 32 * XS : 0));    // [CoutP][IRB_ES] project weights of the chunk      (MX: uint4 [CoutP][13])
    float *Wds = Wps + (MXP ? CoutP * 13 * 4 : CoutP * IRB_ES);         // [9][32]         depthwise weights of the chunk
    uint4 *Wes3 = (uint4 *)Wes, *Wps3 = (uint4 *)Wps;
    float *Bes = Wds + 9 * 32;                              // [CeP]           expand biases (all chunks): no global load inside the chunk loop
    float *Bds = Bes + ((Ce + 31) / 32 * 32);               // [CeP]           depthwise biases
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int bid = xcd_bx();
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, f = bid / tiles_y;
    const int oy0 = ty * TOH, ox0 = tx * TOW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int c4n = Cin >> 2;
    const int nchunks = (Ce + 31) >> 5;
    // weight slices as seen by this thread (one float4 each; the project slice has CoutP*8 float4 <= 4 per thread)
    const int we_n = MX ? (32 * Q1 * 6 + 255) >> 8 : EXPAND ? (32 * c4n + 255) >> 8 : 0;      // float4 per thread in the expand slice (<= 3: Cin <= 96)
    const bool wd_mine = tid < 72;
    const int wd_t = tid >> 3, wd_c4 = tid & 7;
    const int wp_n = MXP ? (CoutP * 12 + 255) >> 8 : (CoutP * 8 + 255) >> 8;                  // float4 per thread in the project slice
    auto load_we = [&](int ch, int q) -> float4 {
        const int idx = tid + q * 256;                       // the slice is 32 rows x Cin floats (MX: x Q1 * 6 uint4), contiguous in We
        if constexpr (MX) return (q < we_n && idx < 32 * Q1 * 6) ? *((const float4 *)We + (size_t)ch * 32 * Q1 * 6 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        return (q < we_n && idx < 32 * c4n) ? *(const float4 *)(We + (size_t)ch * 32 * Cin + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_we = [&](int q, float4 v) {
        const int idx = tid + q * 256;
        if constexpr (MX) {
            if (q < we_n && idx < 32 * Q1 * 6) { const int row = idx / (Q1 * 6), c = idx - row * (Q1 * 6); *((float4 *)Wes + row * RS1 + c) = v; }
            return;
        }
        if (q < we_n && idx < 32 * c4n) {
            const int row = idx / c4n, c4 = idx - row * c4n;
            *(float4 *)(Wes + row * XS + c4 * 4) = v;
        }
    };
    auto load_wd = [&](int ch) -> float4 {
        const int c = ch * 32 + wd_c4 * 4;
        return (wd_mine && c < Ce) ? *(const float4 *)(Wd + (size_t)wd_t * Ce + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_wp = [&](int ch, int q) -> float4 {
        if constexpr (MXP) {                                  // the chunk's two steps of every row: 12 uint4 per row
            const int idx = tid + q * 256, row = idx / 12, c = idx - row * 12, st = 2 * ch + c / 6;
            return (q < wp_n && row < CoutP && st < QP) ? *((const float4 *)Wp + ((size_t)row * QP + st) * 6 + (c % 6)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int idx = tid + q * 256, row = idx >> 3, k4 = idx & 7;
        const int k = ch * 32 + k4 * 4;
        return (q < wp_n && row < CoutP && k < Ce) ? *(const float4 *)(Wp + (size_t)row * Ce + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    {
        // 1. input halo and the first weight slices -> LDS (zeros outside the image)
        const float *xf = X + (size_t)f * H * W * Cin;
        if (!XREG)
        for (int idx = tid; idx < NPX * c4n; idx += 256) {
            const int row = idx / c4n, c4 = idx - row * c4n;
            const int hy = row / IW, hx = row - hy * IW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *(const float4 *)(xf + ((size_t)iy * W + ix) * Cin + c4 * 4);
            *(float4 *)(Xs + row * XS + c4 * 4) = v;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) store_we(q, load_we(0, q));
        if (wd_mine) *(float4 *)(Wds + wd_t * 32 + wd_c4 * 4) = load_wd(0);
    }
    // XREG: the lane's halo pixel(s), Cin floats each, straight from global memory into registers (requested before the
    // prologue's barrier; rows beyond the halo and pixels outside the image read a clamped address: their E rows are
    // masked in the expand epilogue)
    constexpr int XK = XREG ? CIN / 8 : 1, XT = XREG ? (G::MT + 3) / 4 : 1;
    float4 xa[XT][XK];
    if constexpr (XREG) {
        const float *xf = X + (size_t)f * H * W * Cin;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int mt = (G::MT == 5 && u == 1) ? 4 : wave + 4 * u;
            const int rr = min(mt * 32 + r, NPX - 1);
            const int hy = rr / IW, hx = rr - hy * IW;
            const int iy = min(max(iy0 + hy, 0), H - 1), ix = min(max(ix0 + hx, 0), W - 1);
            const float *xp = xf + ((size_t)iy * W + ix) * Cin + 4 * hh;
#pragma unroll
            for (int q = 0; q < XK; ++q) xa[u][q] = *(const float4 *)(xp + 8 * q);
        }
    }
    for (int i = tid; i < ((Ce + 31) / 32 * 32); i += 256) {
        if (EXPAND) Bes[i] = i < Ce ? be[i] : 0.f;
        Bds[i] = i < Ce ? bd[i] : 0.f;
    }
    X3 xs[MX ? XT : 1][MX ? Q1 : 1];
    if constexpr (MX) {
#pragma unroll
        for (int u = 0; u < XT; ++u)
#pragma unroll
            for (int q = 0; q < Q1; ++q)
                xs[u][q] = x3_split(xa[u][2 * q], 2 * q + 1 < XK ? xa[u][2 * q + 1 < XK ? 2 * q + 1 : 0] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    // project: the NOUT x CoutP output is MP x NT tiles of 32 x 32.  With four or more tiles every wave owns one
    // or two of them over the whole contraction; with fewer (Cout <= 32) the waves also split each chunk's k range
    // KS ways, keep partial accumulators, and the partials are summed in a fixed order in the epilogue -- otherwise
    // two or three of the four SIMDs would idle through half of the block's MFMA work.
    const int NT = CoutP >> 5, ptiles = MP * NT;
    const int KS = ptiles >= 4 ? 1 : 4 / ptiles, kw = 32 / KS;
    const int t0 = KS > 1 ? wave % ptiles : wave, ks = KS > 1 ? wave / ptiles : 0;
    // MFMA operands are swapped throughout (weights as A, pixels as B): a lane owns ONE pixel of the tile and 16
    // channels in runs of four, so bias / ReLU6 / masking and the LDS or global writes of the epilogues move float4.
    // Per expand tile of this wave: is the lane's row a halo pixel at all (row_ok), and inside the image (in_img)?
    constexpr int MTW = (MT + 3) / 4;                     // expand tiles per wave
    bool row_ok[MTW], in_img[MTW];
#pragma unroll
    for (int u = 0; u < MTW; ++u) {
        const int mt = (MT == 5 && u == 1) ? 4 : wave + 4 * u;    // MT == 5: the odd tile rotates over the waves
        const int rr = mt * 32 + r;
        const int hy = rr / IW, hx = rr - hy * IW;
        row_ok[u] = mt < MT && rr < NPX;
        in_img[u] = row_ok[u] && (unsigned)(iy0 + hy) < (unsigned)H && (unsigned)(ix0 + hx) < (unsigned)W;
    }
    IRB_STAMP(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        IRB_STAMP(1);
        const int kend = min(32, Ce - ch * 32);
        // requests for the slices needed one phase (Wp) or one chunk (We, Wd) from now
        float4 s_wp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s_wp[q] = load_wp(ch, q);
        const bool more = ch + 1 < nchunks;
        float4 s_we[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) s_we[q] = more ? load_we(ch + 1, q) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 s_wd = more ? load_wd(ch + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (EXPAND) {
#pragma unroll
            for (int u = 0; u < MTW; ++u) {
                int mt = wave + 4 * u;
                if (MT == 5 && u == 1) mt = (wave == (ch & 3)) ? 4 : MT;
                if (mt >= MT) break;
                f32x16 e;
#pragma unroll
                for (int i = 0; i < 16; ++i) e[i] = 0.f;
                if constexpr (MX) {
                    const uint4 *bq = Wes3 + r * RS1 + 3 * hh;
#pragma unroll
                    for (int q = 0; q < Q1; ++q) x3_mma(e, x3_load(bq + 6 * q), xs[MX ? u : 0][MX ? q : 0]);
                } else {
                const float *ap = Xs + (mt * 32 + r) * XS + 4 * hh;
                const float *bq = Wes + r * XS + 4 * hh;
#pragma unroll
                for (int k = 0; k < Cin; k += 8) {
                    float4 a;
                    if constexpr (XREG) a = xa[u][k >> 3]; else a = *(const float4 *)(ap + k);
                    const float4 b = *(const float4 *)(bq + k);
                    e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, e, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, e, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, e, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, e, 0, 0, 0);
                }
                }
                // lane = halo pixel mt*32 + r; e[4g + j] = channel ch*32 + 8g + 4hh + j
                if (row_ok[u]) {
                    float *ep = E + (mt * 32 + r) * IRB_ES + 4 * hh;
                    const bool in = in_img[u];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = ch * 32 + 8 * g + 4 * hh;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (in && c < Ce) {                 // channels beyond Ce (zero weight rows) stay 0, like out-of-image pixels
                            const float4 bv = *(const float4 *)(Bes + c);
                            v.x = fminf(fmaxf(e[4 * g] + bv.x, 0.f), 6.f);
                            v.y = fminf(fmaxf(e[4 * g + 1] + bv.y, 0.f), 6.f);
                            v.z = fminf(fmaxf(e[4 * g + 2] + bv.z, 0.f), 6.f);
                            v.w = fminf(fmaxf(e[4 * g + 3] + bv.w, 0.f), 6.f);
                        }
                        *(float4 *)(ep + 8 * g) = v;
                    }
                }
            }
            IRB_STAMP(2);
            __syncthreads();                                // barrier 1: E complete; expand(ch), project(ch-1) retired
            IRB_STAMP(3);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            if constexpr (MXP) {
                if (q < wp_n && idx < CoutP * 12) *((float4 *)Wps + (idx / 12) * 13 + idx % 12) = s_wp[q];
            } else if (q < wp_n && (idx >> 3) < CoutP) *(float4 *)(Wps + (idx >> 3) * IRB_ES + (idx & 7) * 4) = s_wp[q];
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < 3; ++q) store_we(q, s_we[q]);
        }
        // depthwise 3x3 on the chunk.  The phase is bound by LDS bandwidth (18 float4 reads per output pixel x 4 channels
        // when every tap and its weight are read per output), so a thread produces two horizontally adjacent pixels from
        // four (stride 2: five) columns per row and keeps its nine weights in registers: 21 (24) reads for two outputs
        // instead of 36.  The sums run in the same tap order as before.
        if ((TOW & 1) == 0) {
            for (int idx = tid; idx < (NOUT / 2) * 8; idx += 256) {
                // Which pixel pair a thread takes decides the LDS banks its 16-lane group touches: eight lanes read one pixel's
                // 32 channels (32 consecutive banks), the other eight another pixel's -- conflict-free only when the two are
                // 32 banks (mod 64) apart.  Neighbouring pairs of a row are 72 (stride 2: 144) floats apart = 8 (16) banks: a
                // two-way conflict on most of every read.  Rows 4 apart (stride 1: 4 x 10 x 36 floats) are 32 banks apart: with
                // them in one group SQ_LDS_BANK_CONFLICT of the stride-1 blocks halves (6.0 -> 3.2 M cycles per launch of block 3,
                // LDS-active cycles -14 %).  (Stride 2: pairs 4 outputs apart are 32 banks apart too; measured: no change.)
                const int c4 = idx & 7;
                int oy, ox;
                if (S == 1 && TOH == 8 && TOW == 8) { const int rest = idx >> 4; oy = (rest >> 2) + 4 * ((idx >> 3) & 1); ox = 2 * (rest & 3); }
                else { const int pp = idx >> 3; oy = pp / (TOW / 2); ox = 2 * (pp - oy * (TOW / 2)); }
                const int c = ch * 32 + c4 * 4;
                float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
                if (c < Ce) {
                    float4 w[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) w[t] = *(const float4 *)(Wds + t * 32 + c4 * 4);
                    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float *ep = E + ((oy * S + ky) * IW + ox * S) * IRB_ES + c4 * 4;
                        float4 x[3 + S];
#pragma unroll
                        for (int j = 0; j < 3 + S; ++j) x[j] = *(const float4 *)(ep + j * IRB_ES);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float4 wk = w[ky * 3 + kx];
                            fma4(a0, x[kx], wk);
                            fma4(a1, x[kx + S], wk);
                        }
                    }
                    const float4 b = *(const float4 *)(Bds + c);
                    o0.x = fminf(fmaxf(a0.x + b.x, 0.f), 6.f);
                    o0.y = fminf(fmaxf(a0.y + b.y, 0.f), 6.f);
                    o0.z = fminf(fmaxf(a0.z + b.z, 0.f), 6.f);
                    o0.w = fminf(fmaxf(a0.w + b.w, 0.f), 6.f);
                    o1.x = fminf(fmaxf(a1.x + b.x, 0.f), 6.f);
                    o1.y = fminf(fmaxf(a1.y + b.y, 0.f), 6.f);
                    o1.z = fminf(fmaxf(a1.z + b.z, 0.f), 6.f);
                    o1.w = fminf(fmaxf(a1.w + b.w, 0.f), 6.f);
                }
                float *dp = D + (oy * TOW + ox) * IRB_ES + c4 * 4;
                *(float4 *)dp = o0;
                *(float4 *)(dp + IRB_ES) = o1;
            }
        } else
        for (int idx = tid; idx < NOUT * 8; idx += 256) {
            const int px = idx >> 3, c4 = idx & 7;
            const int oy = px / TOW, ox = px - oy * TOW;
            const int c = ch * 32 + c4 * 4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < Ce) {
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 x = *(const float4 *)(E + ((oy * S + ky) * IW + ox * S + kx) * IRB_ES + c4 * 4);
                        const float4 w = *(const float4 *)(Wds + (ky * 3 + kx) * 32 + c4 * 4);
                        a4.x = fmaf(x.x, w.x, a4.x);
                        a4.y = fmaf(x.y, w.y, a4.y);
                        a4.z = fmaf(x.z, w.z, a4.z);
                        a4.w = fmaf(x.w, w.w, a4.w);
                    }
                const float4 b = *(const float4 *)(Bds + c);
                o.x = fminf(fmaxf(a4.x + b.x, 0.f), 6.f);
                o.y = fminf(fmaxf(a4.y + b.y, 0.f), 6.f);
                o.z = fminf(fmaxf(a4.z + b.z, 0.f), 6.f);
                o.w = fminf(fmaxf(a4.w + b.w, 0.f), 6.f);
            }
            *(float4 *)(D + px * IRB_ES + c4 * 4) = o;
        }
        IRB_STAMP(4);
        __syncthreads();                                    // barrier 2: D and the Wp slice complete; depthwise(ch) retired
        IRB_STAMP(5);
        if (wd_mine && more) *(float4 *)(Wds + wd_t * 32 + wd_c4 * 4) = s_wd;
        // project: acc[tile] += D . Wp^T over this wave's share of the chunk's channels
        const int k_lo = ks * kw, k_hi = min(kend, k_lo + kw);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tile = t0 + 4 * j;
            if (tile < ptiles && (j == 0 || KS == 1)) {
                const int pm = tile % MP, nt = tile / MP;
                const float *ap = D + (pm * 32 + r) * IRB_ES + 4 * hh;
                const float *bq = Wps + (nt * 32 + r) * IRB_ES + 4 * hh;
                if constexpr (MXP) {
                    const uint4 *wq = Wps3 + (nt * 32 + r) * 13 + 3 * hh;
                    for (int k = k_lo; k < k_hi; k += 16)
                        x3_mma(acc[j], x3_load(wq + 6 * (k >> 4)), x3_split(*(const float4 *)(ap + k), *(const float4 *)(ap + k + 8)));
                } else
                for (int k = k_lo; k < k_hi; k += 8) {
                    const float4 a = *(const float4 *)(ap + k);
                    const float4 b = *(const float4 *)(bq + k);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[j], 0, 0, 0);
                }
            }
        }
        IRB_STAMP(6);
        if (!EXPAND) __syncthreads();
    }
    IRB_STAMP(7);
    if (KS > 1) {
        // k-split partials -> LDS (over the dead Xs / E arrays), summed in split order; then bias (+ residual),
        // one float4 of channels per thread
        const int PS = CoutP + 4;
        float *Pp = sm_irb;                                 // [KS][NOUT][PS]
        __syncthreads();
        {
            const int pm = t0 % MP, nt = t0 / MP;
            float *pp = Pp + ((size_t)ks * NOUT + pm * 32 + r) * PS + nt * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4 *)(pp + 8 * g) = make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
        }
        __syncthreads();
        const int c4o = CoutP >> 2;
        for (int idx = tid; idx < NOUT * c4o; idx += 256) {
            const int px = idx / c4o, col = (idx - px * c4o) * 4;
            const int oy = oy0 + px / TOW, ox = ox0 + px % TOW;
            if (col >= Cout || oy >= OH || ox >= OW) continue;
            float4 v = *(const float4 *)(Pp + (size_t)px * PS + col);
            for (int q = 1; q < KS; ++q) {
                const float4 t = *(const float4 *)(Pp + ((size_t)q * NOUT + px) * PS + col);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const float4 b = *(const float4 *)(bp + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            const size_t pix = ((size_t)f * OH + oy) * OW + ox;
            if (R) {
                const float4 rv = *(const float4 *)(R + pix * Cout + col);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            *(float4 *)(Y + pix * ldy + col) = v;
        }
        return;
    }
    // epilogue: bias (+ residual) and store; lane = output pixel pm*32 + r, 16 channels in runs of four
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tile = t0 + 4 * j;
        if (tile >= ptiles) continue;
        const int pm = tile % MP, nt = tile / MP;
        const int px = pm * 32 + r;
        const int oy = oy0 + px / TOW, ox = ox0 + px % TOW;
        if (oy >= OH || ox >= OW) continue;
        const size_t pix = ((size_t)f * OH + oy) * OW + ox;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = nt * 32 + 8 * g + 4 * hh;
            if (col >= Cout) continue;
            const float4 bv = *(const float4 *)(bp + col);
            float4 v = make_float4(acc[j][4 * g] + bv.x, acc[j][4 * g + 1] + bv.y, acc[j][4 * g + 2] + bv.z, acc[j][4 * g + 3] + bv.w);
            if (R) {
                const float4 rv = *(const float4 *)(R + pix * Cout + col);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            *(float4 *)(Y + pix * ldy + col) = v;
        }
    }
}

static int launch_irb(SvcHandle *h, hipStream_t s, const float *X, int n, int H, int W, int Cin, const SvcLayer *Le,
                      const SvcLayer &Ld, const SvcLayer &Lp, int stride, const float *R, float *Y) {
    if (h->seg_off >> h->seg_cur & 1u) return SVC_OK;
    ProfScope ps(h, SVC_K_PW, s);
    const int Ce = Ld.cout, Cout = Lp.cout, CoutP = (Cout + 31) / 32 * 32;
    const int OH = H / stride, OW = W / stride;
    // split-bf16 form (h->mx; the fixed-shape expanding instances; h->irb_mx: one bit per instance, in the order below): We and Wp
    // as X3_ROWS copies
    const float *We_ = Le ? Le->w.dev : nullptr, *Wp_ = Lp.w.dev;
    auto mx_weights = [&]() -> int {
        const uint4 *a = nullptr, *b = nullptr;
        int rc = x3_weights(h, s, Le->w.dev, Cin, Cin, (Ce + 31) / 32 * 32, X3_ROWS, &a);
        We_ = (const float *)a;
        if (!rc && Cout == 64) {                             // (MXP: block 7)
            rc = x3_weights(h, s, Lp.w.dev, Ce, Ce, CoutP, X3_ROWS, &b);
            Wp_ = (const float *)b;
        }
        return rc;
    };
#define IRB_LAUNCH(S_, TOH_, TOW_, EXP_) IRB_LAUNCH4(S_, TOH_, TOW_, EXP_, 0, 0, 0, false)
#define IRB_LAUNCH3(S_, TOH_, TOW_, EXP_, CI_, CE_, CO_, BIT_)                                                       \
    do {                                                                                                             \
        if (h->mx && (h->mx_mask & 2) && (h->irb_mx >> (BIT_) & 1)) {                                                                   \
            int rc = mx_weights();                                                                                   \
            if (rc) return rc;                                                                                       \
            IRB_LAUNCH4(S_, TOH_, TOW_, EXP_, CI_, CE_, CO_, true);                                                  \
        } else IRB_LAUNCH4(S_, TOH_, TOW_, EXP_, CI_, CE_, CO_, false);                                              \
    } while (0)
#define IRB_LAUNCH4(S_, TOH_, TOW_, EXP_, CI_, CE_, CO_, MX_)                                                        \
    do {                                                                                                             \
        const int tx = ceil_div(OW, TOW_), ty = ceil_div(OH, TOH_);                                                  \
        const size_t lds = IrbGeom<S_, TOH_, TOW_>::lds_floats(Cin, CoutP, EXP_, Ce, IRB_XREG && EXP_ && (CI_) > 0 && (CI_) <= 32 && ((CI_) % 8) == 0, MX_) * 4; \
        auto kfn = k_irb<S_, TOH_, TOW_, EXP_, CI_, CE_, CO_, MX_>;                                                  \
        /* tiles of the 96-channel blocks need more than the default 64 KB of dynamic LDS; the attribute is per  */  \
        /* device, so the once-flag lives in the handle (one handle = one device), not in the process            */  \
        if (h->lds_attr_done.insert((const void *)kfn).second)                                                       \
            SVC_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024)); \
        kfn<<<dim3((unsigned)(n * tx * ty)), 256, lds, s>>>(                                                          \
            X, H, W, Cin, EXP_ ? We_ : nullptr, EXP_ ? Le->b.dev : nullptr, Ce, Ld.w.dev, Ld.b.dev, Wp_,              \
            Lp.b.dev, Cout, CoutP, R, Y, Cout, OH, OW, tx, ty);                                                       \
    } while (0)
    const bool fixed = h->irb_fixed;
    if (!Le && fixed && Cin == 32 && Ce == 32 && Cout == 16) IRB_LAUNCH4(1, 8, 8, false, 32, 32, 16, false);
    else if (!Le) IRB_LAUNCH(1, 8, 8, false);
    else if (stride == 2 && fixed && Cin == 16 && Ce == 96 && Cout == 24) IRB_LAUNCH3(2, 4, 8, true, 16, 96, 24, 0);
    else if (stride == 2 && fixed && Cin == 24 && Ce == 144 && Cout == 32) IRB_LAUNCH3(2, 4, 8, true, 24, 144, 32, 2);
    else if (stride == 2) IRB_LAUNCH(2, 4, 8, true);
    else if (fixed && Cin == 24 && Ce == 144 && Cout == 24) IRB_LAUNCH3(1, 8, 8, true, 24, 144, 24, 1);
    else if (fixed && Cin == 32 && Ce == 192 && Cout == 32) IRB_LAUNCH3(1, 8, 8, true, 32, 192, 32, 3);
    else if (fixed && Cin == 32 && Ce == 192 && Cout == 64) IRB_LAUNCH3(1, 8, 8, true, 32, 192, 64, 4);
    else IRB_LAUNCH(1, 8, 8, true);
#undef IRB_LAUNCH
#undef IRB_LAUNCH3
#undef IRB_LAUNCH4
    SVC_CHECK_LAUNCH();
    return SVC_OK;
}

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// One pass of the network over n <= plan->nb frames.
static int forward_chunk(SvcHandle *h, const uint8_t *frames, int n, uint8_t *maps, hipStream_t s, int thr = 0, unsigned *census_rows = nullptr) {
    NetPlan *p = h->plan;
    const int NH = p->NH, NW = p->NW;
    int H = NH / 2, W = NW / 2;
    size_t li = 0;
    auto next = [&]() -> const SvcLayer & { return h->layers[li++]; };
    float *IN = p->buf(B_IN), *P[2] = {p->buf(B_P0), p->buf(B_P1)}, *E0 = p->buf(B_E0), *E1 = p->buf(B_E1);
    // the front of the network (LANCZOS, features.0, features.1) as one kernel where a tile's resampling arrays fit in LDS
    const int fr_lds = front_lds_bytes();
    const bool front = h->front && h->fuse_max >= 1 && p->fr_ok;
    p->last_front = front;
    h->seg_cur = 0;
    auto seg_on = [&]() { return !(h->seg_off >> h->seg_cur & 1u); };   // SVC_SEG_OFF (measurement aid): see svc_internal.h
    // K0
    if (!front && seg_on()) {
        ProfScope ps(h, SVC_K_LANCZOS, s);
        dim3 grid(ceil_div(NH, p->lz_rows), n);
        size_t lds = ((size_t)p->lz_tile_cap * p->w * 3 + 15) / 16 * 16 + ((size_t)p->lz_tile_cap * NW * 3 + 15) / 16 * 16 +
                     (size_t)NW * p->hks * 4 + 768 * 4;
        k_lanczos_norm<<<grid, 256, lds, s>>>(frames, IN, p->h, p->w, NH, NW, (const int *)p->hb.p, (const int *)p->hk.p,
                                             p->hks, (const int *)p->vb.p, (const int *)p->vk.p, p->vks,
                                             (const float *)p->lut.p, p->lz_rows, p->lz_tile_cap);
        SVC_CHECK_LAUNCH();
    }
    const SvcLayer &Lstem = next();
    if (!front && seg_on()) {
        ProfScope ps(h, SVC_K_STEM, s);
        const int tx = ceil_div(W, STEM_TW), ty = ceil_div(H, STEM_TH);
        k_stem_mfma<<<dim3((unsigned)(n * tx * ty)), 256, 0, s>>>(IN, (const float *)h->stem_wt.p, Lstem.b.dev, P[0], NH, NW,
                                                                 H, W, tx, ty);
        SVC_CHECK_LAUNCH();
    }
    int cur = 0;
    // backbone blocks 1..17  (MobileNetV2.py:111-136)
    static const int T[7] = {1, 6, 6, 6, 6, 6, 6}, Cc[7] = {16, 24, 32, 64, 96, 160, 320}, Nn[7] = {1, 2, 3, 4, 3, 3, 1},
                     Ss[7] = {1, 2, 2, 2, 1, 2, 1};
    int idx = 1, inp = 32;
    for (int st = 0; st < 7; ++st)
        for (int i = 0; i < Nn[st]; ++i, ++idx) {
            const int oup = Cc[st], stride = (i == 0) ? Ss[st] : 1, t = T[st];
            const bool res = (stride == 1 && inp == oup);
            const bool tap = (idx == 7 || idx == 14);             // full-resolution output feeds a skip
            h->seg_cur = idx <= 1 ? 0 : idx <= 3 ? 1 : idx <= 7 ? 2 : idx <= 14 ? 3 : 4;
            const int dws = (stride == 2 && !tap) ? 2 : 1;        // stride-2 dw == stride-1 dw + ::2 sub-sampling
            const float *x = P[cur];
            int OH = H / dws, OW = W / dws;
            float *y = tap ? p->buf(idx == 7 ? B_F4X : B_F2X) : P[cur ^ 1];
            // the first fuse_max blocks (of 1..13: Cin <= 96, Cout <= 128) run as one fused kernel; the rest un-fused
            if (front && idx == 1) {
                ProfScope ps(h, SVC_K_STEM, s);
                const SvcLayer &Ld = next();
                const SvcLayer &Lp = next();
                FrontArgs A;
                A.frames = frames; A.h = p->h; A.w = p->w; A.NH = NH; A.NW = NW; A.OH = H; A.OW = W;
                A.tiles_x = ceil_div(W, FR_TW); A.tiles_y = ceil_div(H, FR_TH);
                A.hb = (const int *)p->hb.p; A.hk = (const int *)p->hk.p; A.vb = (const int *)p->vb.p; A.vk = (const int *)p->vk.p;
                A.hks = p->hks; A.vks = p->vks; A.lut = (const float *)p->lut.p;
                A.Wstem = (const float *)h->stem_wt.p; A.bstem = Lstem.b.dev;
                RC(lane_weights(h, s, (const float *)h->stem_wt.p, 32, 32, 32, &A.Wstem_l));
                RC(lane_weights(h, s, Lp.w.dev, 32, 32, 16, &A.Wp_l));
                A.Wd = Ld.w.dev; A.bd = Ld.b.dev; A.Wp = Lp.w.dev; A.bp = Lp.b.dev;
                A.Y = y; A.in_dbg = h->keep_input ? IN : nullptr;
                A.nr_cap = p->fr_nr; A.nc_cap = p->fr_nc;
                for (int i = 0; i < FR_MAXT; ++i) {
                    A.row_lo[i] = p->fr_row_lo[i]; A.row_n[i] = p->fr_row_n[i];
                    A.col_lo[i] = p->fr_col_lo[i]; A.col_n[i] = p->fr_col_n[i];
                }
                if (seg_on()) k_front<<<dim3((unsigned)(n * A.tiles_x * A.tiles_y)), 256, fr_lds, s>>>(A);
                SVC_CHECK_LAUNCH();
            } else if (idx <= h->fuse_max && (t != 1 || inp == 32)) {
                const SvcLayer *Le = (t != 1) ? &next() : nullptr;
                const SvcLayer &Ld = next();
                const SvcLayer &Lp = next();
                RC(launch_irb(h, s, x, n, H, W, inp, Le, Ld, Lp, dws, res ? x : nullptr, y));
            } else {
                const float *dwin = x;
                if (t != 1) {
                    RC(launch_pw(h, s, x, inp, next(), nullptr, 0, E0, inp * t, n * H * W, n));
                    dwin = E0;
                }
                const SvcLayer &Ld = next();
                if (h->dwpw && dws == 1 && H * W >= h->dwpw_min_px) {
                    const SvcLayer &Lp = next();
                    RC(launch_dwpw(h, s, dwin, Ld, Lp, res ? x : nullptr, oup, y, oup, n, H, W));
                } else {
                    RC(launch_dw(h, s, dwin, Ld, E1, n, H, W, dws));
                    RC(launch_pw(h, s, E1, inp * t, next(), res ? x : nullptr, oup, y, oup, n * OH * OW, n));
                }
            }
            if (tap) {
                ProfScope ps(h, SVC_K_RESAMPLE, s);
                if (seg_on()) k_subsample<<<blocks256((size_t)n * (OH / 2) * (OW / 2) * (oup / 4)), 256, 0, s>>>(y, P[cur ^ 1], n, OH,
                                                                                                  OW, oup);
                SVC_CHECK_LAUNCH();
                OH /= 2; OW /= 2;
            }
            H = OH; W = OW;
            cur ^= 1;
            inp = oup;
        }
    // features.18 -> CAT1[:, 0:1280], Gaussian maps -> CAT1[:, 1280:1296]
    const int H5 = H, W5 = W, H4 = 2 * H5, W4 = 2 * W5, H3 = 4 * H5, W3 = 4 * W5;
    float *CAT1 = p->buf(B_CAT1);
    h->seg_cur = 5;
    RC(launch_pw(h, s, P[cur], 320, next(), nullptr, 0, CAT1, 1296, n * H5 * W5, n));
    // skips (model.py:443-444)
    float *CAT2 = p->buf(B_CAT2), *CAT3 = p->buf(B_CAT3);
    {
        const SvcLayer &Le2 = next();
        const SvcLayer &Lr2 = next();
        if (pwpw_takes(h, Le2, Lr2)) RC(launch_pwpw(h, s, p->buf(B_F2X), 160, Le2, Lr2, CAT2 + 256, 384, n * H4 * W4));
        else {
            RC(launch_pw(h, s, p->buf(B_F2X), 160, Le2, nullptr, 0, p->buf(B_S2E), 320, n * H4 * W4, n));
            RC(launch_pw(h, s, p->buf(B_S2E), 320, Lr2, nullptr, 0, CAT2 + 256, 384, n * H4 * W4, n));
        }
        const SvcLayer &Le4 = next();
        const SvcLayer &Lr4 = next();
        if (pwpw_takes(h, Le4, Lr4)) RC(launch_pwpw(h, s, p->buf(B_F4X), 64, Le4, Lr4, CAT3 + 128, 192, n * H3 * W3));
        else {
            RC(launch_pw(h, s, p->buf(B_F4X), 64, Le4, nullptr, 0, p->buf(B_S4E), 128, n * H3 * W3, n));
            RC(launch_pw(h, s, p->buf(B_S4E), 128, Lr4, nullptr, 0, CAT3 + 128, 192, n * H3 * W3, n));
        }
    }
    next();   // GAUSS placeholder layer (raw parameters; maps live in plan->gauss)
    if (p->gauss_filled < n) {       // the prior maps are constants: nothing else writes channels 1280..1295 of CAT1
        ProfScope ps(h, SVC_K_RESAMPLE, s);
        k_gauss_fill<<<blocks256((size_t)n * H5 * W5 * 16), 256, 0, s>>>((const float *)p->gauss.p, CAT1, n, H5 * W5,
                                                                         1296, 1280);
        SVC_CHECK_LAUNCH();
        p->gauss_filled = n;
    }
    // post_cnn
    if (h->dwpw && H5 * W5 >= h->dwpw_min_px) {
        const SvcLayer &Ld = next();
        RC(launch_dwpw(h, s, CAT1, Ld, next(), nullptr, 0, p->buf(B_PC), 256, n, H5, W5));
    } else {
        RC(launch_dw(h, s, CAT1, next(), p->buf(B_PCD), n, H5, W5, 1));
        RC(launch_pw(h, s, p->buf(B_PCD), 1296, next(), nullptr, 0, p->buf(B_PC), 256, n * H5 * W5, n));
    }
    // US1 + concat, US2 block
    h->seg_cur = 6;
    if (h->split_up) {
        // expand(concat(up(PC), skip)) = relu6(up(W[:, :256] . PC) + W[:, 256:] . skip + b)
        const SvcLayer &Le = next();
        RC(launch_pw_ex(h, s, p->buf(B_PC), 256, Le.w.dev, 384, 256, nullptr, 0, 768, nullptr, 0, p->buf(B_T1), 768,
                        n * H5 * W5, n, nullptr));
        const UpsAdd ua{p->buf(B_T1), H5, W5, 768, make_fdiv(W4), make_fdiv(H4)};
        RC(launch_pw_ex(h, s, CAT2 + 256, 384, Le.w.dev + 256, 384, 128, Le.b.dev, Le.relu6, 768, nullptr, 0,
                        p->buf(B_U2E), 768, n * H4 * W4, n, &ua));
    } else {
        {
            ProfScope ps(h, SVC_K_RESAMPLE, s);
            if (seg_on()) k_upsample2x<<<blocks256((size_t)n * H4 * W4 * 64), 256, 0, s>>>(p->buf(B_PC), CAT2, n, H5, W5, 256, 384,
                                                                             make_fdiv(64), make_fdiv(W4), make_fdiv(H4));
            SVC_CHECK_LAUNCH();
        }
        RC(launch_pw(h, s, CAT2, 384, next(), nullptr, 0, p->buf(B_U2E), 768, n * H4 * W4, n));
    }
    if (h->dwpw) {
        const SvcLayer &Ld = next();
        RC(launch_dwpw(h, s, p->buf(B_U2E), Ld, next(), nullptr, 0, p->buf(B_U2), 128, n, H4, W4));
    } else {
        RC(launch_dw(h, s, p->buf(B_U2E), next(), p->buf(B_U2D), n, H4, W4, 1));
        RC(launch_pw(h, s, p->buf(B_U2D), 768, next(), nullptr, 0, p->buf(B_U2), 128, n * H4 * W4, n));
    }
    h->seg_cur = 7;
    if (h->split_up) {
        const SvcLayer &Le = next();
        RC(launch_pw_ex(h, s, p->buf(B_U2), 128, Le.w.dev, 192, 128, nullptr, 0, 384, nullptr, 0, p->buf(B_T2), 384,
                        n * H4 * W4, n, nullptr));
        const UpsAdd ua{p->buf(B_T2), H4, W4, 384, make_fdiv(W3), make_fdiv(H3)};
        RC(launch_pw_ex(h, s, CAT3 + 128, 192, Le.w.dev + 128, 192, 64, Le.b.dev, Le.relu6, 384, nullptr, 0,
                        p->buf(B_P3E), 384, n * H3 * W3, n, &ua));
    } else {
        {
            ProfScope ps(h, SVC_K_RESAMPLE, s);
            if (seg_on()) k_upsample2x<<<blocks256((size_t)n * H3 * W3 * 32), 256, 0, s>>>(p->buf(B_U2), CAT3, n, H4, W4, 128, 192,
                                                                             make_fdiv(32), make_fdiv(W3), make_fdiv(H3));
            SVC_CHECK_LAUNCH();
        }
        RC(launch_pw(h, s, CAT3, 192, next(), nullptr, 0, p->buf(B_P3E), 384, n * H3 * W3, n));
    }
    if (h->dwpw) {
        const SvcLayer &Ld = next();
        RC(launch_dwpw(h, s, p->buf(B_P3E), Ld, next(), nullptr, 0, p->buf(B_DEC), 64, n, H3, W3));
    } else {
        RC(launch_dw(h, s, p->buf(B_P3E), next(), p->buf(B_P3D), n, H3, W3, 1));
        RC(launch_pw(h, s, p->buf(B_P3D), 384, next(), nullptr, 0, p->buf(B_DEC), 64, n * H3 * W3, n));
    }
    // adaptation, smoothing, resize, quantise
    h->seg_cur = 8;
    const SvcLayer &La = next();
    {
        ProfScope ps(h, SVC_K_RESAMPLE, s);
        if (seg_on()) k_adapt<<<blocks256((size_t)n * H3 * W3), 256, 0, s>>>(p->buf(B_DEC), La.w.dev, La.b.dev, p->buf(B_LOGIT),
                                                              (size_t)n * H3 * W3, (unsigned *)p->fmax.p, n,
                                                              (unsigned long long *)h->census.p, h->chunk);
        SVC_CHECK_LAUNCH();
    }
    // B_LOGIT per-frame stride may exceed H3*W3 (rounded to 4): compact layout is used instead
    const SvcLayer &Ls = next();
    ProfScope ps_smooth(h, SVC_K_SMOOTH, s);
    {
        dim3 grid(ceil_div(p->h, p->sd_rows), n);
        if (h->smooth_mfma && NW % 8 == 0 && NW == 8 * W3 && NH == 8 * H3) {
            const size_t lds = ((size_t)H3 * W3 + 64 * SD_KP + (size_t)p->sd_tile_cap * NW) * sizeof(float);
            if (seg_on()) k_smooth_down_mfma<<<grid, 256, lds, s>>>(p->buf(B_LOGIT), Ls.w.dev, p->buf(B_PRE), (unsigned *)p->fmax.p, H3, W3,
                                                      NH, NW, p->h, p->w, p->sd_rows, p->sd_tile_cap, make_fdiv(p->w));
        } else {
            size_t lds = ((size_t)H3 * W3 + 64 * 49 + (size_t)p->sd_tile_cap * NW) * sizeof(float);
            if (seg_on()) k_smooth_down<<<grid, 256, lds, s>>>(p->buf(B_LOGIT), Ls.w.dev, p->buf(B_PRE), (unsigned *)p->fmax.p, H3, W3, NH,
                                                 NW, p->h, p->w, p->sd_rows, p->sd_tile_cap, make_fdiv(NW), make_fdiv(p->w));
        }
        SVC_CHECK_LAUNCH();
    }
    if (seg_on()) k_quantise<<<blocks256((size_t)n * p->h * p->w), 256, 0, s>>>(p->buf(B_PRE), (const unsigned *)p->fmax.p, maps, n,
                                                                 p->h * p->w, make_fdiv(p->h * p->w), thr,
                                                                 (unsigned long long *)h->census.p, census_rows);
    SVC_CHECK_LAUNCH();
    if (thr > 0) h->census_maps += (unsigned long long)n;
    p->last_n = n;
    return SVC_OK;
}

static int saliency_impl(SvcHandle *h, const uint8_t *frames, int n, int height, int width, uint8_t *maps, int thr, void *stream,
                         uint32_t *census_rows = nullptr);

extern "C" int svc_saliency_u8(SvcHandle *h, const uint8_t *frames, int n, int height, int width, uint8_t *maps,
                               void *stream) {
    return saliency_impl(h, frames, n, height, width, maps, 0, stream);
}

extern "C" int svc_saliency_thresholded_u8(SvcHandle *h, const uint8_t *frames, int n, int height, int width, uint8_t *maps,
                                           int t, void *stream) {
    if (t < 0 || t > 255) { svc_set_error("svc_saliency_thresholded_u8: threshold outside 0..255"); return SVC_E_INVALID; }
    return saliency_impl(h, frames, n, height, width, maps, t, stream);
}

// svc_saliency_thresholded_u8 that also ADDS, per frame, the pixels of the un-thresholded map at t - 1, t, t + 1 to the caller's
// device rows census_n4[n][4] (u32; column 3 unused): the per-video form of svc_threshold_census (the multi-video job's lanes
// mix the frames of many videos in one pass).  The caller zeroes the rows.
extern "C" int svc_saliency_census_u8(SvcHandle *h, const uint8_t *frames, int n, int height, int width, uint8_t *maps,
                                      int t, uint32_t *census_n4, void *stream) {
    if (t < 1 || t > 255) { svc_set_error("svc_saliency_census_u8: threshold outside 1..255"); return SVC_E_INVALID; }
    return saliency_impl(h, frames, n, height, width, maps, t, stream, census_n4);
}

static int saliency_impl(SvcHandle *h, const uint8_t *frames, int n, int height, int width, uint8_t *maps, int thr, void *stream,
                         uint32_t *census_rows) {
    if (!h || n < 0 || (n > 0 && (!frames || !maps)) || height < 8 || width < 8) {     // n = 0: a no-op, null buffers allowed
        svc_set_error("svc_saliency_u8: invalid argument");
        return SVC_E_INVALID;
    }
    if (n == 0) return SVC_OK;
    SVC_HIP(hipSetDevice(h->device));
    const int nb = std::min(n, h->chunk);
    RC(build_plan(h, height, width, nb));
    const size_t fin = (size_t)height * width * 3, fout = (size_t)height * width;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; i += h->plan->nb) {
        int m = std::min(h->plan->nb, n - i);
        const uint8_t *fr = frames + i * fin;
        uint8_t *mp = maps + i * fout;
        RC(forward_chunk(h, fr, m, mp, s, thr, census_rows ? census_rows + (size_t)i * 4 : nullptr));
    }
    return SVC_OK;
}

extern "C" int svc_debug_tap(SvcHandle *h, int which, int frame, float *out_host, size_t cap_floats) {
    if (!h || !h->plan || !out_host) { svc_set_error("svc_debug_tap: no forward pass has run"); return SVC_E_INVALID; }
    NetPlan *p = h->plan;
    static const int map[7] = {B_IN, B_F4X, B_F2X, B_CAT1, B_PC, B_DEC, B_PRE};
    if (which < 0 || which > 6 || frame < 0 || frame >= p->last_n) { svc_set_error("svc_debug_tap: bad tap/frame"); return SVC_E_INVALID; }
    const int NH = p->NH, NW = p->NW;
    size_t count;
    switch (which) {
        case SVC_TAP_INPUT: count = (size_t)NH * NW * 3; break;
        case SVC_TAP_FEAT4X: count = (size_t)(NH / 8) * (NW / 8) * 64; break;
        case SVC_TAP_FEAT2X: count = (size_t)(NH / 16) * (NW / 16) * 160; break;
        case SVC_TAP_FEAT1X: count = (size_t)(NH / 32) * (NW / 32) * 1296; break;
        case SVC_TAP_POSTCNN: count = (size_t)(NH / 32) * (NW / 32) * 256; break;
        case SVC_TAP_DEC: count = (size_t)(NH / 8) * (NW / 8) * 64; break;
        default: count = (size_t)p->h * p->w; break;
    }
    if (count > cap_floats) { svc_set_error("svc_debug_tap: buffer too small (%zu needed)", count); return SVC_E_INVALID; }
    if (which == SVC_TAP_INPUT && p->last_front && !h->keep_input) {
        svc_set_error("svc_debug_tap: the fused front kernel keeps the network input on chip; create the handle with SVC_KEEP_INPUT=1 (or SVC_FRONT=0)");
        return SVC_E_INVALID;
    }
    SVC_HIP(hipSetDevice(h->device));
    SVC_HIP(hipDeviceSynchronize());
    SVC_HIP(hipMemcpy(out_host, p->buf(map[which]) + (size_t)frame * count, count * sizeof(float), hipMemcpyDeviceToHost));
    return (int)count;
}

extern "C" int svc_front_fused(const SvcHandle *h) { return h && h->plan && h->plan->last_front ? 1 : 0; }
extern "C" int svc_matrix_pipe(const SvcHandle *h) { return h ? h->mx : 0; }

// out[0] = maps that went through svc_saliency_thresholded_u8 since the last reset, out[1..3] = pixels of their un-thresholded
// maps at t - 1, t, t + 1 (t = the threshold of the call).  Synchronises the device.
extern "C" int svc_threshold_census(SvcHandle *h, unsigned long long *out, int reset) {
    if (!h || !out) { svc_set_error("svc_threshold_census: invalid argument"); return SVC_E_INVALID; }
    SVC_HIP(hipSetDevice(h->device));
    SVC_HIP(hipDeviceSynchronize());
    std::vector<unsigned char> host(h->census.bytes);
    SVC_HIP(hipMemcpy(host.data(), h->census.p, h->census.bytes, hipMemcpyDeviceToHost));
    const unsigned long long *tot = (const unsigned long long *)host.data();
    const unsigned *pf = (const unsigned *)(tot + 4);
    out[0] = h->census_maps;
    for (int j = 0; j < 3; ++j) {
        unsigned long long v = tot[j];
        for (int f = 0; f < h->chunk; ++f) v += pf[f * 4 + j];
        out[1 + j] = v;
    }
    if (reset) {
        SVC_HIP(hipMemset(h->census.p, 0, h->census.bytes));
        h->census_maps = 0;
    }
    return SVC_OK;
}

// --------------------------------------------------------------------------------------
// create / destroy
// --------------------------------------------------------------------------------------
static const uint64_t BLOB_MAGIC = 0x53564331ull;

static int take(SvcHandle *h, size_t &ti, size_t expect, SvcTensor &out, const char *what) {
    if (ti >= h->tensors.size()) { svc_set_error("blob: ran out of tensors at %s", what); return SVC_E_BLOB; }
    if (h->tensors[ti].n != expect) {
        svc_set_error("blob: tensor %zu (%s) has %zu floats, expected %zu", ti, what, h->tensors[ti].n, expect);
        return SVC_E_BLOB;
    }
    out = h->tensors[ti++];
    return SVC_OK;
}

static int add_pw(SvcHandle *h, size_t &ti, int cin, int cout, int relu6, const char *what) {
    SvcLayer L{SvcLayer::PW, cin, cout, 1, relu6, {}, {}};
    const size_t npad = (cout + 31) / 32 * 32;
    RC(take(h, ti, npad * cin, L.w, what));
    RC(take(h, ti, (size_t)cout, L.b, what));
    h->layers.push_back(L);
    return SVC_OK;
}
static int add_dw(SvcHandle *h, size_t &ti, int c, int stride, const char *what) {
    SvcLayer L{SvcLayer::DW, c, c, stride, 1, {}, {}};
    RC(take(h, ti, (size_t)9 * c, L.w, what));
    RC(take(h, ti, (size_t)c, L.b, what));
    h->layers.push_back(L);
    return SVC_OK;
}
static int add_inv_res(SvcHandle *h, size_t &ti, int inp, int oup, int t, int stride, const char *what) {
    if (t != 1) RC(add_pw(h, ti, inp, inp * t, 1, what));
    RC(add_dw(h, ti, inp * t, stride, what));
    RC(add_pw(h, ti, inp * t, oup, 0, what));
    return SVC_OK;
}

extern "C" int svc_create(const void *blob_host, size_t n_bytes, int device, SvcHandle **out) {
    if (!blob_host || !out || n_bytes < 16) { svc_set_error("svc_create: invalid argument"); return SVC_E_INVALID; }
    const uint64_t *hd = (const uint64_t *)blob_host;
    if (hd[0] != BLOB_MAGIC) { svc_set_error("svc_create: bad blob magic"); return SVC_E_BLOB; }
    const size_t nt = hd[1];
    if (16 + 16 * nt > n_bytes) { svc_set_error("svc_create: truncated blob header"); return SVC_E_BLOB; }
    int ndev = 0;
    SVC_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { svc_set_error("svc_create: device %d of %d", device, ndev); return SVC_E_INVALID; }
    SVC_HIP(hipSetDevice(device));
    SvcHandle *h = new SvcHandle();
    h->device = device;
    const char *env = getenv("SVC_CHUNK");
    if (env && atoi(env) > 0) h->chunk = atoi(env);
    env = getenv("SVC_PW_MIN_WG");
    if (env && atoi(env) > 0) h->pw_min_wg = atoi(env);
    env = getenv("SVC_PW_SK");
    if (env) h->pw_sk = atoi(env) != 0;
    env = getenv("SVC_PW_TR");
    if (env) h->pw_tr = atoi(env);
    env = getenv("SVC_PW_SK_MAX");
    if (env && atoi(env) > 0) h->pw_sk_max = atoi(env);
    env = getenv("SVC_PW_SMALL");
    if (env) h->pw_small = atoi(env);
    env = getenv("SVC_PW16");
    if (env) h->pw16 = atoi(env) != 0;
    env = getenv("SVC_PWR");
    if (env) h->pwr = atoi(env) != 0;
    env = getenv("SVC_PWR_NT");
    if (env) h->pwr_nt = atoi(env);
    env = getenv("SVC_PWR_MIN_WG");
    if (env && atoi(env) > 0) h->pwr_min_wg = atoi(env);
    env = getenv("SVC_FUSE_MAX");
    if (env) h->fuse_max = std::min(13, std::max(0, atoi(env)));
    env = getenv("SVC_SPLIT_UP");
    if (env) h->split_up = atoi(env) != 0;
    env = getenv("SVC_IRB_FIXED");
    if (env) h->irb_fixed = atoi(env) != 0;
    env = getenv("SVC_SMOOTH_MFMA");
    if (env) h->smooth_mfma = atoi(env);
    env = getenv("SVC_FRONT");
    if (env) h->front = atoi(env) != 0;
    env = getenv("SVC_KEEP_INPUT");
    if (env) h->keep_input = atoi(env) != 0;
    env = getenv("SVC_DWPW");
    if (env) h->dwpw = atoi(env) != 0;
    env = getenv("SVC_DWPW_MIN_PX");
    if (env) h->dwpw_min_px = atoi(env);
    env = getenv("SVC_MX");
    if (env) {                                               // a typo must not silently select another pipe
        if (!strcmp(env, "bf16x6") || !strcmp(env, "6")) h->mx = 6;
        else if (!strcmp(env, "f32") || !strcmp(env, "0")) h->mx = 0;
        else { svc_set_error("svc_create: SVC_MX=%s (expected f32 or bf16x6)", env); delete h; return SVC_E_INVALID; }
    }
    if (h->mx && !getenv("SVC_DWPW_MIN_PX")) h->dwpw_min_px = 100;
    env = getenv("SVC_IRB_MX");
    if (env) h->irb_mx = (unsigned)strtoul(env, nullptr, 0);
    env = getenv("SVC_MX_MASK");
    if (env) h->mx_mask = (unsigned)strtoul(env, nullptr, 0);
    env = getenv("SVC_SK_LANE");
    if (env) h->sk_lane = atoi(env) != 0;
    env = getenv("SVC_PWPW");
    if (env) h->pwpw = atoi(env) != 0;
    env = getenv("SVC_SEG_OFF");
    if (env) h->seg_off = (unsigned)strtoul(env, nullptr, 0);
    env = getenv("SVC_DWPW_NT");
    if (env) h->dwpw_max_nt = atoi(env);
    env = getenv("SVC_DW_TILE");
    if (env) h->dw_tile = atoi(env);
    env = getenv("SVC_SHOT_MX");
    if (env) {
        if (!strcmp(env, "bf16x6") || !strcmp(env, "6")) h->shot_mx = 6;
        else if (!strcmp(env, "bf16x3") || !strcmp(env, "3")) h->shot_mx = 3;
        else if (!strcmp(env, "f32") || !strcmp(env, "0")) h->shot_mx = 0;
        else { svc_set_error("svc_create: SVC_SHOT_MX=%s (expected f32, bf16x6 or bf16x3)", env); delete h; return SVC_E_INVALID; }
    }
    env = getenv("SVC_SHOT_M16");
    if (env) { const int v = atoi(env); if (v >= 2 && v <= 4) h->shot_m16 = v; }
    env = getenv("SVC_SHOT_XCD");
    if (env) h->shot_xcd = atoi(env) != 0;
    env = getenv("SVC_PRIM_PT");
    if (env && atoi(env) > 0) h->prim_pt = atoi(env);
    env = getenv("SVC_TAIL_MERGE");
    if (env) h->tail_merge = atoi(env);
    env = getenv("SVC_TREE_PAR");
    if (env) h->tree_par = atoi(env);
    env = getenv("SVC_PRIM_LVL");
    if (env) h->prim_lvl = atoi(env);
    int rc = h->blob.ensure(n_bytes);
    if (rc) { delete h; return rc; }
    if ((rc = h->census.ensure(32 + (size_t)h->chunk * 16)) || hipMemset(h->census.p, 0, 32 + (size_t)h->chunk * 16) != hipSuccess) {
        if (!rc) { svc_set_error("svc_create: census buffer"); rc = SVC_E_HIP; }
        h->blob.release(); h->census.release(); delete h;
        return rc;
    }
    if (hipMemcpy(h->blob.p, blob_host, n_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        svc_set_error("svc_create: blob upload failed");
        h->blob.release(); delete h;
        return SVC_E_HIP;
    }
    for (size_t i = 0; i < nt; ++i) {
        uint64_t off = hd[2 + 2 * i], cnt = hd[3 + 2 * i];
        if ((off + cnt) * 4 > n_bytes) { svc_set_error("svc_create: tensor %zu out of range", i); svc_destroy(h); return SVC_E_BLOB; }
        h->tensors.push_back(SvcTensor{(const float *)h->blob.p + off, (size_t)cnt});
    }
    // walk the fixed graph (same order as retargetvid_amd.weights.fold_state_dict)
    size_t ti = 0;
    auto fail = [&](int code) { svc_destroy(h); return code; };
    {
        SvcLayer L{SvcLayer::STEM, 3, 32, 2, 1, {}, {}};
        if ((rc = take(h, ti, 27 * 32, L.w, "stem.w")) || (rc = take(h, ti, 32, L.b, "stem.b"))) return fail(rc);
        h->layers.push_back(L);
        // [27 taps][32 out] -> [32 out][32 taps] for k_stem_mfma
        const float *w_host = (const float *)blob_host + (L.w.dev - (const float *)h->blob.p);
        std::vector<float> wt(32 * 32, 0.f);
        for (int k = 0; k < 27; ++k)
            for (int co = 0; co < 32; ++co) wt[co * 32 + k] = w_host[k * 32 + co];
        if ((rc = h->stem_wt.ensure(wt.size() * 4))) return fail(rc);
        if (hipMemcpy(h->stem_wt.p, wt.data(), wt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            svc_set_error("svc_create: stem weight upload failed");
            return fail(SVC_E_HIP);
        }
    }
    static const int T[7] = {1, 6, 6, 6, 6, 6, 6}, Cc[7] = {16, 24, 32, 64, 96, 160, 320}, Nn[7] = {1, 2, 3, 4, 3, 3, 1},
                     Ss[7] = {1, 2, 2, 2, 1, 2, 1};
    int inp = 32;
    for (int st = 0; st < 7; ++st)
        for (int i = 0; i < Nn[st]; ++i) {
            if ((rc = add_inv_res(h, ti, inp, Cc[st], T[st], i == 0 ? Ss[st] : 1, "backbone"))) return fail(rc);
            inp = Cc[st];
        }
    if ((rc = add_pw(h, ti, 320, 1280, 1, "features.18"))) return fail(rc);
    if ((rc = add_pw(h, ti, 160, 320, 1, "skip_2x.expansion")) || (rc = add_pw(h, ti, 320, 128, 0, "skip_2x.reduction")) ||
        (rc = add_pw(h, ti, 64, 128, 1, "skip_4x.expansion")) || (rc = add_pw(h, ti, 128, 64, 0, "skip_4x.reduction")))
        return fail(rc);
    {
        SvcLayer L{SvcLayer::GAUSS, 16, 16, 1, 0, {}, {}};
        if ((rc = take(h, ti, 64, L.w, "gaussians"))) return fail(rc);
        h->layers.push_back(L);
        h->gauss_params.resize(64);
        const uint8_t *base = (const uint8_t *)blob_host;
        memcpy(h->gauss_params.data(), base + ((const uint8_t *)L.w.dev - (const uint8_t *)h->blob.p), 64 * 4);
    }
    if ((rc = add_inv_res(h, ti, 1296, 256, 1, 1, "post_cnn")) || (rc = add_inv_res(h, ti, 384, 128, 2, 1, "upsampling_2")) ||
        (rc = add_inv_res(h, ti, 192, 64, 2, 1, "post_upsampling_2")))
        return fail(rc);
    {
        SvcLayer L{SvcLayer::ADAPT, 64, 1, 1, 0, {}, {}};
        if ((rc = take(h, ti, 64, L.w, "adaptation.w")) || (rc = take(h, ti, 1, L.b, "adaptation.b"))) return fail(rc);
        h->layers.push_back(L);
        SvcLayer S{SvcLayer::SMOOTH, 1, 1, 1, 0, {}, {}};
        if ((rc = take(h, ti, 64 * 49, S.w, "smoothing phase table"))) return fail(rc);
        h->layers.push_back(S);
    }
    if (ti != h->tensors.size()) { svc_set_error("svc_create: %zu unused tensors in blob", h->tensors.size() - ti); return fail(SVC_E_BLOB); }
    *out = h;
    return SVC_OK;
}

extern "C" int svc_profile_enable(SvcHandle *h, int kernel_class) {
    if (!h || kernel_class < -1 || kernel_class >= SVC_K_COUNT) { svc_set_error("svc_profile_enable: invalid argument"); return SVC_E_INVALID; }
    h->prof_class = kernel_class;
    return SVC_OK;
}

// raw_total_ms: the event-pair durations as measured; pair_ms: what an EMPTY event pair costs on a stream of this
// device (median of 15 on a stream the handle owns -- the caller's stream may be gone by now); launches.
extern "C" int svc_profile_read_raw(SvcHandle *h, double *raw_total_ms, double *pair_ms, int *launches) {
    if (!h || !raw_total_ms || !pair_ms || !launches) { svc_set_error("svc_profile_read_raw: invalid argument"); return SVC_E_INVALID; }
    SVC_HIP(hipSetDevice(h->device));
    SVC_HIP(hipDeviceSynchronize());
    *pair_ms = 0.0;
    if (!h->prof_events.empty()) {
        if (!h->prof_cal_stream) SVC_HIP(hipStreamCreateWithFlags(&h->prof_cal_stream, hipStreamNonBlocking));
        std::vector<float> emp;
        for (int i = 0; i < 15; ++i) {
            hipEvent_t a, b;
            SVC_HIP(hipEventCreate(&a)); SVC_HIP(hipEventCreate(&b));
            SVC_HIP(hipEventRecord(a, h->prof_cal_stream)); SVC_HIP(hipEventRecord(b, h->prof_cal_stream));
            SVC_HIP(hipEventSynchronize(b));
            float ms = 0.f;
            SVC_HIP(hipEventElapsedTime(&ms, a, b));
            emp.push_back(ms);
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        }
        std::sort(emp.begin(), emp.end());
        *pair_ms = emp[emp.size() / 2];
    }
    double tot = 0.0;
    for (auto &e : h->prof_events) {
        float ms = 0.f;
        SVC_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        tot += (double)ms;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    *raw_total_ms = tot;
    *launches = (int)h->prof_events.size();
    h->prof_events.clear();
    return SVC_OK;
}

// The corrected sum: an event pair around a launch also times the event packets themselves; 3/4 of an empty pair's
// cost (the share that is not hidden behind the kernel: calibrated against rocprofv3's kernel durations of the same
// run -- 39 launches: raw events 1.696 ms, minus a whole pair each 1.519, rocprofv3 1.564, minus 3/4 of a pair 1.563)
// is taken off every launch.  svc_profile_read_raw returns the two ingredients separately.
extern "C" int svc_profile_read(SvcHandle *h, double *total_ms, int *launches) {
    if (!h || !total_ms || !launches) { svc_set_error("svc_profile_read: invalid argument"); return SVC_E_INVALID; }
    double raw = 0.0, pair = 0.0;
    const int rc = svc_profile_read_raw(h, &raw, &pair, launches);
    if (rc) return rc;
    *total_ms = std::max(0.0, raw - 0.75 * pair * *launches);
    return SVC_OK;
}

extern "C" int svc_destroy(SvcHandle *h) {
    if (!h) return SVC_OK;
    (void)hipSetDevice(h->device);
    for (auto &e : h->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    h->prof_events.clear();
    if (h->prof_cal_stream) { (void)hipStreamDestroy(h->prof_cal_stream); h->prof_cal_stream = nullptr; }
    svc_net_release(h);
    h->tail_ws.release();
    h->tail_offsets.release();
    h->tail_ring_cnt.release();
    for (auto &kv : h->lane_w) kv.second.release();
    for (auto &kv : h->x3_w) kv.second.release();
    h->census.release();
    for (auto &kv : h->tail_delta) kv.second.release();
    h->stem_wt.release();
    h->shot_blob.release();
    h->shot_ws.release();
    h->shot_w3.release();
    h->rs_maps.release();
    for (auto &kv : h->rs_tabs) { kv.second.first.release(); kv.second.second.release(); }
    for (auto &e : h->depth_ev) if (e) (void)hipEventDestroy(e);
    if (h->depth_pinned) (void)hipHostFree(h->depth_pinned);
    h->blob.release();
    delete h;
    return SVC_OK;
}
