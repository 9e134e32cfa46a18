// svc_shot.hip — TransNet V1 (shot-boundary network) forward for gfx950.
//
// Reference: 3rd_party_libs/transnetv1/transnetv1_handler.py:25-97 (TensorFlow 1.x graph): uint8 frames [B, T, 27, 48, 3]
// / 255 -> three blocks of two DDCNN cells (a cell = four Conv3D 3x3x3 with temporal dilation 1, 2, 4, 8 on the same
// input, bias, ReLU, concatenated) with a (1, 2, 2) max-pool after each block -> flatten -> Dense 256 ReLU -> Dense 2 ->
// softmax[:, :, 1].  The CPU restatement the tests compare with is oracle/transnet_ref.py.
//
// Layout: NDHWC fp32 (a frame position's channels contiguous), so a 3x3x3 convolution is 27 shifted 1x1 convolutions
// accumulated into one MFMA tile: k_shot_conv is an implicit GEMM [positions x 27 C] . [27 C x filters] on
// v_mfma_f32_32x32x2_f32 (exact fp32 products) with the operands the way round the network kernels use them (weights as
// A, positions as B: a lane ends with 16 channels of ONE position in runs of four -> bias, ReLU, float4 stores).  The four
// dilations of a cell are the y dimension of one launch and write disjoint channel ranges of the cell's output.
// Dense(256) is the same kernel with one tap over the 4608 flattened inputs, K cut in eight parts (k_shot_head adds them).
//
// Default since round 5 (SVC_SHOT_MX = SVC_MX = bf16x6): the cells run on v_mfma_f32_32x32x16_bf16 with every f32 operand as
// three bf16 planes (svc_x3.h) and the activations kept split and planar between the cells -- "split-bf16 activations" below:
// k_shot_in_x3 -> k_shot_first_x3 -> k_shot_conv_x3m ... k_shot_pool_x3 -> Dense -> k_shot_head.  The fp32 kernels
// (k_shot_conv for the first cell and Dense, k_shot_conv_lds2* for the cells) are SVC_SHOT_MX=f32.
// Round 6 removed the measured losers: the fp32 forms with operands straight from global memory / weights only through LDS
// (SVC_SHOT_FORM 0 / 1: 19.5 / 17.0 ms per 8 windows against 10.9) and the split-bf16 cells on the 32x32x16 shape (k_shot_conv_x3,
// SVC_SHOT_M16=0: 89.4 k against 94.1 k video frames/s); DESIGN_HISTORY.md keeps their measurements.
#include <algorithm>

#include "svc_internal.h"
#include "svc_x3.h"


#define SHOT_H 27
#define SHOT_W 48
#define SHOT_F 16
#define SHOT_L 3
#define SHOT_S 2
#define SHOT_D 256

// uint8 [n][27][48][3] -> float [n][27][48][4] = v / 255 (tf.cast(float32) / 255.), fourth channel 0
__global__ __launch_bounds__(256) void k_shot_in(const uint8_t *__restrict__ in, float *__restrict__ out, size_t npix) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const uint8_t *p = in + i * 3;
    *(float4 *)(out + i * 4) = make_float4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, 0.f);
}

struct ShotConv {
    const float *X;         // [B][T][H][W][C]
    const float *Wt;        // [branches][Fpad][kpad]: rows = output channel, k = tap * C + channel
    const float *bias;      // [branches * F]
    float *Y;               // [B][T][H][W][ldy], a branch writes channels br * F ..
    long long M;            // positions = B * T * H * W
    int T, H, W, C, logC;   // C = 1 << logC for the 27-tap form
    int F, Fpad, kpad, ntaps, ldy, relu;
    int ksplit;             // Dense only: > 1 = grid z cuts K in ksplit parts, part z writes its raw sums to Y + z * M * ldy (k_shot_head adds them, in order)
    int rT, rt0, rtn;       // Dense only: rows (frames) rt0 .. rt0 + rtn - 1 of every window of rT rows are computed (svc_transnet_predict_rows); rtn == rT: all
};

// ---- temporal crop (svc_transnet_predict_rows) --------------------------------------------------------------------------
// The caller keeps rows [r0, r1) of every window (the reference keeps the middle 50 of 100: transnetv1_handler.py:117-121).  A
// cell reaches 8 frames to either side (its largest dilation), so the last cell is needed on [r0, r1) only, the one before on
// [r0 - 8, r1 + 8), ... -- every layer computes the frames t0 .. t0 + tn - 1 of every window and nothing else (T = 100, rows
// 25 .. 74: cells 6 / 5 / 4 / 3 compute 50 / 66 / 82 / 98 frames: a fifth of the network's FLOPs less, the kept rows bit for bit
// those of the full pass: a kernel's tiles walk a COMPACT index space (window, frame - t0, y, x) and shot_real maps an index to
// the position in the full-length planes; what the kw-shift needs of neighbouring lanes -- consecutive positions inside a
// frame row -- holds in both).
__device__ __forceinline__ long long shot_real(long long c, int T, int t0, int tn, int HW) {
    if (tn == T) return c;
    const long long span = (long long)tn * HW, w = c / span;
    return c + (w * (T - tn) + t0) * HW;
}

// ---- split-bf16 activations (SVC_MX=bf16x6: the cells with >= 64 input channels on the bf16 matrix pipe) ---------------
// Between the cells an activation is kept SPLIT (svc_x3.h: x = hi + mid + lo, three bf16, exact) and PLANAR:
//   X3 [C / 16 groups q][3 planes][Mp positions][2 halves hh] uint4,   a uint4 = the eight channels
//   16 q + 8 (j >> 2) + 4 hh + (j & 3), j = 0..7, of one position = the fragment lane (r, hh) feeds to
//   v_mfma_f32_32x32x16_bf16 for the 16-deep step q (svc_x3.h's operand convention).
// So the 32 positions of an MFMA tile are ONE contiguous kilobyte per (q, plane), a tap of the 3x3x3 window is that kilobyte
// shifted by a constant, and a wavefront loads its position operand straight into the MFMA registers with whole-line loads:
// no LDS round trip, no barrier, no split arithmetic in the loop (a value is split once, by the kernel that produces it, and
// read 27 taps x 4 dilations x filter groups times).  A plane starts with four zero positions (128 bytes): a lane whose tap
// falls outside the frame / the window (SAME padding) is pointed at position 0 instead of being zeroed in registers.
// 6 bytes per value instead of 4; positions in the plane sit at index 4 + m.
#define SHOT_PAD 4
__device__ __forceinline__ void shot_store_x3(uint4 *Y3, long long Mp, long long m, int hh, int q, const float4 v0, const float4 v1) {
    const X3 s = x3_split(v0, v1);
    X3Q H, Mi, L;
    H.v = s.h; Mi.v = s.m; L.v = s.l;
    uint4 *o = Y3 + ((size_t)(q * 3) * Mp + SHOT_PAD + m) * 2 + hh;
    o[0] = H.q;
    o[(size_t)Mp * 2] = Mi.q;
    o[(size_t)Mp * 4] = L.q;
}
// the zero positions in front of the planes q0 .. q0 + nq - 1 (called by the first workgroup of the producing launch)
__device__ __forceinline__ void shot_zero_pads(uint4 *Y3, long long Mp, int q0, int nq, int tid) {
    for (int i = tid; i < nq * 3 * SHOT_PAD * 2; i += 256) Y3[(size_t)(q0 * 3 + i / (SHOT_PAD * 2)) * Mp * 2 + i % (SHOT_PAD * 2)] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_shot_conv(const ShotConv A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    long long m = std::min((long long)blockIdx.x * 128 + wave * 32 + r, A.M - 1);
    if (A.ntaps == 1) m = shot_real(m, A.rT, A.rt0, A.rtn, 1);       // Dense: A.M counts the rows that are computed (svc_transnet_predict_rows)
    const int tiles = A.Fpad >> 5, br = blockIdx.y / tiles, nt = blockIdx.y - br * tiles;
    const int d = 1 << br;                                   // temporal dilation of this branch: 1, 2, 4, 8
    // position of this lane
    const int x = (int)(m % A.W);
    long long q = m / A.W;
    const int y = (int)(q % A.H);
    q /= A.H;
    const int t = (int)(q % A.T);
    const long long frame0 = q - t;                          // first frame of this position's window
    const float *wrow = A.Wt + ((size_t)br * A.Fpad + nt * 32 + r) * A.kpad + 4 * hh;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (A.ntaps == 1) {                                      // Dense: one tap, the position's own row of C inputs
        const float *xp = A.X + (size_t)m * A.C + 4 * hh;
        const int kpart = A.kpad / (A.ksplit > 1 ? A.ksplit : 1), kend = ((int)blockIdx.z + 1) * kpart;
        for (int kk = (int)blockIdx.z * kpart; kk < kend; kk += 8) {
            const float4 a = *(const float4 *)(xp + kk);
            const float4 b = *(const float4 *)(wrow + kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
        }
    } else {
        for (int kk = 0; kk < A.kpad; kk += 8) {
            const int k0 = kk + 4 * hh, tap = k0 >> A.logC, c0 = k0 & (A.C - 1);
            const int kt = tap / 9, kh = (tap - 9 * kt) / 3, kw = tap - 9 * kt - 3 * kh;
            const int tt = t + (kt - 1) * d, yy = y + kh - 1, xx = x + kw - 1;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);     // SAME padding: zeros outside the window / the frame
            if (tap < 27 && (unsigned)tt < (unsigned)A.T && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W)
                a = *(const float4 *)(A.X + ((((size_t)(frame0 + tt)) * A.H + yy) * A.W + xx) * A.C + c0);
            const float4 b = *(const float4 *)(wrow + kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
        }
    }
    if (A.ksplit > 1) {                                      // a K part of Dense: raw sums, bias and ReLU are k_shot_head's
        if ((long long)blockIdx.x * 128 + wave * 32 + r >= A.M) return;
        float *yp = A.Y + ((size_t)blockIdx.z * (A.M / A.rtn * A.rT) + m) * A.ldy;      // part z of ALL rows (real row index)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4 *)(yp + nt * 32 + 8 * g + 4 * hh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        return;
    }
    if ((long long)blockIdx.x * 128 + wave * 32 + r >= A.M) return;
    float *yp = A.Y + (size_t)m * A.ldy + br * A.F;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = nt * 32 + 8 * g + 4 * hh;
        if (col >= A.F) continue;
        const float4 b = *(const float4 *)(A.bias + br * A.F + col);
        float4 v = make_float4(acc[4 * g] + b.x, acc[4 * g + 1] + b.y, acc[4 * g + 2] + b.z, acc[4 * g + 3] + b.w);
        if (A.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4 *)(yp + col) = v;
    }
}

// Both operands through LDS: per (tap, 64-channel slice) the workgroup's 128 positions x 64 channels and the NT x 32 output
// channels x 64 weights are fetched with whole-line loads (a wave's float4 load covers four complete 256-byte rows; the
// direct form reads 32 rows 16 bytes at a time: 4x the L1 transactions), parked in registers while the previous slice is
// multiplied, and written to LDS between two barriers.  NT = 2 (64-filter cells) shares the positions between two tiles.
// operand fetch / park of k_shot_conv_lds2 as functions over register arrays passed by reference (as lambdas capturing the
// arrays they kept part of them -- the weight registers -- in a private segment: 48 / 80 bytes of scratch per thread)
template <int NT>
__device__ __forceinline__ void shot_fetch(const ShotConv &A, int it, int nsl, int C, int d, int c4, int row0, const float *wbase,
                                           const int (&pxy)[8], const int (&pfr)[8], float4 (&areg)[8], float4 (&breg)[2 * NT]) {
    const int tap = it / nsl, sl = it - tap * nsl;
    const int kt = tap / 9, kh = (tap - 9 * kt) / 3, kw = tap - 9 * kt - 3 * kh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int x = (pxy[j] & 255) + kw - 1, y = ((pxy[j] >> 8) & 255) + kh - 1, t = (pxy[j] >> 16) + (kt - 1) * d;
        const bool ok = (unsigned)t < (unsigned)A.T && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
        const size_t pos = ok ? ((size_t)(pfr[j] + (kt - 1) * d) * A.H + y) * A.W + x : 0;
        const float4 v = *(const float4 *)(A.X + pos * C + sl * 64 + c4 * 4);
        areg[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);          // SAME padding
    }
#pragma unroll
    for (int j = 0; j < 2 * NT; ++j) {
        // (element-wise: a plain float4 copy from memory into the array is turned into a memcpy, which keeps the array
        // in a private segment)
        const float *wp = wbase + (size_t)(row0 + 16 * j) * A.kpad + tap * C + sl * 64 + c4 * 4;
        typedef float shot_f4 __attribute__((ext_vector_type(4)));
        const shot_f4 v = *(const shot_f4 *)wp;
        breg[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_shot_conv_lds2(const ShotConv A) {
    extern __shared__ float sm_shot2[];                      // As [128][68] | Bs [NT * 32][68]
    constexpr int WS = 68;
    float *As = sm_shot2, *Bs = sm_shot2 + 128 * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int groups = A.Fpad / (32 * NT), br = blockIdx.y / groups, ng = blockIdx.y - br * groups;
    const int d = 1 << br, C = A.C, HW = A.H * A.W, nsl = C >> 6;
    const long long m0 = (long long)blockIdx.x * 128;
    // fetch role: float4 c4 = tid & 15 of rows (tid >> 4) + 16 j, j = 0..7 (positions) / j = 0..2 NT - 1 (weights)
    const int c4 = tid & 15, row0 = tid >> 4;
    int pxy[8], pfr[8];                                      // x | y << 8 | t << 16 and the frame index of the thread's eight positions
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long m = std::min(m0 + row0 + 16 * j, A.M - 1);
        const long long fr = m / HW;
        const int rem = (int)(m - fr * HW), y = rem / A.W, x = rem - y * A.W;
        pxy[j] = x | (y << 8) | ((int)(fr % A.T) << 16);
        pfr[j] = (int)fr;
    }
    const float *wbase = A.Wt + ((size_t)br * A.Fpad + ng * 32 * NT) * A.kpad;
    float4 areg[8], breg[2 * NT];
#define SHOT_FETCH(it_) shot_fetch<NT>(A, it_, nsl, C, d, c4, row0, wbase, pxy, pfr, areg, breg)
    auto park = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) *(float4 *)(As + (row0 + 16 * j) * WS + c4 * 4) = areg[j];
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) *(float4 *)(Bs + (row0 + 16 * j) * WS + c4 * 4) = breg[j];
    };
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const int niter = 27 * nsl;
    SHOT_FETCH(0);
    for (int it = 0; it < niter; ++it) {
        __syncthreads();                                     // the previous slice's readers are done
        park();
        __syncthreads();
        if (it + 1 < niter) SHOT_FETCH(it + 1);              // in flight during this slice's MFMAs
        const float *ap = As + (wave * 32 + r) * WS + 4 * hh;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const float4 a = *(const float4 *)(ap + 8 * st);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b = *(const float4 *)(Bs + (n * 32 + r) * WS + 4 * hh + 8 * st);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[n], 0, 0, 0);
            }
        }
    }
    const long long m = m0 + wave * 32 + r;
    if (m >= A.M) return;
    float *yp = A.Y + (size_t)m * A.ldy + br * A.F;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (ng * NT + n) * 32 + 8 * g + 4 * hh;
            if (col >= A.F) continue;
            const float4 b = *(const float4 *)(A.bias + br * A.F + col);
            float4 v = make_float4(acc[n][4 * g] + b.x, acc[n][4 * g + 1] + b.y, acc[n][4 * g + 2] + b.z, acc[n][4 * g + 3] + b.w);
            if (A.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *(float4 *)(yp + col) = v;
        }
}

// The 16-filter cells (SDDCNN_1): a 32-column tile would be half padding -- 40 % of the network's executed FLOPs sit in
// SDDCNN_1/DDCNN_2 -- so the tile is 16 output channels on v_mfma_f32_16x16x4_f32, two 16-position tiles per wave.  A lane
// group g = lane / 16 owns k = 4 g .. 4 g + 3 of every 16-deep block (one float4 per operand; element e feeds the e-th
// MFMA, for both operands alike), and ends with channels 4 g .. 4 g + 3 of its position: one float4 store.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_shot_conv_lds2_n16(const ShotConv A) {
    extern __shared__ float sm_shot3[];                      // As [128][68] | Bs [16][68]
    constexpr int WS = 68;
    float *As = sm_shot3, *Bs = sm_shot3 + 128 * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int br = blockIdx.y, d = 1 << br, C = A.C, HW = A.H * A.W, nsl = C >> 6;
    const long long m0 = (long long)blockIdx.x * 128;
    const int c4 = tid & 15, row0 = tid >> 4;
    int pxy[8], pfr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long m = std::min(m0 + row0 + 16 * j, A.M - 1);
        const long long fr = m / HW;
        const int rem = (int)(m - fr * HW), y = rem / A.W, x = rem - y * A.W;
        pxy[j] = x | (y << 8) | ((int)(fr % A.T) << 16);
        pfr[j] = (int)fr;
    }
    const float *wbase = A.Wt + (size_t)br * A.Fpad * A.kpad;
    float4 areg[8], breg;
    auto fetch = [&](int it) {
        const int tap = it / nsl, sl = it - tap * nsl;
        const int kt = tap / 9, kh = (tap - 9 * kt) / 3, kw = tap - 9 * kt - 3 * kh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = (pxy[j] & 255) + kw - 1, y = ((pxy[j] >> 8) & 255) + kh - 1, t = (pxy[j] >> 16) + (kt - 1) * d;
            const bool ok = (unsigned)t < (unsigned)A.T && (unsigned)y < (unsigned)A.H && (unsigned)x < (unsigned)A.W;
            const size_t pos = ok ? ((size_t)(pfr[j] + (kt - 1) * d) * A.H + y) * A.W + x : 0;
            const float4 v = *(const float4 *)(A.X + pos * C + sl * 64 + c4 * 4);
            areg[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        breg = *(const float4 *)(wbase + (size_t)row0 * A.kpad + tap * C + sl * 64 + c4 * 4);       // 16 rows x 16 float4
    };
    f32x4 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
    const int niter = 27 * nsl;
    fetch(0);
    for (int it = 0; it < niter; ++it) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) *(float4 *)(As + (row0 + 16 * j) * WS + c4 * 4) = areg[j];
        *(float4 *)(Bs + row0 * WS + c4 * 4) = breg;
        __syncthreads();
        if (it + 1 < niter) fetch(it + 1);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float4 w = *(const float4 *)(Bs + p16 * WS + 16 * st + 4 * g);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float4 a = *(const float4 *)(As + (wave * 32 + n * 16 + p16) * WS + 16 * st + 4 * g);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc[n], 0, 0, 0);
            }
        }
    }
    const float4 b = *(const float4 *)(A.bias + br * 16 + 4 * g);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const long long m = m0 + wave * 32 + n * 16 + p16;
        if (m >= A.M) continue;
        float4 v = make_float4(acc[n][0] + b.x, acc[n][1] + b.y, acc[n][2] + b.z, acc[n][3] + b.w);
        if (A.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4 *)(A.Y + (size_t)m * A.ldy + br * 16 + 4 * g) = v;
    }
}

// MaxPool3D (1, 2, 2), VALID: [n][H][W][C] -> [n][H/2][W/2][C]
__global__ __launch_bounds__(256) void k_shot_pool(const float *__restrict__ X, float *__restrict__ Y, size_t total, int H, int W,
                                                   int C4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int OH = H / 2, OW = W / 2;
    const int c = (int)(i % C4);
    size_t q = i / C4;
    const int ox = (int)(q % OW);
    q /= OW;
    const int oy = (int)(q % OH);
    const size_t f = q / OH;
    const float4 *p = (const float4 *)X + ((f * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const float4 a = p[0], b = p[C4], e = p[(size_t)W * C4], g = p[(size_t)W * C4 + C4];
    float4 v;
    v.x = fmaxf(fmaxf(a.x, b.x), fmaxf(e.x, g.x));
    v.y = fmaxf(fmaxf(a.y, b.y), fmaxf(e.y, g.y));
    v.z = fmaxf(fmaxf(a.z, b.z), fmaxf(e.z, g.z));
    v.w = fmaxf(fmaxf(a.w, b.w), fmaxf(e.w, g.w));
    ((float4 *)Y)[i] = v;
}

// Dense(2) + softmax, class 1: one wavefront per frame.  nsplit > 0: X holds Dense(256)'s K parts [nsplit][rows][256] as raw
// sums -- added here in part order, then its bias b1 and ReLU.
__global__ __launch_bounds__(64) void k_shot_head(const float *__restrict__ X, const float *__restrict__ W2, const float *__restrict__ b2,
                                                  float *__restrict__ prob, int rows, int nsplit, const float *__restrict__ b1,
                                                  int T, int t0, int tn) {
    // blockIdx.x counts the rows that are computed (tn of every T: svc_transnet_predict_rows); rows = all rows (the parts' stride)
    const int row = (int)shot_real(blockIdx.x, T, t0, tn, 1), lane = threadIdx.x;
    if (row >= rows) return;
    float4 xv = *(const float4 *)(X + (size_t)row * SHOT_D + lane * 4);
    if (nsplit > 0) {
        for (int z = 1; z < nsplit; ++z) {
            const float4 p = *(const float4 *)(X + ((size_t)z * rows + row) * SHOT_D + lane * 4);
            xv.x += p.x; xv.y += p.y; xv.z += p.z; xv.w += p.w;
        }
        const float4 b = *(const float4 *)(b1 + lane * 4);
        xv = make_float4(fmaxf(xv.x + b.x, 0.f), fmaxf(xv.y + b.y, 0.f), fmaxf(xv.z + b.z, 0.f), fmaxf(xv.w + b.w, 0.f));
    }
    const float4 w0 = *(const float4 *)(W2 + lane * 4), w1 = *(const float4 *)(W2 + SHOT_D + lane * 4);
    float s0 = xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
    float s1 = xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (lane == 0) {
        const float l0 = s0 + b2[0], l1 = s1 + b2[1], mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        prob[row] = e1 / (e0 + e1);
    }
}

struct ShotX3 {
    const uint4 *X3;        // input planes [C / 16][3][Mp][2]
    const uint4 *W3;        // [branch][filter group][iteration = (q, kt, kh)][kw][NT tiles][NPL planes][64 lanes] uint4 (k_shot_x3m_weights)
    const float *bias;      // [branches * F]
    uint4 *Y3;              // output planes [4 F / 16][3][Mp][2]: branch br writes channels br * F ..
    long long M, Mp;        // positions the launch computes = B * tn * H * W (compact, see shot_real); plane stride
    int T, H, W, C, F, Fpad, relu, xcd;
    int t0, tn;             // frames t0 .. t0 + tn - 1 of every window (tn == T: all)
};

// The epilogue of the kernels that keep the kw taps in the accumulators: y[m] = P_1[m] + P_0[m - 1] + P_2[m + 1] (+ bias, ReLU)
// on the wave's inner positions, stored as split planes.  acc[pt][kw * NT + n], or (F16) tile 0 = kw 0 (rows 0..15) | kw 1
// (rows 16..31), tile KT - 1 = kw 2 | 0.  mb = the wave's first computed position, xs = the x of the lane's positions.
template <int NT, int PT, int KT, bool F16>
__device__ __forceinline__ void shot_kw_epilogue(const f32x16 (&acc)[PT][KT], uint4 *Y3, long long Mp, long long M, int W, int F,
                                                 const float *bias, int relu, int br, int ng, long long mb, int r, int hh, const int (&xs)[PT],
                                                 int T, int t0, int tn, int HW) {
    // lane (r, hh) of tile pt fetches lane r -+ 1 of the same half (ds_bpermute); r = 0 / 31 take the neighbouring tile's last / first lane
    const int lo = (hh * 32 + ((r + 31) & 31)) * 4, hi = (hh * 32 + ((r + 1) & 31)) * 4;
    auto from_left = [&](const f32x16 (&P)[PT], int pt, int i) -> float {       // P[m - 1]
        const float same = __int_as_float(__builtin_amdgcn_ds_bpermute(lo, __float_as_int(P[pt][i])));
        const float prev = pt > 0 ? __int_as_float(__builtin_amdgcn_ds_bpermute(lo, __float_as_int(P[pt > 0 ? pt - 1 : 0][i]))) : 0.f;
        return r == 0 ? prev : same;
    };
    auto from_right = [&](const f32x16 (&P)[PT], int pt, int i) -> float {      // P[m + 1]
        const float same = __int_as_float(__builtin_amdgcn_ds_bpermute(hi, __float_as_int(P[pt][i])));
        const float next = pt + 1 < PT ? __int_as_float(__builtin_amdgcn_ds_bpermute(hi, __float_as_int(P[pt + 1 < PT ? pt + 1 : pt][i]))) : 0.f;
        return r == 31 ? next : same;
    };
    constexpr int NOUT = F16 ? 1 : NT;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        f32x16 P0[PT], P1[PT], P2[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if (F16) {                                       // tile 0 = kw 0 (rows 0..15) | kw 1 (rows 16..31), tile 1 = kw 2 | 0
#pragma unroll
                for (int i = 0; i < 8; ++i) { P0[pt][i] = acc[pt][0][i]; P1[pt][i] = acc[pt][0][8 + i]; P2[pt][i] = acc[pt][KT - 1][i]; }
#pragma unroll
                for (int i = 8; i < 16; ++i) P0[pt][i] = P1[pt][i] = P2[pt][i] = 0.f;
            } else {
                P0[pt] = acc[pt][n]; P1[pt] = acc[pt][F16 ? 0 : NT + n]; P2[pt] = acc[pt][F16 ? 0 : 2 * NT + n];
            }
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const long long cm = mb + 32 * pt + r;
            const int idx = 32 * pt + r;
            const bool out = idx >= 1 && idx <= 32 * PT - 2 && cm < M;
            const long long m = out ? shot_real(cm, T, t0, tn, HW) : 0;
            const bool useL = xs[pt] >= 1, useR = xs[pt] <= W - 2;
            const int c0 = (ng * NT + n) * 32;               // first channel of the tile inside the branch
            float4 v[4];
#pragma unroll
            for (int g = 0; g < (F16 ? 2 : 4); ++g) {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float l = from_left(P0, pt, 4 * g + i), rr = from_right(P2, pt, 4 * g + i);
                    e[i] = (P1[pt][4 * g + i] + (useL ? l : 0.f)) + (useR ? rr : 0.f);
                }
                const int col = c0 + 8 * g + 4 * hh;
                const float4 b = col < F ? *(const float4 *)(bias + br * F + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[g] = make_float4(e[0] + b.x, e[1] + b.y, e[2] + b.z, e[3] + b.w);
                if (relu) { v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f); }
            }
            if (!out) continue;
#pragma unroll
            for (int ql = 0; ql < (F16 ? 1 : 2); ++ql)
                if (c0 + 16 * ql < F) shot_store_x3(Y3, Mp, m, hh, ((br * F + c0) >> 4) + ql, v[2 * ql], v[2 * ql + 1]);
        }
    }
}

// A DDCNN cell on split-bf16 operands: implicit GEMM [positions x 27 C] . [27 C x filters] on v_mfma_f32_16x16x32_bf16 (16 filters x
// 16 positions per MFMA, 32 channels deep), NP = 6 plane pairs per step (svc_x3.h; NP = 3: hi.hi + hi.mid + mid.hi only, SVC_SHOT_MX=bf16x3).
//
// What bounds this kernel is the delivery of the position operand (measured with one load per tap: 21 TB/s from L2 to the CUs' L1s in
// the 32-filter cells, the MFMAs at half their rate), so the three kw taps of a (kt, kh) share ONE load:
//   y[m] = sum_kw W_kw . x[m + kw - 1]  =  P_0[m - 1] + P_1[m] + P_2[m + 1],   P_kw[m'] = W_kw . x[m']  (un-shifted),
// i.e. a wavefront keeps one accumulator tile PER kw over the whole K loop, all three fed by the same position registers, and the
// shift by one position happens ONCE, on the accumulators, in the epilogue (ds_bpermute; the frame's left / right border masks the
// P_0 / P_2 term there; the (kt, kh) border is a property of m' and masks the load).  A wave computes 16 PT consecutive positions and
// emits the inner 16 PT - 2 (the two ends only serve their neighbours): 3 % more MFMAs for a third of the operand traffic.
// K order: the 32-channel group s OUTERMOST, then (kt, kh) = one iteration -- what the workgroups of an XCD have in flight is one
// group's planes of a band of frames (a few MB: it stays in the XCD's L2 while the taps re-read it; with the taps outermost the
// 128-channel cell took 2.29 ms, with the group outermost 1.00).
// Positions: straight from the planes into the MFMA registers, requested one iteration ahead.  Lane (p, g) = (lane & 15, lane >> 4)
// holds the eight channels of k group g of one position / filter; group g is the uint4 (q = 2 s + (g >> 1), hh = g & 1) of the planar
// layout, so the position operand is a plain load (two runs of 512 bytes per instruction).  Weights: the iteration's [tile][plane][64
// lanes] block is contiguous in W3 in exactly the order the lanes read it -- copied to LDS by all four waves one iteration ahead (double
// buffer, ONE barrier per iteration), read back lane-contiguous (conflict-free) one tile ahead.  A lane ends with filters 4 g .. 4 g + 3
// of one position: half a uint4 per plane (8-byte stores).
// Why this MFMA shape: 16-filter weight tiles fit the 16-filter cell exactly (three kw tiles; on 32x32x16 two 32-row tiles, one half
// empty: a quarter more MFMAs), and three 16-position tiles per wavefront keep two waves per SIMD with two filter tiles.
// NT (filter tiles per wavefront) = min(F / 16, 2): the 64-filter cells run their filter groups as separate workgroups.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int NT, int PT, int NP>
__global__ __launch_bounds__(256) void k_shot_conv_x3m(const ShotX3 A) {
    extern __shared__ uint4 sm_w3[];
    constexpr int NPL = NP == 3 ? 2 : 3;
    constexpr int KT = 3 * NT;                               // weight tiles (kw, n) of 16 filters per iteration
    constexpr int WCH = KT * NPL * 64;
    constexpr int WPT = (WCH + 255) / 256;
    constexpr int WS = 16 * PT - 2, WGS = 4 * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
    const int groups = A.F / (16 * NT), br = blockIdx.y / groups, ng = blockIdx.y - br * groups;
    const int d = 1 << br, HW = A.H * A.W, niter = 9 * (A.C >> 5);
    long long tile = blockIdx.x;
    if (A.xcd) tile = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (blockIdx.x == 0) shot_zero_pads(A.Y3, A.Mp, (br * A.F + ng * 16 * NT) >> 4, NT, tid);
    if (tile * WGS >= A.M) return;
    const long long mb = tile * WGS + wave * WS - 1;
    const unsigned planeB = (unsigned)(A.Mp * 32);
    const unsigned goff = (unsigned)((g & 1) * 16) + (unsigned)(g >> 1) * 3u * planeB;      // the lane's k group: half hh of group q + (g >> 1)
    unsigned vo[PT], okb[PT];
    int xs[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const long long m = mb + 16 * pt + p;
        const bool in = m >= 0 && m < A.M;
        const long long mm = in ? shot_real(m, A.T, A.t0, A.tn, HW) : 0, fr = mm / HW;
        const int rem = (int)(mm - fr * HW), y = rem / A.W, x = rem - y * A.W, t = (int)(fr % A.T);
        unsigned b = 0;
        for (int k = 0; k < 9; ++k) {
            const int kt = k / 3, kh = k - 3 * kt;
            const bool ok = in && (unsigned)(t + (kt - 1) * d) < (unsigned)A.T && (unsigned)(y + kh - 1) < (unsigned)A.H;
            b |= (unsigned)ok << k;
        }
        okb[pt] = b;
        xs[pt] = x;
        vo[pt] = (unsigned)((SHOT_PAD + mm) * 32) + goff;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)A.X3, 0, (int)((size_t)(A.C >> 4) * 3 * A.Mp * 32), 0x00020000);
    const uint4 *wsrc = A.W3 + (size_t)(br * groups + ng) * niter * WCH + tid;
    bf16x8 a[2][PT][NPL], bw[2][NPL];
#define SHOT_LDA(set_, it_)                                                                                          \
    {                                                                                                                \
        const int s_ = (it_) / 9, k_ = (it_) - 9 * s_, kt_ = k_ / 3, kh_ = k_ - 3 * kt_;                             \
        const int sh_ = (((kt_ - 1) * d * HW) + (kh_ - 1) * A.W) * 32;                                               \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) {                                                          \
            const unsigned v_ = ((it_) < niter && ((okb[pt] >> k_) & 1)) ? vo[pt] + sh_ : goff;                      \
            _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                     \
                X3Q t_;                                                                                              \
                const auto ld_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)v_, (int)((unsigned)(s_ * 6 + pl) * planeB), 0); \
                t_.u[0] = ld_[0]; t_.u[1] = ld_[1]; t_.u[2] = ld_[2]; t_.u[3] = ld_[3];                              \
                a[set_][pt][pl] = t_.v;                                                                              \
            }                                                                                                        \
        }                                                                                                            \
    }
#define SHOT_LDB(set_, buf_, wt_)                                                                                    \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                             \
        X3Q t_;                                                                                                      \
        t_.q = sm_w3[(buf_) * WCH + ((wt_) * NPL + pl) * 64 + lane];                                                 \
        bw[set_][pl] = t_.v;                                                                                         \
    }
#define SHOT_MMA(aset_, bset_, wt_)                                                                                  \
    {                                                                                                                \
        constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};                                        \
        _Pragma("unroll") for (int pr = (NP == 3 ? 3 : 0); pr < 6; ++pr)                                             \
            _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)                                                        \
                acc[pt][wt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[bset_][PW[pr]], a[aset_][pt][PA[pr]], acc[pt][wt_], 0, 0, 0); \
    }
    f32x4m acc[PT][KT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[pt][k][i] = 0.f;
    typedef unsigned shot_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < WPT; ++k)
        if (tid + 256 * k < WCH) ((shot_u4 *)sm_w3)[tid + 256 * k] = ((const shot_u4 *)wsrc)[256 * k];
    SHOT_LDA(0, 0);
    __syncthreads();
    SHOT_LDB(0, 0, 0);
#define SHOT_ITER(it_, sa_, b0_)                                                                                     \
    {                                                                                                                \
        const int cb_ = (it_) & 1;                                                                                   \
        shot_u4 wreg[WPT];                                                                                           \
        {                                                                                                            \
            const shot_u4 *wp = (const shot_u4 *)(wsrc + (size_t)((it_) + 1 < niter ? (it_) + 1 : (it_)) * WCH);     \
            _Pragma("unroll") for (int k = 0; k < WPT; ++k) wreg[k] = wp[tid + 256 * k < WCH ? 256 * k : 0];         \
        }                                                                                                            \
        SHOT_LDA((sa_) ^ 1, (it_) + 1);                                                                              \
        _Pragma("unroll") for (int wt = 0; wt < KT; ++wt) {                                                          \
            if (wt + 1 < KT) {                                                                                       \
                SHOT_LDB(((b0_) + wt + 1) & 1, cb_, wt + 1);                                                         \
            } else {                                                                                                 \
                shot_u4 *wn = (shot_u4 *)(sm_w3 + (cb_ ^ 1) * WCH + tid);                                            \
                _Pragma("unroll") for (int k = 0; k < WPT; ++k)                                                      \
                    if (tid + 256 * k < WCH) wn[256 * k] = wreg[k];                                                  \
                __syncthreads();                                                                                     \
                SHOT_LDB(((b0_) + wt + 1) & 1, cb_ ^ 1, 0);                                                          \
            }                                                                                                        \
            SHOT_MMA(sa_, ((b0_) + wt) & 1, wt);                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NPL; ++i_) {                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, NP * PT / NPL, 0);                                       \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
                if (wt * NPL + i_ < PT * NPL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     \
            }                                                                                                        \
        }                                                                                                            \
    }
    for (int it = 0; it < niter; it += 2) {                  // niter = 9 C / 32 is even (C = 64, 128, 256)
        SHOT_ITER(it, 0, 0);
        SHOT_ITER(it + 1, 1, KT & 1);
    }
#undef SHOT_ITER
#undef SHOT_MMA
#undef SHOT_LDB
#undef SHOT_LDA
    // epilogue: y[m] = P_1[m] + P_0[m - 1] + P_2[m + 1]; lane (p, g) fetches lane p -+ 1 of its group, p = 0 / 15 from the neighbouring tile
    const int lo = (g * 16 + ((p + 15) & 15)) * 4, hi = (g * 16 + ((p + 1) & 15)) * 4;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const long long cm = mb + 16 * pt + p;
            const int idx = 16 * pt + p;
            const bool out = idx >= 1 && idx <= 16 * PT - 2 && cm < A.M;
            const long long m = out ? shot_real(cm, A.T, A.t0, A.tn, HW) : 0;
            const bool useL = xs[pt] >= 1, useR = xs[pt] <= A.W - 2;
            const int c0 = (ng * NT + n) * 16 + 4 * g;       // the lane's four channels inside the branch
            const float4 b = *(const float4 *)(A.bias + br * A.F + c0);
            const float bb[4] = {b.x, b.y, b.z, b.w};
            uint32_t hw[4], mw[4], lw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float same_l = __int_as_float(__builtin_amdgcn_ds_bpermute(lo, __float_as_int(acc[pt][n][i])));
                const float prev_l = pt > 0 ? __int_as_float(__builtin_amdgcn_ds_bpermute(lo, __float_as_int(acc[pt > 0 ? pt - 1 : 0][n][i]))) : 0.f;
                const float same_r = __int_as_float(__builtin_amdgcn_ds_bpermute(hi, __float_as_int(acc[pt][2 * NT + n][i])));
                const float next_r = pt + 1 < PT ? __int_as_float(__builtin_amdgcn_ds_bpermute(hi, __float_as_int(acc[pt + 1 < PT ? pt + 1 : pt][2 * NT + n][i]))) : 0.f;
                const float l = p == 0 ? prev_l : same_l, rr = p == 15 ? next_r : same_r;
                float v = ((acc[pt][NT + n][i] + (useL ? l : 0.f)) + (useR ? rr : 0.f)) + bb[i];
                if (A.relu) v = fmaxf(v, 0.f);
                x3_split1<false>(v, hw[i], mw[i], lw[i]);
            }
            if (!out) continue;
            // channels 4 g .. 4 g + 3 of group q: half (g >> 1) of the uint4 of half-plane hh = g & 1
            uint2 *o = (uint2 *)(A.Y3 + ((size_t)((((br * A.F) >> 4) + ng * NT + n) * 3) * A.Mp + SHOT_PAD + m) * 2 + (g & 1)) + (g >> 1);
            o[0] = make_uint2(x3_pack(hw[0], hw[1]), x3_pack(hw[2], hw[3]));
            o[(size_t)A.Mp * 4] = make_uint2(x3_pack(mw[0], mw[1]), x3_pack(mw[2], mw[3]));
            o[(size_t)A.Mp * 8] = make_uint2(x3_pack(lw[0], lw[1]), x3_pack(lw[2], lw[3]));
        }
}

// k_shot_conv_x3m's weights: one thread per (branch, group, iteration = (s, kt, kh), weight tile (kw, n), lane (f, g))
__global__ __launch_bounds__(256) void k_shot_x3m_weights(const float *__restrict__ Wt, int Fpad, int kpad, int C, int F, int NT, int NPL,
                                                          uint4 *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int KT = 3 * NT;
    const int lane = (int)(i & 63), f = lane & 15, g = lane >> 4;
    size_t u = i >> 6;
    const int wt = (int)(u % KT); u /= KT;
    const int niter = 9 * (C >> 5), groups = F / (16 * NT);
    const int it = (int)(u % niter); u /= niter;
    const int ng = (int)(u % groups), br = (int)(u / groups);
    const int s = it / 9, k9 = it - 9 * s, kw = wt / NT, row = (ng * NT + wt % NT) * 16 + f;
    const float *w = Wt + ((size_t)br * Fpad + row) * kpad + (k9 * 3 + kw) * C + 16 * (2 * s + (g >> 1)) + 4 * (g & 1);
    const X3 sp = x3_split<true>(*(const float4 *)w, *(const float4 *)(w + 8));
    X3Q P[3];
    P[0].v = sp.h; P[1].v = sp.m; P[2].v = sp.l;
    uint4 *o = out + ((((size_t)(br * groups + ng) * niter + it) * KT + wt) * NPL) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        if (pl < NPL) o[pl * 64] = P[pl].q;
}

// ---- the first cell (3 input channels) on the same pipe --------------------------------------------------------------------
// Its input is the frames themselves: k_shot_in_x3 writes v / 255 as three bf16 planes [3][Mp][4 channels] (8 bytes per
// position and plane, channel 3 = 0, SHOT_PAD zero positions in front).  K = 27 taps x 3 channels is far too short per tap for
// a 16-deep step, so a step packs FOUR (kt, kh) pairs x 4 channels: pair pi = 4 q + 2 (j >> 2) + hh for element j of lane
// (r, hh) (9 pairs in 3 steps, the last three empty), the kw taps stay in the accumulators and the 16 filters pack as in
// a 16-filter packing -- tile 0 = kw 0's filters (rows 0..15) | kw 1's (rows 16..31), tile 1 = kw 2's | zeros: 3 steps x 2 tiles x NP
// MFMAs per 32 positions and dilation (fp32 form: 54 x 64-cycle MFMAs).
struct ShotFirst {
    const uint2 *X3;        // [3 planes][Mp][4 bf16]
    const uint4 *W3;        // [branch][3 steps][2 tiles][NPL planes][64 lanes] (k_shot_first_weights)
    const float *bias;
    uint4 *Y3;              // output planes [4][3][Mp][2] (64 channels: branch br = group br)
    long long M, Mp;
    int T, H, W, relu, xcd;
};

// uint8 [n][27][48][3] -> the three planes of v / 255 (tf.cast(float32) / 255., split exactly: svc_x3.h)
__global__ __launch_bounds__(256) void k_shot_in_x3(const uint8_t *__restrict__ in, uint2 *__restrict__ out, long long M, long long Mp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const uint8_t *p = in + i * 3;
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) x3_split1<false>((float)p[c] / 255.0f, h[c], m[c], l[c]);
    h[3] = m[3] = l[3] = 0;
    out[SHOT_PAD + i] = make_uint2(x3_pack(h[0], h[1]), x3_pack(h[2], h[3]));
    out[Mp + SHOT_PAD + i] = make_uint2(x3_pack(m[0], m[1]), x3_pack(m[2], m[3]));
    out[2 * Mp + SHOT_PAD + i] = make_uint2(x3_pack(l[0], l[1]), x3_pack(l[2], l[3]));
    if (i < SHOT_PAD) out[i] = out[Mp + i] = out[2 * Mp + i] = make_uint2(0, 0);
}

__global__ __launch_bounds__(256) void k_shot_first_weights(const float *__restrict__ Wt, int Fpad, int kpad, int NPL, uint4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;             // (branch, step, tile, lane)
    if (i >= 4 * 3 * 2 * 64) return;
    const int lane = i & 63, r = lane & 31, hh = lane >> 5, wt = (i >> 6) & 1, q = (i >> 7) % 3, br = i / (3 * 2 * 64);
    const int row = r & 15, kw = 2 * wt + (r >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pi = 4 * q + 2 * (j >> 2) + hh;             // (kt, kh) pair of this element, channel j & 3
        v[j] = (pi < 9 && kw < 3) ? Wt[((size_t)br * Fpad + row) * kpad + (pi * 3 + kw) * 4 + (j & 3)] : 0.f;
    }
    const X3 s = x3_split<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    X3Q P[3];
    P[0].v = s.h; P[1].v = s.m; P[2].v = s.l;
    uint4 *o = out + (size_t)(((br * 3 + q) * 2 + wt) * NPL) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        if (pl < NPL) o[pl * 64] = P[pl].q;
}

#define SHOT_FIRST_REP 4
template <int PT, int NP>
__global__ __launch_bounds__(256) void k_shot_first_x3(const ShotFirst A) {
    constexpr int NPL = NP == 3 ? 2 : 3, KT = 2, WCH = 3 * KT * NPL * 64;
    constexpr int WS = 32 * PT - 2, WGS = 4 * WS;
    __shared__ uint4 sm_w[WCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int br = blockIdx.y, d = 1 << br, HW = A.H * A.W;
    long long tile0 = blockIdx.x;
    if (A.xcd) tile0 = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (blockIdx.x == 0) shot_zero_pads(A.Y3, A.Mp, br, 1, tid);
    if (tile0 * SHOT_FIRST_REP * WGS >= A.M) return;
    for (int k = tid; k < WCH; k += 256) sm_w[k] = A.W3[(size_t)br * WCH + k];
    __syncthreads();
    // a workgroup's work on one tile is short (3 steps): SHOT_FIRST_REP consecutive tiles per workgroup share the weight
    // copy and the launch (228 -> ... us per 800 frames)
    for (int rep = 0; rep < SHOT_FIRST_REP; ++rep) {
    const long long tile = tile0 * SHOT_FIRST_REP + rep;
    if (tile * WGS >= A.M) break;
    const long long mb = tile * WGS + wave * WS - 1;
    int xs[PT];
    unsigned off[PT][3][2];                                  // byte offset of the lane's pair (q, jg) inside a plane, 0 = the zero position
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const long long m = mb + 32 * pt + r;
        const bool in = m >= 0 && m < A.M;
        const long long mm = in ? m : 0, fr = mm / HW;
        const int rem = (int)(mm - fr * HW), y = rem / A.W, x = rem - y * A.W, t = (int)(fr % A.T);
        xs[pt] = x;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int jg = 0; jg < 2; ++jg) {
                const int pi = 4 * q + 2 * jg + hh, kt = pi / 3, kh = pi - 3 * kt;
                const bool ok = in && pi < 9 && (unsigned)(t + (kt - 1) * d) < (unsigned)A.T && (unsigned)(y + kh - 1) < (unsigned)A.H;
                off[pt][q][jg] = ok ? (unsigned)((SHOT_PAD + mm + (long long)(kt - 1) * d * HW + (kh - 1) * A.W) * 8) : 0u;
            }
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)A.X3, 0, (int)(3 * A.Mp * 8), 0x00020000);
    const unsigned planeB = (unsigned)(A.Mp * 8);
    bf16x8 a[3][PT][NPL];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                X3Q t_;
                const auto lo = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)off[pt][q][0], (int)(pl * planeB), 0);
                const auto hi = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)off[pt][q][1], (int)(pl * planeB), 0);
                t_.u[0] = lo[0]; t_.u[1] = lo[1]; t_.u[2] = hi[0]; t_.u[3] = hi[1];
                a[q][pt][pl] = t_.v;
            }
    f32x16 acc[PT][KT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][k][i] = 0.f;
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int wt = 0; wt < KT; ++wt) {
            bf16x8 bw[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                X3Q t_;
                t_.q = sm_w[((q * KT + wt) * NPL + pl) * 64 + lane];
                bw[pl] = t_.v;
            }
#pragma unroll
            for (int pr = (NP == 3 ? 3 : 0); pr < 6; ++pr)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[pt][wt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[PW[pr]], a[q][pt][PA[pr]], acc[pt][wt], 0, 0, 0);
        }
    shot_kw_epilogue<1, PT, KT, true>(acc, A.Y3, A.Mp, A.M, A.W, 16, A.bias, A.relu, br, 0, mb, r, hh, xs, A.T, 0, A.T, HW);
    }
}

// MaxPool3D (1, 2, 2), VALID, on the planes: one thread per (q, output position, hh).  A value is the exact sum of its three
// planes, so the maximum is taken on the sums and split again (the same planes come out).  Yf: the last pool writes fp32
// [position][C] for Dense(256) instead.
__global__ __launch_bounds__(256) void k_shot_pool_x3(const uint4 *__restrict__ X3, long long Mp_in, uint4 *__restrict__ Y3, long long Mp_out,
                                                      float *__restrict__ Yf, size_t total, int H, int W, int Q, int T, int t0, int tn) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int OH = H / 2, OW = W / 2;
    const int hh = (int)(i & 1);
    size_t u = i >> 1;
    const size_t Mo = total / ((size_t)2 * Q);               // output positions of the frames that are computed (tn of every T)
    const size_t mc = u % Mo;
    const int q = (int)(u / Mo);
    const size_t mo = (size_t)shot_real((long long)mc, T, t0, tn, OH * OW);      // ... in the full-length planes
    const int ox = (int)(mo % OW);
    size_t v = mo / OW;
    const int oy = (int)(v % OH);
    const size_t f = v / OH;
    const size_t mi = (f * H + 2 * oy) * W + 2 * ox;
    float best[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 *p = X3 + ((size_t)(q * 3) * Mp_in + SHOT_PAD + mi + (k >> 1) * W + (k & 1)) * 2 + hh;
        const uint4 h = p[0], m = p[(size_t)Mp_in * 2], l = p[(size_t)Mp_in * 4];
        const uint32_t hu[4] = {h.x, h.y, h.z, h.w}, mu[4] = {m.x, m.y, m.z, m.w}, lu[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // element j = the low (even j) / high (odd j) half of word j / 2; a bf16 is the upper half of an f32
            const int w = j >> 1;
            const float x = (j & 1) ? (__uint_as_float(hu[w] & 0xffff0000u) + __uint_as_float(mu[w] & 0xffff0000u)) + __uint_as_float(lu[w] & 0xffff0000u)
                                    : (__uint_as_float(hu[w] << 16) + __uint_as_float(mu[w] << 16)) + __uint_as_float(lu[w] << 16);
            best[j] = k == 0 ? x : fmaxf(best[j], x);
        }
    }
    const float4 v0 = make_float4(best[0], best[1], best[2], best[3]), v1 = make_float4(best[4], best[5], best[6], best[7]);
    if (Yf) {
        float *o = Yf + mo * ((size_t)Q * 16) + 16 * q + 4 * hh;
        *(float4 *)o = v0;
        *(float4 *)(o + 8) = v1;
    } else {
        shot_store_x3(Y3, Mp_out, (long long)mo, hh, q, v0, v1);
        if (mc < SHOT_PAD) {
            uint4 *z = Y3 + ((size_t)(q * 3) * Mp_out + mc) * 2 + hh;
            z[0] = z[(size_t)Mp_out * 2] = z[(size_t)Mp_out * 4] = make_uint4(0, 0, 0, 0);
        }
    }
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct ShotCell { int cin, cpad, f, fpad, kpad; size_t w_off, b_off; };

static size_t shot_layout(ShotCell cells[SHOT_L * SHOT_S], size_t *d1w, size_t *d1b, size_t *d2w, size_t *d2b) {
    size_t o = 0;
    int cin = 3;
    for (int b = 0; b < SHOT_L; ++b)
        for (int c = 0; c < SHOT_S; ++c) {
            ShotCell &k = cells[b * SHOT_S + c];
            k.cin = cin; k.cpad = std::max(4, cin); k.f = SHOT_F << b; k.fpad = (k.f + 31) / 32 * 32;
            k.kpad = (27 * k.cpad + 7) / 8 * 8;
            k.w_off = o; o += (size_t)4 * k.fpad * k.kpad;
            k.b_off = o; o += (size_t)4 * k.f;
            cin = 4 * k.f;
        }
    const size_t nflat = (size_t)3 * 6 * 4 * (SHOT_F << (SHOT_L - 1));
    *d1w = o; o += (size_t)SHOT_D * nflat;
    *d1b = o; o += SHOT_D;
    *d2w = o; o += 2 * SHOT_D;
    *d2b = o; o += 2;
    return o;
}

extern "C" int svc_transnet_load(SvcHandle *h, const float *blob_host, size_t n_floats) {
    if (!h || !blob_host) { svc_set_error("svc_transnet_load: invalid argument"); return SVC_E_INVALID; }
    ShotCell cells[SHOT_L * SHOT_S];
    size_t a, b, c, d;
    // the packed layout keeps kpad == 27 * cpad except for the first cell (108 -> 112): the packer pads rows itself
    const size_t need = shot_layout(cells, &a, &b, &c, &d);
    if (n_floats != need) {
        svc_set_error("svc_transnet_load: blob has %zu floats, the F16 L3 S2 D256 network needs %zu", n_floats, need);
        return SVC_E_INVALID;
    }
    SVC_HIP(hipSetDevice(h->device));
    int rc = h->shot_blob.ensure(need * sizeof(float));
    if (rc) return rc;
    SVC_HIP(hipMemcpy(h->shot_blob.p, blob_host, need * sizeof(float), hipMemcpyHostToDevice));
    h->shot_loaded = true;
    h->shot_w3_mx = 0;                                       // the split copies are made again by the next predict
    return SVC_OK;
}

extern "C" int svc_transnet_matrix_pipe(const SvcHandle *h) { return h ? (h->shot_mx < 0 ? h->mx : h->shot_mx) : 0; }

// TransNet knobs of a handle as an array {matrix pipe (-1 = the handle's SVC_MX), 16-position tiles per wavefront, XCD-aware tile order}:
// what ShotTransNet.clone() copies to the engine of a second network instead of re-reading the environment
extern "C" int svc_transnet_config_get(const SvcHandle *h, int32_t *cfg3) {
    if (!h || !cfg3) { svc_set_error("svc_transnet_config_get: invalid argument"); return SVC_E_INVALID; }
    cfg3[0] = h->shot_mx; cfg3[1] = h->shot_m16; cfg3[2] = h->shot_xcd;
    return SVC_OK;
}
extern "C" int svc_transnet_config_set(SvcHandle *h, const int32_t *cfg3) {
    if (!h || !cfg3) { svc_set_error("svc_transnet_config_set: invalid argument"); return SVC_E_INVALID; }
    const int mx = cfg3[0], m16 = cfg3[1];
    if (!(mx == -1 || mx == 0 || mx == 3 || mx == 6) || m16 < 2 || m16 > 4) {
        svc_set_error("svc_transnet_config_set: {%d, %d, %d} is not a configuration", mx, m16, cfg3[2]);
        return SVC_E_INVALID;
    }
    h->shot_mx = mx; h->shot_m16 = m16; h->shot_xcd = cfg3[2] != 0;
    return SVC_OK;                                           // (the split weight copies follow at the next predict: shot_w3_mx keys them)
}

static int transnet_predict_rows(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, int row0, int row1, float *probs,
                                 void *stream);

extern "C" int svc_transnet_predict(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, float *probs,
                                    void *stream) {
    return transnet_predict_rows(h, frames, n_windows, frames_per_window, 0, frames_per_window, probs, stream);
}

extern "C" int svc_transnet_predict_rows(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, int row0, int row1,
                                         float *probs, void *stream) {
    return transnet_predict_rows(h, frames, n_windows, frames_per_window, row0, row1, probs, stream);
}

static int transnet_predict_rows(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, int row0, int row1, float *probs,
                                 void *stream) {
    if (!h || n_windows < 0 || frames_per_window < 1 || (n_windows > 0 && (!frames || !probs)) || row0 < 0 || row1 > frames_per_window || row0 >= row1) {
        svc_set_error("svc_transnet_predict: invalid argument");
        return SVC_E_INVALID;
    }
    if (!h->shot_loaded) { svc_set_error("svc_transnet_predict: no weights (svc_transnet_load)"); return SVC_E_INVALID; }
    if (n_windows == 0) return SVC_OK;
    SVC_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    ShotCell cells[SHOT_L * SHOT_S];
    size_t d1w, d1b, d2w, d2b;
    shot_layout(cells, &d1w, &d1b, &d2w, &d2b);
    const float *blob = (const float *)h->shot_blob.p;
    const int T = frames_per_window;
    const int mx = h->shot_mx < 0 ? h->mx : h->shot_mx;      // 0: fp32 MFMA; 6 / 3: split-bf16 planes (SVC_SHOT_MX, default = SVC_MX)
    const int NPL = mx == 3 ? 2 : 3;
    // frames [ca[i], cb[i]) of every window that cell i computes (shot_real): the last cell the kept rows, every cell in front of
    // it 8 more on either side (the fp32 pipe computes every frame; so does the first cell)
    int ca[SHOT_L * SHOT_S], cb[SHOT_L * SHOT_S];
    ca[SHOT_L * SHOT_S - 1] = mx ? row0 : 0;
    cb[SHOT_L * SHOT_S - 1] = mx ? row1 : T;
    for (int i = SHOT_L * SHOT_S - 2; i >= 0; --i) { ca[i] = std::max(0, ca[i + 1] - 8); cb[i] = std::min(T, cb[i + 1] + 8); }
    ca[0] = 0; cb[0] = T;
    const int ka = ca[SHOT_L * SHOT_S - 1], kn = cb[SHOT_L * SHOT_S - 1] - ka;        // Dense + head: the kept rows
    // split-bf16 weights of the cells with >= 64 input channels, packed once per load in the kernels' read order
    size_t w3_off[SHOT_L * SHOT_S] = {0};
    if (mx) {
        size_t tot = 0;
        // uint4 per cell: (4 branches) x (F / 16 filter tiles) x 3 kw x 9 (kt, kh) x (C / 32 groups) x planes x 64 lanes
        for (int i = 1; i < SHOT_L * SHOT_S; ++i) { w3_off[i] = tot; tot += (size_t)4 * (cells[i].f / 16) * 3 * 9 * (cells[i].cpad / 32) * NPL * 64; }
        const size_t w3_first = tot;                          // the first cell's block behind the others
        tot += (size_t)4 * 3 * 2 * NPL * 64;
        w3_off[0] = w3_first;
        if (h->shot_w3_mx != mx * 2 + h->shot_m16) {
            int rc = h->shot_w3.ensure(tot * sizeof(uint4));
            if (rc) return rc;
            k_shot_first_weights<<<(4 * 3 * 2 * 64 + 255) / 256, 256, 0, s>>>(blob + cells[0].w_off, cells[0].fpad, cells[0].kpad, NPL,
                                                                              (uint4 *)h->shot_w3.p + w3_first);
            SVC_CHECK_LAUNCH();
            for (int i = 1; i < SHOT_L * SHOT_S; ++i) {
                const ShotCell &k = cells[i];
                const size_t total = (size_t)4 * (k.f / 16) * 3 * 9 * (k.cpad / 32) * 64;        // threads: one per (tile, lane)
                k_shot_x3m_weights<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(blob + k.w_off, k.fpad, k.kpad, k.cpad, k.f, std::min(k.f / 16, 2), NPL,
                                                                                  (uint4 *)h->shot_w3.p + w3_off[i], total);
                SVC_CHECK_LAUNCH();
            }
            h->shot_w3_mx = mx * 2 + h->shot_m16;
        }
    }
    // windows per pass: two ping-pong activation buffers of T x 27 x 48 x 64 values per window (the largest tensor), fp32 or
    // three bf16 planes
    const size_t per_win = (size_t)T * SHOT_H * SHOT_W * 64;
    const size_t val_bytes = mx ? 6 : 4, cap = (size_t)(mx ? 1536 : 768) << 20;
    const int chunk = std::max(1, std::min(n_windows, (int)(cap / (2 * per_win * val_bytes))));
    const size_t buf_bytes = per_win * chunk * val_bytes + (mx ? (size_t)64 * 3 * (SHOT_PAD + 4) * 32 : 0);
    int rc = h->shot_ws.ensure(2 * buf_bytes);
    if (rc) return rc;
    float *P[2] = {(float *)h->shot_ws.p, (float *)((char *)h->shot_ws.p + buf_bytes)};
    for (int w0 = 0; w0 < n_windows; w0 += chunk) {
        const int nw = std::min(chunk, n_windows - w0);
        const size_t nfr = (size_t)nw * T;
        int H = SHOT_H, W = SHOT_W, cur = 0;
        const bool first_x3 = mx != 0;
        {
            const size_t npix = nfr * H * W;
            if (first_x3)
                k_shot_in_x3<<<(unsigned)((npix + 255) / 256), 256, 0, s>>>(frames + (size_t)w0 * T * H * W * 3, (uint2 *)P[0], (long long)npix,
                                                                           (long long)((npix + SHOT_PAD + 3) / 4 * 4));
            else
                k_shot_in<<<(unsigned)((npix + 255) / 256), 256, 0, s>>>(frames + (size_t)w0 * T * H * W * 3, P[0], npix);
            SVC_CHECK_LAUNCH();
        }
        for (int b = 0; b < SHOT_L; ++b) {
            const long long Mc = (long long)nfr * H * W, Mp = (Mc + SHOT_PAD + 3) / 4 * 4;       // plane stride of this resolution
            for (int c = 0; c < SHOT_S; ++c) {
                const ShotCell &k = cells[b * SHOT_S + c];
                if (first_x3 && b == 0 && c == 0) {
                    ShotFirst X;
                    X.X3 = (const uint2 *)P[cur]; X.W3 = (const uint4 *)h->shot_w3.p + w3_off[0]; X.bias = blob + k.b_off; X.Y3 = (uint4 *)P[cur ^ 1];
                    X.M = Mc; X.Mp = Mp; X.T = T; X.H = H; X.W = W; X.relu = 1; X.xcd = h->shot_xcd;
                    const int PT = 2, WGS = 4 * (32 * PT - 2) * SHOT_FIRST_REP;      // two 32-position tiles per wavefront (one: measured slower)
                    unsigned gx = (unsigned)((Mc + WGS - 1) / WGS);
                    if (X.xcd) gx = (gx + 7) / 8 * 8;
                    dim3 grid(gx, 4);
                    if (mx == 3) k_shot_first_x3<2, 3><<<grid, 256, 0, s>>>(X);
                    else k_shot_first_x3<2, 6><<<grid, 256, 0, s>>>(X);
                    SVC_CHECK_LAUNCH();
                    cur ^= 1;
                    continue;
                }
                if (mx && k.cpad >= 64) {
                    ShotX3 X;
                    X.X3 = (const uint4 *)P[cur]; X.W3 = (const uint4 *)h->shot_w3.p + w3_off[b * SHOT_S + c]; X.bias = blob + k.b_off;
                    X.Y3 = (uint4 *)P[cur ^ 1]; X.Mp = Mp; X.T = T; X.H = H; X.W = W; X.C = k.cpad; X.F = k.f; X.Fpad = k.fpad;
                    X.relu = 1; X.xcd = h->shot_xcd;
                    X.t0 = ca[b * SHOT_S + c]; X.tn = cb[b * SHOT_S + c] - X.t0; X.M = (long long)nw * X.tn * H * W;
                    // position tiles of 16 per wavefront: shot_m16 with two filter tiles, one more with one (the 16-filter cell: 1 167 us at
                    // three tiles, 1 066 at four; two filter tiles at four tiles leave one wave per SIMD: 465 / 872 -> 592 / 1 028 us)
                    const int NT = std::min(k.f / 16, 2), PT = NT == 1 ? std::min(h->shot_m16 + 1, 4) : h->shot_m16;
                    const int WGS = 4 * (16 * PT - 2);
                    unsigned gx = (unsigned)((X.M + WGS - 1) / WGS);
                    if (X.xcd) gx = (gx + 7) / 8 * 8;
                    dim3 grid(gx, (unsigned)(4 * (k.f / (16 * NT))));
                    const size_t lds = (size_t)2 * 3 * NT * NPL * 64 * sizeof(uint4);
#define SHOT_X3M(NT_, PT_, NP_) k_shot_conv_x3m<NT_, PT_, NP_><<<grid, 256, lds, s>>>(X)
#define SHOT_X3M_NP(NP_)                                                                                             \
                    if (NT == 1) { if (PT == 2) SHOT_X3M(1, 2, NP_); else if (PT == 3) SHOT_X3M(1, 3, NP_); else SHOT_X3M(1, 4, NP_); } \
                    else { if (PT == 2) SHOT_X3M(2, 2, NP_); else if (PT == 3) SHOT_X3M(2, 3, NP_); else SHOT_X3M(2, 4, NP_); }
                    if (mx == 3) { SHOT_X3M_NP(3) } else { SHOT_X3M_NP(6) }
#undef SHOT_X3M_NP
#undef SHOT_X3M
                    SVC_CHECK_LAUNCH();
                    cur ^= 1;
                    continue;
                }
                ShotConv A;
                A.X = P[cur]; A.Wt = blob + k.w_off; A.bias = blob + k.b_off; A.Y = P[cur ^ 1];
                A.M = (long long)nfr * H * W; A.T = T; A.H = H; A.W = W; A.C = k.cpad;
                A.logC = 0;
                while ((1 << A.logC) < k.cpad) ++A.logC;
                A.F = k.f; A.Fpad = k.fpad; A.kpad = k.kpad; A.ntaps = 27; A.ldy = 4 * k.f; A.relu = 1;
                A.ksplit = 1; A.rT = A.rtn = 1; A.rt0 = 0;
                dim3 grid((unsigned)((A.M + 127) / 128), (unsigned)(4 * (k.fpad / 32)));
                if (k.cpad >= 64 && k.cpad % 64 == 0) {            // the cells: both operands through LDS
                    if (k.f == 16) {
                        dim3 g16(grid.x, 4);
                        k_shot_conv_lds2_n16<<<g16, 256, (128 + 16) * 68 * sizeof(float), s>>>(A);
                    } else if (k.fpad % 64 == 0) {
                        dim3 g2(grid.x, (unsigned)(4 * (k.fpad / 64)));
                        k_shot_conv_lds2<2><<<g2, 256, (128 + 64) * 68 * sizeof(float), s>>>(A);
                    } else {
                        k_shot_conv_lds2<1><<<grid, 256, (128 + 32) * 68 * sizeof(float), s>>>(A);
                    }
                } else {                                           // the first cell (3 input channels padded to 4)
                    k_shot_conv<<<grid, 256, 0, s>>>(A);
                }
                SVC_CHECK_LAUNCH();
                cur ^= 1;
            }
            const int C = 4 * (SHOT_F << b);
            if (mx) {
                const long long Mo = (long long)nfr * (H / 2) * (W / 2), Mpo = (Mo + SHOT_PAD + 3) / 4 * 4;
                const int pa = ca[b * SHOT_S + SHOT_S - 1], pn = cb[b * SHOT_S + SHOT_S - 1] - pa;    // the frames the block's last cell computed
                const size_t total = (size_t)nw * pn * (H / 2) * (W / 2) * (C / 16) * 2;
                const bool last = b == SHOT_L - 1;
                k_shot_pool_x3<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const uint4 *)P[cur], Mp, (uint4 *)P[cur ^ 1], Mpo,
                                                                                last ? P[cur ^ 1] : nullptr, total, H, W, C / 16, T, pa, pn);
            } else {
                const size_t total = nfr * (H / 2) * (W / 2) * (C / 4);
                k_shot_pool<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(P[cur], P[cur ^ 1], total, H, W, C / 4);
            }
            SVC_CHECK_LAUNCH();
            cur ^= 1;
            H /= 2; W /= 2;
        }
        {
            ShotConv A;
            const int nflat = H * W * 4 * (SHOT_F << (SHOT_L - 1));
            A.X = P[cur]; A.Wt = blob + d1w; A.bias = blob + d1b; A.Y = P[cur ^ 1];
            A.M = (long long)nw * kn; A.rT = T; A.rt0 = ka; A.rtn = kn;      // the kept rows of every window
            A.T = 1; A.H = 1; A.W = 1; A.C = nflat; A.logC = 0;
            A.F = SHOT_D; A.Fpad = SHOT_D; A.kpad = nflat; A.ntaps = 1; A.ldy = SHOT_D; A.relu = 1;
            // 7 x 8 workgroups of 576 steps leave the chip idle (184 us per 800 frames): K in eight parts, summed by the head
            A.ksplit = nflat % (8 * 8) == 0 ? 8 : 1;
            dim3 grid((unsigned)((A.M + 127) / 128), (unsigned)(SHOT_D / 32), (unsigned)A.ksplit);
            k_shot_conv<<<grid, 256, 0, s>>>(A);
            SVC_CHECK_LAUNCH();
            cur ^= 1;
            k_shot_head<<<(unsigned)(nw * kn), 64, 0, s>>>(P[cur], blob + d2w, blob + d2b, probs + (size_t)w0 * T, (int)nfr,
                                                           A.ksplit > 1 ? A.ksplit : 0, blob + d1b, T, ka, kn);
            SVC_CHECK_LAUNCH();
        }
    }
    return SVC_OK;
}
