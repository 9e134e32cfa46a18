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
// Dense(256) is the same kernel with one tap over the 4608 flattened inputs.
#include <algorithm>

#include "svc_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

__global__ __launch_bounds__(256) void k_shot_conv(const ShotConv A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const long long m = std::min((long long)blockIdx.x * 128 + wave * 32 + r, A.M - 1);
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
        for (int kk = 0; kk < A.kpad; kk += 8) {
            const float4 a = *(const float4 *)(xp + kk);
            const float4 b = *(const float4 *)(wrow + kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
        }
    } else if (A.C >= 8) {
        // 27 taps x C / 8 steps.  Measured first: with the operand loads taken out the loop runs at 102 TFLOP/s, with a load
        // guarded by the tap's bounds test in front of every step's MFMAs at 57 -- the wave waits for each load.  So a tap's
        // source row is decoded once, the loads are unconditional (an out-of-range tap reads a valid dummy address and is
        // zeroed by a select: SAME padding), and the next step's operands are requested before this step's MFMAs.
        const int spt = A.C >> 3;
        auto tap_ptr = [&](int tap, bool &ok) -> const float * {
            const int kt = tap / 9, kh = (tap - 9 * kt) / 3, kw = tap - 9 * kt - 3 * kh;
            const int tt = t + (kt - 1) * d, yy = y + kh - 1, xx = x + kw - 1;
            ok = (unsigned)tt < (unsigned)A.T && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
            return ok ? A.X + ((((size_t)(frame0 + tt)) * A.H + yy) * A.W + xx) * A.C + 4 * hh : A.X + 4 * hh;
        };
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        bool okc, okn = false;
        const float *pc = tap_ptr(0, okc), *pn = pc;
        float4 an = *(const float4 *)pc, bn = *(const float4 *)wrow;
        if (!okc) an = z4;
        int kk = 0;
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) pn = tap_ptr(tap + 1, okn);
            for (int st = 0; st < spt; ++st) {
                const float4 a = an, b = bn;
                if (st + 1 < spt) {
                    an = *(const float4 *)(pc + 8 * (st + 1));
                    if (!okc) an = z4;
                } else if (tap + 1 < 27) {
                    an = *(const float4 *)pn;
                    if (!okn) an = z4;
                }
                kk += 8;
                if (kk < A.kpad) bn = *(const float4 *)(wrow + kk);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
            }
            pc = pn; okc = okn;
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

// The 27-tap cells with C >= 64 input channels: the tap's [32 output channels x C] slice of the weights goes through LDS
// (double-buffered, fetched with whole-line loads by all four waves, one barrier per tap) instead of every wave reading
// its 32 weight rows 16 bytes at a time: k_shot_conv's loads, not its MFMAs, set its pace (with the operand loads taken
// out it runs at 102 TFLOP/s, with them at 57: a wave-wide float4 load of 32 different rows is 32 L1 transactions), and
// the weight operand is the half of them that the four waves of a workgroup share.
__global__ __launch_bounds__(256) void k_shot_conv_lds(const ShotConv A) {
    extern __shared__ float sm_shot[];                       // [2][32][C + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const long long m = std::min((long long)blockIdx.x * 128 + wave * 32 + r, A.M - 1);
    const int tiles = A.Fpad >> 5, br = blockIdx.y / tiles, nt = blockIdx.y - br * tiles;
    const int d = 1 << br;
    const int x = (int)(m % A.W);
    long long q = m / A.W;
    const int y = (int)(q % A.H);
    q /= A.H;
    const int t = (int)(q % A.T);
    const long long frame0 = q - t;
    const int C = A.C, WS = C + 4, spt = C >> 3, c4n = C >> 2;
    const int per_thread = (32 * c4n) >> 8;                  // float4 of a weight slice per thread: 2 (C = 64), 4, 8
    const float *wbase = A.Wt + ((size_t)br * A.Fpad + nt * 32) * A.kpad;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto tap_ptr = [&](int tap, bool &ok) -> const float * {
        const int kt = tap / 9, kh = (tap - 9 * kt) / 3, kw = tap - 9 * kt - 3 * kh;
        const int tt = t + (kt - 1) * d, yy = y + kh - 1, xx = x + kw - 1;
        ok = (unsigned)tt < (unsigned)A.T && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
        return ok ? A.X + ((((size_t)(frame0 + tt)) * A.H + yy) * A.W + xx) * C + 4 * hh : A.X + 4 * hh;
    };
    typedef float shot_f4 __attribute__((ext_vector_type(4)));
    shot_f4 wreg[8];                                         // (an ext_vector type: a float4 struct array indexed under a run-time bound became a 144-byte private segment)
    auto fetch_w = [&](int tap) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < per_thread) {
                const int idx = tid + 256 * i, row = idx / c4n, c4 = idx - row * c4n;
                wreg[i] = *(const shot_f4 *)(wbase + (size_t)row * A.kpad + tap * C + c4 * 4);
            }
    };
    auto store_w = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < per_thread) {
                const int idx = tid + 256 * i, row = idx / c4n, c4 = idx - row * c4n;
                *(shot_f4 *)(dst + row * WS + c4 * 4) = wreg[i];
            }
    };
    fetch_w(0);
    store_w(sm_shot);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool okc, okn = false;
    const float *pc = tap_ptr(0, okc), *pn = pc;
    float4 an = *(const float4 *)pc;
    if (!okc) an = z4;
    for (int tap = 0; tap < 27; ++tap) {
        float *Bs = sm_shot + (tap & 1) * 32 * WS;
        __syncthreads();                                     // this tap's slice is in LDS; the other buffer's readers are done
        if (tap + 1 < 27) { fetch_w(tap + 1); pn = tap_ptr(tap + 1, okn); }
        const float *bp = Bs + r * WS + 4 * hh;
        for (int st = 0; st < spt; ++st) {
            const float4 a = an;
            const float4 b = *(const float4 *)(bp + 8 * st);
            if (st + 1 < spt) {
                an = *(const float4 *)(pc + 8 * (st + 1));
                if (!okc) an = z4;
            } else if (tap + 1 < 27) {
                an = *(const float4 *)pn;
                if (!okn) an = z4;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
        }
        if (tap + 1 < 27) store_w(sm_shot + ((tap + 1) & 1) * 32 * WS);
        pc = pn; okc = okn;
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

// Dense(2) + softmax, class 1: one wavefront per frame
__global__ __launch_bounds__(64) void k_shot_head(const float *__restrict__ X, const float *__restrict__ W2, const float *__restrict__ b2,
                                                  float *__restrict__ prob, int rows) {
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row >= rows) return;
    const float4 xv = *(const float4 *)(X + (size_t)row * SHOT_D + lane * 4);
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
    return SVC_OK;
}

extern "C" int svc_transnet_predict(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, float *probs,
                                    void *stream) {
    if (!h || n_windows < 0 || frames_per_window < 1 || (n_windows > 0 && (!frames || !probs))) {
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
    // windows per pass: two ping-pong activation buffers of T x 27 x 48 x 64 floats per window (the largest tensor)
    const size_t per_win = (size_t)T * SHOT_H * SHOT_W * 64;
    const int chunk = std::max(1, std::min(n_windows, (int)(((size_t)768 << 20) / (2 * per_win * sizeof(float)))));
    int rc = h->shot_ws.ensure(2 * per_win * chunk * sizeof(float));
    if (rc) return rc;
    float *P[2] = {(float *)h->shot_ws.p, (float *)h->shot_ws.p + per_win * chunk};
    for (int w0 = 0; w0 < n_windows; w0 += chunk) {
        const int nw = std::min(chunk, n_windows - w0);
        const size_t nfr = (size_t)nw * T;
        int H = SHOT_H, W = SHOT_W, cur = 0;
        {
            const size_t npix = nfr * H * W;
            k_shot_in<<<(unsigned)((npix + 255) / 256), 256, 0, s>>>(frames + (size_t)w0 * T * H * W * 3, P[0], npix);
            SVC_CHECK_LAUNCH();
        }
        for (int b = 0; b < SHOT_L; ++b) {
            for (int c = 0; c < SHOT_S; ++c) {
                const ShotCell &k = cells[b * SHOT_S + c];
                ShotConv A;
                A.X = P[cur]; A.Wt = blob + k.w_off; A.bias = blob + k.b_off; A.Y = P[cur ^ 1];
                A.M = (long long)nfr * H * W; A.T = T; A.H = H; A.W = W; A.C = k.cpad;
                A.logC = 0;
                while ((1 << A.logC) < k.cpad) ++A.logC;
                A.F = k.f; A.Fpad = k.fpad; A.kpad = k.kpad; A.ntaps = 27; A.ldy = 4 * k.f; A.relu = 1;
                dim3 grid((unsigned)((A.M + 127) / 128), (unsigned)(4 * (k.fpad / 32)));
                const int shot_form = h->shot_form;             // 0: direct operand loads, 1: weights through LDS, 2: both operands (SVC_SHOT_FORM)
                if (shot_form == 2 && k.cpad >= 64 && k.cpad % 64 == 0) {
                    if (k.f == 16) {
                        dim3 g16(grid.x, 4);
                        k_shot_conv_lds2_n16<<<g16, 256, (128 + 16) * 68 * sizeof(float), s>>>(A);
                    } else if (k.fpad % 64 == 0) {
                        dim3 g2(grid.x, (unsigned)(4 * (k.fpad / 64)));
                        k_shot_conv_lds2<2><<<g2, 256, (128 + 64) * 68 * sizeof(float), s>>>(A);
                    } else {
                        k_shot_conv_lds2<1><<<grid, 256, (128 + 32) * 68 * sizeof(float), s>>>(A);
                    }
                } else if (shot_form >= 1 && k.cpad >= 64 && k.cpad <= 256) {
                    const size_t lds = (size_t)2 * 32 * (k.cpad + 4) * sizeof(float);
                    if (lds > 64 * 1024 && h->lds_attr_done.insert((const void *)k_shot_conv_lds).second)
                        SVC_HIP(hipFuncSetAttribute((const void *)k_shot_conv_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                    k_shot_conv_lds<<<grid, 256, lds, s>>>(A);
                } else {
                    k_shot_conv<<<grid, 256, 0, s>>>(A);
                }
                SVC_CHECK_LAUNCH();
                cur ^= 1;
            }
            const int C = 4 * (SHOT_F << b);
            const size_t total = nfr * (H / 2) * (W / 2) * (C / 4);
            k_shot_pool<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(P[cur], P[cur ^ 1], total, H, W, C / 4);
            SVC_CHECK_LAUNCH();
            cur ^= 1;
            H /= 2; W /= 2;
        }
        {
            ShotConv A;
            const int nflat = H * W * 4 * (SHOT_F << (SHOT_L - 1));
            A.X = P[cur]; A.Wt = blob + d1w; A.bias = blob + d1b; A.Y = P[cur ^ 1];
            A.M = (long long)nfr; A.T = 1; A.H = 1; A.W = 1; A.C = nflat; A.logC = 0;
            A.F = SHOT_D; A.Fpad = SHOT_D; A.kpad = nflat; A.ntaps = 1; A.ldy = SHOT_D; A.relu = 1;
            dim3 grid((unsigned)((A.M + 127) / 128), (unsigned)(SHOT_D / 32));
            k_shot_conv<<<grid, 256, 0, s>>>(A);
            SVC_CHECK_LAUNCH();
            cur ^= 1;
            k_shot_head<<<(unsigned)nfr, 64, 0, s>>>(P[cur], blob + d2w, blob + d2b, probs + (size_t)w0 * T, (int)nfr);
            SVC_CHECK_LAUNCH();
        }
    }
    return SVC_OK;
}
