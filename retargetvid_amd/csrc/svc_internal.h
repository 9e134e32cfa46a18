// svc_internal.h — handle layout and helpers shared by svc_net.hip / svc_tail.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/svc.h"

void svc_set_error(const char *fmt, ...);

// Division of a 31-bit index by a launch-invariant divisor as multiply + shift (a runtime integer
// division costs ~40 VALU instructions on gfx950, more than the arithmetic of most element-wise
// kernels here).  Exact for n < 2^31, 1 <= d < 2^16.
struct FDiv {
    uint64_t m;
    uint32_t d;
    int s;
};
static inline FDiv make_fdiv(uint32_t d) {
    FDiv f;
    f.d = d;
    int l = 0;
    while ((1u << l) < d) ++l;
    f.s = 32 + l;
    f.m = (((uint64_t)1 << f.s) + d - 1) / d;
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FDiv &f) { return (uint32_t)(((uint64_t)n * f.m) >> f.s); }
__device__ __forceinline__ uint32_t fdivmod(uint32_t n, const FDiv &f, uint32_t &rem) {
    const uint32_t q = fdiv(n, f);
    rem = n - q * f.d;
    return q;
}
#endif

// SVC_NO_PK: a kernel compiled WITHOUT packed-f32 instructions (v_pk_mul/add/fma_f32).  Round 5 found one product lost in a
// `v_pk_mul_f32 x2 ; v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` sequence (a packed instruction that routes the halves of an
// operand CROSSWISE) of the smoothing kernel when a bf16-MFMA workgroup shared the CU (svc_net.hip, sd_bilinear); the trigger is not
// fully understood, so no kernel of the library may contain a half-swapping packed f32 instruction: the kernels whose C code the
// compiler packs that way carry this attribute (the whole library without packed f32 is 1 - 3 % slower: measured, round 6), and
// tools/packed_f32_census.py (run by tests/test_kernel_specs.py) fails the build check if one appears.  Device pass only: the
// host compiler does not know the feature.
#if defined(__HIP_DEVICE_COMPILE__)
#define SVC_NO_PK __attribute__((target("no-packed-fp32-ops")))
#else
#define SVC_NO_PK
#endif

#define SVC_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            svc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SVC_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define SVC_CHECK_LAUNCH()  SVC_HIP(hipGetLastError())

struct SvcTensor {
    const float *dev;     // device pointer into the blob
    size_t n;             // floats
};

// One folded layer of the static SALICON graph, in execution order (weights.fold_state_dict).
struct SvcLayer {
    enum Kind { STEM, PW, DW, GAUSS, ADAPT, SMOOTH } kind;
    int cin, cout, stride, relu6;
    SvcTensor w, b;
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return SVC_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { svc_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); return SVC_E_NOMEM; }
        bytes = need;
        return SVC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Fixed-point resampling tables for one (in, out) size pair, device resident.
struct ResampleTab {
    int ksize = 0;
    DevBuf bounds;    // int32[out][2]  (first, count)
    DevBuf coeff;     // int32[out][ksize]
    int max_span = 0; // max over blocks of rows spanned (host-computed helper)
    std::vector<int> bounds_host;
};

struct NetPlan;     // svc_net.hip

struct SvcHandle {
    int device = 0;
    DevBuf blob;
    std::vector<SvcTensor> tensors;
    std::vector<SvcLayer> layers;
    std::vector<float> gauss_params;      // coarse_gaussians_salicon [16][2][2]
    // network workspace + plan (svc_net.hip)
    NetPlan *plan = nullptr;
    // ingest resize tables keyed by (in_h, in_w, out_h, out_w)
    std::map<std::tuple<int, int, int, int>, DevBuf> cvtabs;
    // tail workspace (svc_tail.hip)
    DevBuf tail_ws;
    DevBuf tail_offsets;     // ring-walk offset table
    DevBuf tail_ring_cnt;    // u16[RING_R^2 + 1]: number of offsets with d2 <= index
    int tail_n_offsets = 0, tail_n_offsets1 = 0;
    std::vector<uint32_t> tail_offsets_host;
    std::map<int, DevBuf> tail_delta;  // dr * width + dc of every offset, one table per map width (never rewritten:
                                       // a call still in flight on the stream may be reading the one it was given)
    int tail_frames = 0, tail_h = 0, tail_w = 0;
    size_t tail_frame_stride = 0;      // bytes of per-frame tail workspace
    std::vector<int> tail_slot_of;     // last call: map -> workspace slot (-1: held)
    DevBuf rs_maps;                    // resize_factor != 1: the shrunk maps
    std::map<std::tuple<int, int, int>, std::pair<DevBuf, DevBuf>> rs_tabs;   // (h, w, factor) -> INTER_LINEAR tables down / up
    uint8_t *depth_pinned = nullptr;   // pinned staging ring for the per-map round numbers
    unsigned depth_slot = 0;
    hipEvent_t depth_ev[8] = {};       // recorded behind the upload of a slot; waited on before the slot is rewritten
    std::map<std::tuple<const void *, int, int, int>, DevBuf> lane_w;   // split-K layers' weights in lane order (svc_net.hip: lane_weights), keyed by (matrix, row stride, K, padded N)
    std::map<std::tuple<const void *, int, int, int>, DevBuf> x3_w;     // split-bf16 copies of weight matrices (svc_net.hip: x3_weights), keyed by (matrix, row stride, K, 2 * padded N + order)
    int mx = 6;                        // matrix pipe of the 1x1-convolution GEMMs: 6 = split-bf16 operands, six plane pairs on v_mfma_f32_32x32x16_bf16 (round 5, the default: a pass 1.42 -> 1.24 ms alone, 1.01 -> 0.87 ms with four passes sharing the chip, every parity gate unchanged); 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, rounds 1-4: SVC_MX=f32).  The one kernel found to miscompute beside bf16 workgroups, the smoothing kernel, lost a product in a packed-instruction sequence of its bilinear stage: written with scalar instructions since (sd_bilinear)
    unsigned irb_mx = 0x1b;            // ... which of k_irb's five fixed-shape instances take that form for their expand GEMM (bit = block 2, 3, 4, 5-6, 7; SVC_IRB_MX).  Measured per instance against the fp32 form, us per pass alone / shared: -17 / -11, -12 / -7, +11 / +11 (block 4: Cin = 24 pads its second step, two halo tiles per wave: 36 spilled registers), -11 / -4, -10 / -6: block 4 stays fp32
    unsigned mx_mask = 0xff;           // ... and which kernel families: bit 0 k_pwr, 1 k_irb, 2 k_dwpw, 3 k_pw_sk, 4 k_pwpw (SVC_MX_MASK; for A/B timing)
    bool sk_lane = true;               // k_pw_sk reads its weights from the lane-order copy: a wave's load is 1 KB contiguous instead of 32 rows x 32 B (SVC_SK_LANE=0: from the [N][K] matrix)
    std::set<const void *> lds_attr_done;   // kernels whose dynamic-LDS limit has been raised on this handle's device
    int chunk = 32;                    // frames per network pass
    DevBuf census;                     // threshold census (svc_threshold_census): [4] u64 totals, then [chunk][4] u32 per-frame counts of the last pass
    unsigned long long census_maps = 0;
    int pw_min_wg = 1024;              // k_pw narrows its column tile until the grid has this many workgroups (SVC_PW_MIN_WG)
    bool pw_sk = true;                 // split-K pointwise kernel for long-K small-M layers (SVC_PW_SK=0 disables)
    int pw_tr = 2;                     // k_pw with swapped MFMA operands (a lane owns one pixel, float4 epilogue): 0 never, 1 always, 2 for wave tiles of 2+ column tiles and up-sample-add launches (single-tile launches store whole 128 B lines with the scalar form)
    int pw_sk_max = 2048;              // ... when row blocks x column tiles (at the nominal batch) do not exceed this (SVC_PW_SK_MAX)
    int pw_small = 0;                  // small-M pointwise layers on 16-row wave tiles (SVC_PW_SMALL: 0 off, 1: 16x32, 2: 16x64, 3: 32x32)
    bool pwr = true;                   // short-K pointwise layers with the activations resident in registers and the weight chunk in LDS (SVC_PWR=0: k_pw)
    int pwr_nt = 2;                    // ... column tiles (of 32) per workgroup (measured at B = 32: 1 / 2 / 3 / 4 -> 1.735 / 1.692 / 1.726 / 1.767 ms per pass); 0 = as many as leave pwr_min_wg workgroups (SVC_PWR_NT)
    int pwr_min_wg = 512;              // (SVC_PWR_MIN_WG)
    bool pw16 = true;                  // 16x16x4 MFMA pointwise kernel for narrow short-K layers (SVC_PW16=0: always 32x32x2)
    int fuse_max = 7;                  // backbone blocks 1..fuse_max run as the fused inverted-residual kernel (SVC_FUSE_MAX, 0..13)
    bool split_up = true;              // decoder expansions as conv(skip) + up-sample(conv(low-res part)) (SVC_SPLIT_UP=0: up-sample, concatenate, one GEMM)
    bool irb_fixed = true;             // fused blocks of the six MobileNetV2 shapes run compile-time-shaped instances (SVC_IRB_FIXED=0: generic)
    DevBuf stem_wt;                    // stem weights transposed to [32 out][32 taps, 27 used] for the MFMA stem
    int smooth_mfma = 1;               // 41x41 smoothing phases as a GEMM on the matrix cores (SVC_SMOOTH_MFMA=0: the FMA kernel)
    bool front = true;                 // LANCZOS + features.0 + features.1 as one kernel, k_front (SVC_FRONT=0: three kernels)
    bool keep_input = false;           // ... which then also writes the normalised network input for svc_debug_tap(SVC_TAP_INPUT) (SVC_KEEP_INPUT=1)
    bool dwpw = true;                  // depthwise 3x3 fused into the following 1x1 project (SVC_DWPW=0: two kernels)
    bool pwpw = true;                  // the skip branches' two 1x1 convolutions (ReLU6 between) as one launch, k_pwpw: the intermediate tensor stays in registers (SVC_PWPW=0: two launches)
    unsigned seg_off = 0;              // MEASUREMENT AID (SVC_SEG_OFF=bitmask): stages of the network pass whose launches are skipped -- the maps are then garbage; tools/time_segments.py prices a stage by leaving it out.  Stages: 0 front, 1 blocks 2-3, 2 blocks 4-7, 3 blocks 8-14, 4 blocks 15-17, 5 features.18 + skips + post_cnn, 6 upsampling block 1, 7 upsampling block 2, 8 adaptation / smoothing / quantisation
    int seg_cur = 0;                   // stage forward_chunk is in
    int dwpw_max_nt = 5;               // output-channel tiles (32 columns each) per k_dwpw workgroup: fewer = more workgroups, the depthwise part redone per group (SVC_DWPW_NT)
    int dwpw_min_px = 400;             // ... on levels with at least this many pixels per frame (SVC_DWPW_MIN_PX); svc_create sets 100 with the split-bf16 pipe (the fused kernel then wins on the 8x13 level too: -18 us per shared pass), 400 is the fp32 pipe's optimum
    int dw_tile = 42;                  // stride-1 depthwise: outputs per thread as TX*10+TY (SVC_DW_TILE: 21, 22, 41, 42, 44; 0 = one output per thread)
    int prim_pt = 2;                   // legacy Prim (k_prim_pt): smallest points-per-thread variant (SVC_PRIM_PT: 2, 4, 8, 16)
    int tail_merge = 1;                // a round's kernels as two fused launches, k_tail_front / k_tail_back (SVC_TAIL_MERGE=0: one launch per stage, for per-kernel profiles)
    int tree_par = 1;                  // data-parallel hierarchy k_tree_par for maps of up to 4352 points (SVC_TREE_PAR=0: the serial builder k_tree)
    int prim_lvl = 1;                  // level-bucketed Prim k_prim_lvl for maps of up to 8192 points (SVC_PRIM_LVL=0: one node per step)
    // TransNet V1 (svc_shot.hip)
    DevBuf shot_blob, shot_ws, shot_w3;   // shot_w3: split-bf16 copies of the cells' weights (svc_shot.hip), valid for shot_w3_mx
    int shot_w3_mx = 0;
    int shot_mx = -1;                  // TransNet cells' matrix pipe: -1 = follow mx, 0 = fp32, 6 = bf16x6, 3 = bf16x3 (SVC_SHOT_MX)
    int shot_m16 = 3;                  // 16-position tiles per wavefront of the split-bf16 cell kernel k_shot_conv_x3m (SVC_SHOT_M16: 2, 3, 4; one more where a wave holds one filter tile)
    int shot_xcd = 1;                  // XCD-aware tile order of k_shot_conv_x3m / k_shot_first_x3 (SVC_SHOT_XCD)
    bool shot_loaded = false;
    // per-kernel-class event log (svc_profile_*)
    int prof_class = -1;
    hipStream_t prof_cal_stream = nullptr;      // a stream of the handle's own for the empty-event-pair calibration of svc_profile_read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
};

// Records an event pair around the launches in its scope when the class is being profiled.
struct ProfScope {
    SvcHandle *h; hipStream_t s; bool on; hipEvent_t a, b;
    ProfScope(SvcHandle *h_, int cls, hipStream_t s_) : h(h_), s(s_), on(h_->prof_class == cls) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, s); h->prof_events.emplace_back(a, b); } }
};

int svc_net_release(SvcHandle *h);
