/*
 * svc.h — C ABI of the MI355X-native SmartVidCrop saliency-to-crop hot path.
 *
 * libsvc_hip.so (retargetvid_amd/csrc) exports exactly these entry points.  Every
 * data pointer is a DEVICE pointer unless its name ends in _host; the caller owns
 * all buffers, the library owns only the weights and scratch workspace inside the
 * handle.  `stream` is a hipStream_t passed as void* (NULL = the default stream);
 * all work is enqueued on it and nothing synchronises unless stated.  Every call
 * returns 0 on success or a negative SVC_E_* code; svc_last_error() then holds a
 * message for the calling thread.  A call with n = 0 frames / maps / boxes is a
 * successful no-op (its buffers may then be NULL).  The library never exits the
 * process and never reads stdin (the reference blocks on input() at smartVidCrop.py:544-545).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   svc_create / svc_destroy    unisal_handler.init_unisal_for_images()
 *                               3rd_party_libs/unisal/unisal_handler.py:68-71 and
 *                               Trainer.model / init_unisal_weights, unisal/train.py:1449-1456, :1200-1209
 *   svc_resize_frames_u8        cv2.resize(frame,(SAL_W,SAL_H),INTER_LINEAR), smartVidCrop.py:333-335, :633-635
 *   svc_saliency_u8             unisal_handler.predictions_from_memory_nuint8_np(),
 *                               3rd_party_libs/unisal/unisal_handler.py:85-86 ->
 *                               Trainer.generate_predictions_from_image_memory_nuint8_np, unisal/train.py:1255-1279
 *   svc_threshold_u8            sc_threshold(), smartVidCrop.py:1050-1059
 *   svc_cluster_center          the clustering loop + centre loop of smart_vid_crop():
 *                               sc_clustering_filt() smartVidCrop.py:1062-1161 with the cut blend :2359-2373,
 *                               and sc_find_center_of_mass() :1163-1219 / :2402-2414
 *   svc_iou_i32                 bb_intersection_over_union(), smartVidCrop.py:927-944
 *                               (== retargetvid_eval.py:10-27)
 */
#ifndef SVC_H_
#define SVC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVC_OK 0
#define SVC_E_INVALID (-1)     /* bad argument / unsupported size           */
#define SVC_E_HIP (-2)         /* a HIP runtime call failed                 */
#define SVC_E_BLOB (-3)        /* malformed weights blob                    */
#define SVC_E_NOMEM (-4)

typedef struct SvcHandle SvcHandle;

/* The crop parameters (sc_init_crop_params, smartVidCrop.py:132-209) that the
 * device path consumes. */
typedef struct SvcParams {
    uint32_t struct_size;         /* = sizeof(SvcParams) as the CALLER compiled it: a binding built against an older
                                     layout is rejected with SVC_E_INVALID instead of being read past its end */
    int32_t hdbscan_min;          /* CP['hdbscan_min']  (min_cluster_size)             */
    int32_t hdbscan_min_samples;  /* CP['hdbscan_min_samples'], 0 = None               */
    int32_t select_sum;           /* CP['select_sum']: 1 = cluster sum, else cluster max */
    int32_t op_close;             /* CP['op_close']                                    */
    int32_t clust_filt;           /* CP['clust_filt']: 0 skips filtering, centres only */
    int32_t resize_factor;        /* CP['resize_factor'] (integer, 1 = off): cluster on the map shrunk by this
                                     factor (INTER_LINEAR down, then up again) and take the centre of the
                                     INTER_NEAREST-shrunk map, smartVidCrop.py:1078-1084, :1158, :1184 */
    int32_t com_km;               /* CP['com_km']: 1 = centre of mass (single-cluster K-means = centroid of the non-zero
                                     pixels), 0 = position of the first maximum of the final map in raster order
                                     (sc_find_center_of_mass with km=False, smartVidCrop.py:1165-1178) */
} SvcParams;

/* Per-call diagnostics written by svc_cluster_center when `stats` is non-NULL:
 * int32[n][4] = { points after threshold/blend, clusters selected, label kept, reserved } */
#define SVC_STATS_STRIDE 4

const char *svc_last_error(void);

/* ABI revision of the loaded library (bumped whenever a struct layout or a signature changes); a binding
 * checks it once after dlopen.  2 = SvcParams starts with struct_size and carries resize_factor.
 * 3 = SvcParams ends with com_km; SVC_MAP_HELD is honoured by svc_cluster_center (a v2 library ignores the bit), svc_debug_cluster_state
 *     returns 32 header words, svc_debug_round_plan / svc_debug_argsort_u32 / svc_transnet_* exist.
 * 4 = the host stages svc_host_* (SvcTemporalParams) and svc_saliency_thresholded_u8 exist. */
#define SVC_ABI_VERSION 5
int svc_abi_version(void);

/* weights_blob_host: the packed, BN-folded static SALICON slice of a UNISAL
 * checkpoint (retargetvid_amd.weights.pack_blob).  The blob is copied to `device`. */
int svc_create(const void *weights_blob_host, size_t n_bytes, int device, SvcHandle **out);
int svc_destroy(SvcHandle *h);

/* frames[n][h][w][3] u8 RGB -> out[n][sh][sw][3] u8, OpenCV INTER_LINEAR semantics. */
int svc_resize_frames_u8(SvcHandle *h, const uint8_t *frames, int n, int height, int width,
                         uint8_t *out, int sh, int sw, void *stream);

/* frames_nhwc[n][h][w][3] u8 RGB (saliency size, e.g. 140x250) -> maps_nhw[n][h][w] u8.
 * Frame-major output; the reference's [h][w][n] view is a transpose done by the
 * caller only when someone asks for VD['smaps']. */
int svc_saliency_u8(SvcHandle *h, const uint8_t *frames_nhwc, int n, int height, int width,
                    uint8_t *maps_nhw, void *stream);

/* svc_saliency_u8 followed by svc_threshold_u8(t) in one pass: the last kernel of the network writes the thresholded map
 * (sc_threshold, smartVidCrop.py:1050-1059, applied to the u8 value as the reference applies it to the stored map).
 * Identical bytes; one launch less per chunk.  t in 0..255 (0 = svc_saliency_u8). */
int svc_saliency_thresholded_u8(SvcHandle *h, const uint8_t *frames_nhwc, int n, int height, int width,
                                uint8_t *maps_nhw, int t, void *stream);

/* svc_saliency_thresholded_u8 (t in 1..255) that also ADDS, per frame, how many pixels of the UN-thresholded map sit at t - 1, t
 * and t + 1 to the caller's DEVICE rows census_n4[n][4] (uint32; column 3 unused; the caller zeroes them): svc_threshold_census
 * per frame, for callers whose passes mix the frames of several videos (retargetvid_amd/scheduler.py reports the regime
 * diagnostic per video from these rows).  census_n4 may be NULL (= svc_saliency_thresholded_u8).  No counterpart in the reference. */
int svc_saliency_census_u8(SvcHandle *h, const uint8_t *frames_nhwc, int n, int height, int width,
                           uint8_t *maps_nhw, int t, uint32_t *census_n4, void *stream);

/* maps[i] = maps[i] < t ? 0 : maps[i], in place. */
int svc_threshold_u8(SvcHandle *h, uint8_t *maps, size_t n_bytes, int t, void *stream);

/* Cluster filter + cut blend + centre of every map, in place on thresholded maps.
 * blend_flags_host[n] (HOST memory, may be NULL), bits per map i:
 *   SVC_BLEND_NEXT  "once map i is final, blend it into map i+1" (smartVidCrop.py:2369-2373; the caller evaluates the
 *                   cut test); map i+1 is then processed after map i;
 *   SVC_MAP_HELD    map i is NOT processed by this call: it is already final (the end of a chain a previous call
 *                   processed, carried over so that the chain can continue: HELD | BLEND_NEXT) or left for a later call.
 *                   A chain that straddles two calls gives the same maps and centres as in one call
 *                   (tests/test_gpu_parity.py::test_blend_chain_carried_over_between_calls).
 * xy[n][2] receives (x, y) centres as float64 in saliency-map pixels, NaN = None (empty map); the entries of held
 * maps are unspecified.  stats may be NULL. */
#define SVC_BLEND_NEXT 1
#define SVC_MAP_HELD 2
int svc_cluster_center(SvcHandle *h, uint8_t *maps_nhw, int n, int height, int width,
                       const uint8_t *blend_flags_host, const SvcParams *params,
                       double *xy, int32_t *stats, void *stream);

/* Test door, host only (no GPU): the rounds svc_cluster_center plans for a flag array.  round_out[i] = round in which
 * map i is processed, -1 = held; blend0_out[i] = 1 when map i first takes a blend from a held predecessor.
 * Returns the number of rounds (0 when every map is held) or a negative error. */
int svc_debug_round_plan(const uint8_t *flags_host, int n, int32_t *round_out, int32_t *blend0_out);

/* a[n][4], b[n][4] int32 boxes (x1,y1,x2,y2) -> out[n] float64 IoU, inclusive +1 pixel convention. */
int svc_iou_i32(const int32_t *a, const int32_t *b, size_t n, double *out, void *stream);

/* ---- host stages between the per-frame centres and the crop windows (csrc/svc_host.cpp) ----
 * north_star keeps them on the host; they are native so that a multi-video job runs them off the interpreter lock.
 * No GPU, no handle, no stream: every pointer is HOST memory, float64 unless stated.  Reference: smartVidCrop.py
 *   svc_host_fill_empty_centres  sc_handle_empty_centers :1221-1300.  NaN = no centre; filled in place; returns the
 *                                number of centres still empty (> 0 only when a run copies from an empty neighbour)
 *   svc_host_interp_segment      interp_handler :1528-1548 for the two series of one shot: n samples at the increasing times
 *                                sampled_t -> n_out values at times 0 .. n_out-1 (constant for n < 3, interp1d 'linear' for
 *                                n <= 6, 'quadratic' beyond, both extrapolating)
 *   svc_host_lowpass             sc_butter_lowpass_filter :1599-1627: scipy.signal.filtfilt(b, a, x) with b, a =
 *                                scipy.signal.butter(order, cutoff / (fr / 2)) and zi = scipy.signal.lfilter_zi(b, a)
 *                                (taps = order + 1 coefficients; zi has taps - 1), or, for n <= 3 * taps, the reference's
 *                                5-point moving average over x[2 : n-2]
 *   svc_host_loess               pyloess.Loess(arange(n), y).estimate(j, window, degree) for every j
 *                                (3rd_party_libs/loess/pyloess.py:13-95); NaN where the reference divides 0 / 0
 *   svc_host_savgol              scipy.signal.savgol_filter(y, window, degree), mode 'interp' (:1643)
 *   svc_host_temporal            sc_interpolate + sc_smoothing (:1550-1597, :1648-1734) of one video: centres of the n_sel
 *                                selected frames (true_inds = their frame numbers) + the shots seg[n_seg][2] (first, last
 *                                decoded frame) / seg_sel[n_seg][2] (first, last selected frame) -> interpolated (xi, yi)
 *                                and smoothed (xs, ys) centres, fc values each.  Returns the number of values produced.
 *   svc_host_boxes               sc_compute_bb :979-1048: smoothed centres (saliency-map pixels) -> boxes[fc][4] int64
 *                                (x1, y1, x2, y2), centres[fc][2] (may be NULL) = the truncated full-resolution centres the
 *                                reference writes back to dxs / dys, fbb_wh[2] (may be NULL) = box width, height;
 *                                borders_tblr (may be NULL = 0) = border_t, border_b, border_l, border_r
 *   svc_host_focus_stability     the jump statistics and the focus hold of the ISM 2021 parameter set (get_points_on_line /
 *                                sc_check_for_extra_cuts :1337-1455, the hold loop :2425-2473): centres of the n selected frames (in
 *                                place), the FILTERED maps frame-major uint8 [n][h][w] (host) -> jumps[n] (mean map value along the
 *                                move from centre i-1 to centre i, 255 = none), inds[] = frames with a jump below stab_t (returned
 *                                count); float32 buffer / slope arithmetic as in the reference */
typedef struct SvcTemporalParams {
    uint32_t struct_size;       /* = sizeof(SvcTemporalParams) as the caller compiled it */
    int32_t lp_filt;            /* CP['lp_filt'] */
    int32_t lp_taps;            /* CP['lp_order'] + 1 = length of lp_b / lp_a */
    int32_t loess_filt;         /* CP['loess_filt']: 1 = LOESS, 0 = Savitzky-Golay */
    int32_t loess_degree;       /* CP['loess_degree'] */
    int32_t reserved;
    double loess_w_secs;        /* CP['loess_w_secs'] */
    double fr;                  /* frames per second of the video */
} SvcTemporalParams;
int svc_host_focus_stability(double *cx, double *cy, int n, const uint8_t *maps_nhw, int h, int w, double fr, int skip,
                             double min_d_jump, double stab_t, double stab_s, double *jumps, int32_t *inds);
int svc_host_fill_empty_centres(double *cx, double *cy, int n_sel, const int32_t *seg_sel, int n_seg);
int svc_host_interp_segment(const double *sampled_t, const double *d1, const double *d2, int n, int n_out, double *out1, double *out2);
int svc_host_lowpass(const double *b, const double *a, const double *zi, int taps, const double *x, int n, double *out);
int svc_host_loess(const double *y, int n, int window, int degree, double *out);
int svc_host_savgol(const double *y, int n, int window, int degree, double *out);
int svc_host_temporal(const SvcTemporalParams *p, const double *lp_b, const double *lp_a, const double *lp_zi,
                      const double *cx, const double *cy, int n_sel, const int32_t *true_inds,
                      const int32_t *seg, const int32_t *seg_sel, int n_seg, int fc,
                      double *xi, double *yi, double *xs, double *ys);
int svc_host_boxes(const double *xs, const double *ys, int fc, int w_orig, int h_orig, int w_process, int h_process,
                   int w_final, int h_final, const int32_t *borders_tblr, int64_t *boxes, int64_t *centres, int32_t *fbb_wh);

/* Measurement door (bench.py): record HIP events around every launch of one kernel class on
 * the stream it is launched on.  kernel_class = one of SVC_K_*, or -1 to switch recording off.
 * svc_profile_read synchronises the device, returns the summed duration (ms; 3/4 of the cost of an empty event pair on
 * the same stream, measured at read time, is taken off every launch: calibrated against rocprofv3 kernel durations) and the number of launches recorded since the last
 * read, and resets the log.  Nothing like it exists in the
 * reference (its timers are host wall-clock accumulators, smartVidCrop.py:98-127). */
#define SVC_K_RESIZE 0
#define SVC_K_LANCZOS 1
#define SVC_K_STEM 2
#define SVC_K_PW 3
#define SVC_K_DW 4
#define SVC_K_RESAMPLE 5   /* subsample / upsample / gauss fill / adapt */
#define SVC_K_SMOOTH 6     /* k_smooth_down + k_quantise */
#define SVC_K_THRESHOLD 7
#define SVC_K_COMPACT 8
#define SVC_K_CORE 9
#define SVC_K_PRIM 10
#define SVC_K_FINISH 11
#define SVC_K_COUNT 12
int svc_profile_enable(SvcHandle *h, int kernel_class);
int svc_profile_read(SvcHandle *h, double *total_ms, int *launches);
/* The same log without the correction: raw_total_ms = sum of the event-pair durations, pair_ms = cost of an empty event pair
 * (median of 15, measured on a stream the handle owns); svc_profile_read returns raw - 0.75 * pair * launches. */
int svc_profile_read_raw(SvcHandle *h, double *raw_total_ms, double *pair_ms, int *launches);

/* Test/diagnostic door: copy internal per-frame clustering state of the LAST
 * svc_cluster_center call to HOST buffers (any may be NULL).  Synchronises.
 *   pts_host[cap]   packed points: row | col<<8 | value<<16, raster order
 *   core_host[cap]  core distances (squared)
 *   mst_host[cap][3] MST edges in Prim order: from, to, weight
 *   labels_host[cap] final labels (-1 = noise)
 *   hdr_host[32]    frame header: [0] points, [1] clusters selected, [2] cluster kept, [3] clustered,
 *                   [4] condensed clusters, [8..11] k_finish phase stamps in 10 ns units
 *                   (sorted, hierarchy built, cluster chosen, done), [12] Prim duration,
 *                   [16] rounds and [17] level rises of k_prim_lvl, [18..22] its phase sums (10 ns units)
 * Returns the number of points of that frame (or a negative error). */
int svc_debug_cluster_state(SvcHandle *h, int frame, int cap, uint32_t *pts_host, uint32_t *core_host,
                            uint32_t *mst_host, int32_t *labels_host, int32_t *hdr_host);

/* Test/diagnostic door: the edge-order routine of the cluster filter (numpy's default argsort, an unstable
 * introsort, emulated on the device; hdbscan sorts the MST edges with it, call site smartVidCrop.py:1099) on
 * arbitrary keys: order_host[p] = index of the key that the sort puts at position p.  n <= 65535.  Synchronises.
 * Returns n (or a negative error). */
int svc_debug_argsort_u32(SvcHandle *h, const uint32_t *keys_host, int n, int32_t *order_host);

/* ---- TransNet V1 shot-boundary network ----
 * Replaces ShotTransNet._restore / predict_raw (3rd_party_libs/transnetv1/transnetv1_handler.py:86-97; the reference
 * calls it through predict_frames at smartVidCrop.py:369).
 * svc_transnet_load: the F16 L3 S2 D256 weights as ONE float32 array in the layout of
 *   retargetvid_amd/weights.pack_transnet_blob (per DDCNN cell four [rows][27 taps x channels] GEMM matrices + biases,
 *   Dense(256) and Dense(2) transposed); n_floats is checked.
 * svc_transnet_predict: frames = DEVICE uint8 [n_windows][frames_per_window][27][48][3] (RGB, already 48x27),
 *   probs = DEVICE float32 [n_windows][frames_per_window] = P(transition) of every frame (softmax class 1).
 *   Windowing of a video (100-frame windows, stride 50, edge padding) is host logic: transnetv1_handler.predict_video. */
int svc_transnet_load(SvcHandle *h, const float *blob_host, size_t n_floats);
int svc_transnet_predict(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, float *probs, void *stream);
/* svc_transnet_predict_rows: the same, but the caller keeps only rows row0 .. row1 - 1 of every window (the reference's
 * predict_video keeps rows 25 .. 74 of its 100-frame windows, transnetv1_handler.py:117-121).  Those rows of probs are bit for bit
 * svc_transnet_predict's; the others are UNSPECIFIED (left as they were on the split-bf16 pipes, computed on the fp32 pipe).
 * A cell's temporal reach is 8 frames (its largest dilation), so on the split-bf16 pipes every layer is computed only on the frames
 * the kept rows depend on: 50 / 66 / 82 / 98 of 100 frames in the last four cells = a fifth of the network's FLOPs less.
 * 0 <= row0 < row1 <= frames_per_window. */
int svc_transnet_predict_rows(SvcHandle *h, const uint8_t *frames, int n_windows, int frames_per_window, int row0, int row1,
                              float *probs, void *stream);
/* The handle's TransNet knobs as three int32: {matrix pipe: -1 = the handle's SVC_MX | 0 fp32 | 6 bf16x6 | 3 bf16x3, 16-position
 * tiles per wavefront of the split-bf16 cell kernel (2..4), XCD-aware tile order (0 | 1)} -- read from SVC_SHOT_MX / SVC_SHOT_M16 /
 * SVC_SHOT_XCD when the handle is created; ShotTransNet.clone() copies them to the engine of the copy with these two calls instead of
 * going through the process environment.  _set rejects values that are not a configuration with SVC_E_INVALID. */
int svc_transnet_config_get(const SvcHandle *h, int32_t *cfg3);
int svc_transnet_config_set(SvcHandle *h, const int32_t *cfg3);

/* Test/diagnostic door: copy an intermediate activation of the LAST svc_saliency_u8
 * call (NHWC fp32, frame 0..n-1) to a HOST buffer.  `which` is one of the SVC_TAP_*
 * ids.  Returns the number of floats per frame (or a negative error).  Synchronises. */
#define SVC_TAP_INPUT 0     /* [256][416][3]  normalised network input          */
#define SVC_TAP_FEAT4X 1    /* [32][52][64]   cnn.features.7 output (pre-subsample) */
#define SVC_TAP_FEAT2X 2    /* [16][26][160]  cnn.features.14 output            */
#define SVC_TAP_FEAT1X 3    /* [8][13][1296]  cnn.features.18 output + 16 Gaussian maps */
#define SVC_TAP_POSTCNN 4   /* [8][13][256]                                     */
#define SVC_TAP_DEC 5       /* [32][52][64]   post_upsampling_2 output          */
#define SVC_TAP_PRE 6       /* [h][w]         pre-softmax map at saliency size  */
int svc_debug_tap(SvcHandle *h, int which, int frame, float *out_host, size_t cap_floats);
/* SVC_TAP_INPUT with the fused front kernel (the default): the input is written to memory only by handles created
 * with SVC_KEEP_INPUT=1 in the environment; otherwise the call fails with SVC_E_INVALID.
 * svc_front_fused: 1 when the last svc_saliency_u8 call ran LANCZOS + features.0 + features.1 as one kernel
 * (k_front), 0 when as three (SVC_FRONT=0, or a source size whose tiles do not fit in LDS).  bench.py uses it to
 * attribute features.1's FLOPs to the right kernel class. */
int svc_front_fused(const SvcHandle *h);
/* svc_matrix_pipe: which matrix pipe the handle's 1x1-convolution GEMMs (the `pw` kernel class) run on: 0 = fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: exact f32 products, rounds 1-4), 6 = split-bf16 operands on v_mfma_f32_32x32x16_bf16 (every f32
 * operand as three bf16 planes = its 24 significant bits, six plane pairs per product, f32 accumulation: csrc/svc_net.hip,
 * "Split-bf16 operands").  Chosen when the handle is created (environment SVC_MX=f32 | bf16x6; default bf16x6: ~10 %
 * faster, every parity gate of the fp32 pipe unchanged, bit-reproducible with several streams sharing the chip -- the one
 * kernel that was not, the smoothing kernel, computes its bilinear stage with scalar instructions since: DESIGN.md 5); bench.py
 * reports it as roofline.matrix_pipe.  No interface of the reference corresponds to it (its arithmetic type is f32 either way). */
int svc_matrix_pipe(const SvcHandle *h);
/* svc_transnet_matrix_pipe: the same for the TransNet cells with >= 64 input channels (svc_transnet_predict): 0 = fp32 MFMA,
 * 6 = split-bf16 operands, six plane pairs (fp32-class results: |dP| against the oracle 6e-7, the fp32 pipe's 6e-7), 3 = three plane
 * pairs (hi.hi + hi.mid + mid.hi: 16 significant bits per product, |dP| 1.4e-5; opt-in).  Environment SVC_SHOT_MX=f32 | bf16x6 |
 * bf16x3 when the handle is created; default = the handle's SVC_MX.  svc_create rejects any other spelling of SVC_MX / SVC_SHOT_MX
 * with SVC_E_INVALID (ABI 5: a typo used to select the fp32 pipe silently). */
int svc_transnet_matrix_pipe(const SvcHandle *h);
/* svc_threshold_census: the regime diagnostic of the threshold (no counterpart in the reference; smartVidCrop.py:1050-1059 only
 * thresholds).  out[0] = maps that went through svc_saliency_thresholded_u8 on this handle since the last reset, out[1..3] = how
 * many pixels of their UN-thresholded u8 maps sat at t - 1, t and t + 1 (t = the threshold passed to the call).  Two correct
 * fp32 implementations of the network differ by one grey level on ~0.3 % of the pixels; how many pixels that moves across
 * the threshold -- and with them points into or out of the clustering -- is this density.  Measured boundary:
 * ~7 pixels per level and map (trained-like checkpoints) or ~45 (the carrier checkpoint): crop windows identical to the
 * oracle's; ~500 (reference-initialised weights): 21 % of the windows differ by more than a pixel.  Synchronises the device.
 * reset != 0 clears the counters. */
int svc_threshold_census(SvcHandle *h, unsigned long long *out /* [4] */, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SVC_H_ */
