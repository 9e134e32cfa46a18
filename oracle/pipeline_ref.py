"""Oracle: smart_vid_crop() end to end on the CPU through the reference's pickle door.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, with the other oracle
modules, what smartVidCrop.py does for a video handed over as decoded RGB frames
(ingest_pickle, smartVidCrop.py:560-836) with default-style parameters
(sc_init_crop_params, :132-209), then smart_vid_crop (:2218-2614): destination size,
threshold, cluster loop with cut blend, centres, empty-centre fill, interpolation,
smoothing, boxes.  Parameter-gated branches that both published parameter sets leave
off (border detection, mean-saliency / coverage gates, rendering) are not restated.
The best-settings branch (resize_factor=4 around the cluster filter and the centre, sum-weighted
cluster choice, focus stability, Savitzky-Golay instead of LOESS) is restated too.

Quirks reproduced on purpose: the off-by-one that leaves the last
selected frame of every read batch with an all-zero map (:696-709), the batch-local
after-cut test (:683), the u8 wrap in the cut blend (:2371-2373).
"""
import numpy as np

from . import cv_ref, tail_ref, temporal_ref, unisal_ref


def init_crop_params(use_best_settings=False):
    """Values of smartVidCrop.py:132-209."""
    cp = dict(out_ratio='4:5', max_input_d=250, skip=6, read_batch=2000, resize_factor=1.0,
              resize_type=1, op_close=True, value_bias=1.0, exit_on_spread_sal=False,
              exit_on_low_cvrg=False, com_km=True, clust_filt=True, select_sum=2, min_d_jump=10,
              focus_stability=False, foces_stab_t=60, foces_stab_s=1.5, hdbscan_min=26,
              hdbscan_min_samples=None, shift_time=0, loess_filt=1, loess_w_secs=2, loess_degree=2,
              lp_filt=1, lp_cutoff=2, lp_order=5, t_sal=40, t_cvrg=0.60, t_threshold=120,
              t_border=-1, t_cut=120)
    if use_best_settings:
        cp.update(t_threshold=90, hdbscan_min=5, hdbscan_min_samples=3, min_d_jump=1, resize_factor=4,
                  op_close=True, value_bias=1.0, select_sum=1, focus_stability=True, foces_stab_t=60,
                  foces_stab_s=1.5, t_border=-1, lp_filt=1, lp_cutoff=1, lp_order=2, loess_filt=0)
    return cp


def sal_size(w, h, max_input_d):
    dsr = float(max(w, h)) / max_input_d
    return int(h / dsr), int(w / dsr)            # (SAL_H, SAL_W), smartVidCrop.py:580-582


def select_frames(n_frames, frame_count, trans_inds, skip, read_batch):
    """Frame selection of ingest_pickle (:621-718).
    -> true_inds, map2orig, batches[(first_sel, n_sel)] per read batch."""
    true_inds, map2orig, batches = [], [], []
    total = -1
    after_cut = False
    for b0 in range(0, n_frames, read_batch):
        blen = min(read_batch, n_frames - b0)
        first = len(true_inds)
        for i in range(blen):
            g = b0 + i
            forced = (g == true_inds[-1] + skip) if true_inds else True
            if forced or after_cut or g == frame_count - 1:
                total += 1
                true_inds.append(g)
            if after_cut:
                after_cut = False
            if (i - 1) in trans_inds:            # batch-local index, as in the reference
                after_cut = True
            map2orig.append(total)
        batches.append((first, len(true_inds) - first))
    return true_inds, map2orig, batches


def scenes_from_trans_inds(trans_inds, frame_count):
    scenes = []
    for i in range(len(trans_inds)):
        if frame_count - trans_inds[i] < 2:
            break
        if i + 1 < len(trans_inds):
            scenes.append([trans_inds[i], trans_inds[i + 1] - 1])
    return np.array(scenes, dtype=np.int32)


def ingest(video, CP, sd, stage=None):
    """video: dict(fr, frame_count, w, h, frames[RGB u8], trans_inds) -> VD dict."""
    fr, frame_count, w, h = video['fr'], video['frame_count'], video['w'], video['h']
    frames_full = video['frames']
    sal_h, sal_w = sal_size(w, h, CP['max_input_d'])
    true_inds, map2orig, batches = select_frames(len(frames_full), frame_count, list(video['trans_inds']),
                                                 CP['skip'], CP['read_batch'])
    n_sel = len(true_inds)
    smaps = np.zeros((sal_h, sal_w, n_sel), np.uint8)
    for first, cnt in batches:
        if cnt > 1:                                # the last selected frame of a batch is never sent
            idx = true_inds[first:first + cnt - 1]
            small = np.stack([cv_ref.resize_linear_u8(frames_full[g], sal_h, sal_w) for g in idx])
            if stage is not None:
                stage.setdefault('sal_frames', []).append(small)
            smaps[:, :, first:first + cnt - 1] = unisal_ref.saliency_u8(sd, small)
    seg = scenes_from_trans_inds(list(video['trans_inds']), frame_count)
    seg_sel = np.array([[map2orig[v] for v in row] for row in seg], dtype=np.int32)
    return dict(smaps=smaps, segmentation=seg, segmentation_sel=seg_sel, true_inds=true_inds,
                inds_to_orig=map2orig, fr=fr, fc=len(frames_full), fc_sel=n_sel, h_orig=h, w_orig=w,
                h_process=sal_h, w_process=sal_w)


def select_frames_video(n_frames, frame_count, trans_probs, trans_threshold, skip, read_batch):
    """Frame selection of read_and_segment_video (smartVidCrop.py:379-399): like select_frames, but the after-cut test
    is the TRANSITION PROBABILITY of the previous frame (global index), not membership in a scene list."""
    true_inds, map2orig, batches = [], [], []
    total, after_cut = -1, False
    for b0 in range(0, n_frames, read_batch):
        first = len(true_inds)
        for i in range(min(read_batch, n_frames - b0)):
            g = b0 + i
            forced = (g == true_inds[-1] + skip) if true_inds else True
            if forced or after_cut or g == frame_count - 1:
                total += 1
                true_inds.append(g)
            after_cut = bool(trans_probs[g] > trans_threshold)      # (:394-396)
            map2orig.append(total)
        batches.append((first, len(true_inds) - first))
    return true_inds, map2orig, batches


def scenes_from_probs(trans_probs, trans_threshold):
    """predictions_to_scenes (smartVidCrop.py:214-230, transnet_utils.py:5-19) followed by the "shot segmentation FIX"
    of the video path (:452-456): every scene ends where the next one starts, the last one on the last frame."""
    pred = (np.asarray(trans_probs) > trans_threshold).astype(np.uint8)
    scenes, t, tp, start, i = [], -1, 0, 0, 0
    for i, t in enumerate(pred):
        if tp == 1 and t == 0:
            start = i
        if tp == 0 and t == 1 and i != 0:
            scenes.append([start, i])
        tp = t
    if t == 0:
        scenes.append([start, i])
    if not scenes:
        scenes = [[0, len(pred) - 1]]
    seg = np.array(scenes, dtype=np.int32)
    for k in range(len(seg) - 1):
        seg[k][1] = seg[k + 1][0] - 1
    seg[-1][1] = len(pred) - 1
    return seg


def ingest_video(video, CP, sd, trans_probs, trans_threshold=0.1, stage=None):
    """read_and_segment_video (smartVidCrop.py:234-556) for decoded frames and given transition probabilities (the
    shot network's output, one per frame): selection, saliency with the off-by-one, segmentation."""
    fr, frame_count, w, h = video['fr'], video['frame_count'], video['w'], video['h']
    frames_full = video['frames']
    sal_h, sal_w = sal_size(w, h, CP['max_input_d'])
    true_inds, map2orig, batches = select_frames_video(len(frames_full), frame_count, trans_probs, trans_threshold,
                                                       CP['skip'], CP['read_batch'])
    n_sel = len(true_inds)
    smaps = np.zeros((sal_h, sal_w, n_sel), np.uint8)
    for first, cnt in batches:
        if cnt > 1:
            idx = true_inds[first:first + cnt - 1]
            small = np.stack([cv_ref.resize_linear_u8(frames_full[g], sal_h, sal_w) for g in idx])
            smaps[:, :, first:first + cnt - 1] = unisal_ref.saliency_u8(sd, small)
    seg = scenes_from_probs(trans_probs, trans_threshold)
    seg_sel = np.array([[map2orig[v] for v in row] for row in seg], dtype=np.int32)
    return dict(smaps=smaps, segmentation=seg, segmentation_sel=seg_sel, true_inds=true_inds,
                inds_to_orig=map2orig, fr=fr, fc=len(frames_full), fc_sel=n_sel, h_orig=h, w_orig=w,
                h_process=sal_h, w_process=sal_w)


def crop_from_maps(VD, CP, stage=None):
    """smart_vid_crop after ingest (:2293-2522).  Adds 'bbs' etc. to VD."""
    VD['w_final'], VD['h_final'], VD['conversion_mode'] = tail_ref.calc_dest_size(
        VD['w_orig'], VD['h_orig'], CP['out_ratio'])
    cuts = tail_ref.segm_cuts_of(VD['segmentation_sel'])
    tail_ref.threshold(VD['smaps'], CP['t_threshold'])
    if stage is not None:
        stage['thresholded'] = VD['smaps'].copy()
    if CP['clust_filt']:
        tail_ref.cluster_loop(VD['smaps'], cuts, CP)
    if stage is not None:
        stage['filtered'] = VD['smaps'].copy()
    dx, dy = tail_ref.centers(VD['smaps'], CP)
    if stage is not None:
        stage['centres_raw'] = (list(dx), list(dy))
    VD['dx'], VD['dy'] = temporal_ref.handle_empty_centers(dx, dy, VD['segmentation_sel'])
    if CP['focus_stability']:
        VD['dx'], VD['dy'], VD['jumps'], VD['jumps_inds'] = temporal_ref.focus_stability(
            VD['dx'], VD['dy'], VD['smaps'], VD['fr'], CP)
    VD['dxi'], VD['dyi'] = temporal_ref.interpolate_centres(VD['dx'], VD['dy'], VD['segmentation'],
                                                            VD['segmentation_sel'], VD['true_inds'])
    VD['dxs'], VD['dys'] = temporal_ref.smoothing(VD['dxi'], VD['dyi'], VD['segmentation'], VD['fr'], CP)
    VD['bbs'], VD['fbb_w'], VD['fbb_h'] = tail_ref.compute_bb(
        VD['dxs'], VD['dys'], VD['fc'], VD['w_orig'], VD['h_orig'], VD['w_process'], VD['h_process'],
        VD['w_final'], VD['h_final'])
    if CP['shift_time'] > 0:
        temporal_ref.shift_time(VD['bbs'], CP['shift_time'])
    return VD


def smart_vid_crop(video, CP, sd, stage=None, trans_probs=None, trans_threshold=0.1):
    """trans_probs given: the reference's video path (shot network inside ingest); else the pickle door (trans_inds)."""
    if trans_probs is not None:
        VD = ingest_video(video, CP, sd, trans_probs, trans_threshold, stage)
    else:
        VD = ingest(video, CP, sd, stage)
    return crop_from_maps(VD, CP, stage)
