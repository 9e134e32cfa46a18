"""Drop-in counterpart of the reference's ``smartVidCrop.py`` for the saliency-to-crop path.

Same entry points, argument meaning, return values and result-file format:

  sc_init_crop_params(print_dict=False, use_best_settings=False)      smartVidCrop.py:132-209
  smart_vid_crop(video_path, CP=None, ...) -> (VD, smart_crop_results)  smartVidCrop.py:2218-2614
  smartVidCrop = smart_vid_crop            (the spelling used in BASELINE.json)
  smart_crop_version() -> '1.4.0'          smartVidCrop.py:2617-2618
  bb_intersection_over_union(boxA, boxB)   smartVidCrop.py:927-944
  write_results(...)                       result / info files, smartVidCrop.py:2777-2785

What runs where: frame down-scale, UNISAL saliency, threshold, cluster filter, cut
blend, centre of focus and IoU run on the MI355X through the C ABI in include/svc.h
(retargetvid_amd.ops); frame selection, shot bookkeeping, empty-centre fill,
interpolation, low-pass, LOESS and box arithmetic stay on the host
(retargetvid_amd.temporal), as BASELINE.json's north_star prescribes.  There is no CPU
fallback for the device stages: without a GPU and the built HIP library every call raises.

Input door: the reference decodes video files with OpenCV/imutils and detects shots with
TransNetV1; both stay outside this package.  ``video_path`` is therefore either a
``.pkl`` written in the reference's ingest_pickle format (smartVidCrop.py:560-573: dict
with fr, frame_count, w, h, frames [RGB uint8], trans_inds) or that dict itself.
Failures raise exceptions; nothing blocks on input() (the reference does at :544-545).
"""
import math
import os
import pickle
import threading
import time

import numpy as np

from . import temporal

_ENGINE = None            # module-level singleton like the reference's unisal_model (:77)


def get_engine(state_dict=None, seed=0):
    """The process-wide device engine (weights + workspace).  Built on first use."""
    global _ENGINE
    if _ENGINE is None or state_dict is not None:
        from . import ops
        _ENGINE = ops.Engine(state_dict, seed=seed)
    return _ENGINE


def set_engine(engine):
    global _ENGINE
    _ENGINE = engine


# ---- time registry (smartVidCrop.py:98-127) ------------------------------------------------
_tls = threading.local()       # per-thread stage timers (the reference keeps one global dict, smartVidCrop.py:98)


_video_reader = None


def set_video_reader(fn):
    """Install the decode hand-off: ``fn(video_path, crop_params) -> ingest_pickle dict`` (see
    retargetvid_amd/ingest.py).  None removes it.  Decoding and shot detection are not part of the path."""
    global _video_reader
    _video_reader = fn


def sc_init_time():
    _tls.times = {}


def sc_register_time(t, key_name):
    times = _tls.__dict__.setdefault('times', {})
    times[key_name] = times.get(key_name, 0.0) + (time.perf_counter() - t)


def sc_all_times(vid_dur):
    t_dict, sum_t, sum_p = {}, 0.0, 0.0
    for k, v in _tls.__dict__.setdefault('times', {}).items():
        if k.startswith('_'):
            sum_t += v
            sum_p += (v / vid_dur) * 100.0
        t_dict[k] = '%7.3fs, %6.3f%%' % (v, (v / vid_dur) * 100.0)
    t_dict['total'] = '%7.3fs, %6.3f%%' % (sum_t, sum_p)
    return t_dict


# ---- parameters (smartVidCrop.py:132-209) ---------------------------------------------------
def sc_init_crop_params(print_dict=False, use_best_settings=False):
    crop_params = {
        'out_ratio': '4:5', 'max_input_d': 250, 'skip': 6, 'read_batch': 2000,
        'resize_factor': 1.0, 'resize_type': 1, 'op_close': True, 'value_bias': 1.0,
        'exit_on_spread_sal': False, 'exit_on_low_cvrg': False, 'com_km': True, 'clust_filt': True,
        'select_sum': 2, 'min_d_jump': 10, 'focus_stability': False, 'foces_stab_t': 60,
        'foces_stab_s': 1.5, 'hdbscan_min': 26, 'hdbscan_min_samples': None, 'shift_time': 0,
        'loess_filt': 1, 'loess_w_secs': 2, 'loess_degree': 2, 'lp_filt': 1, 'lp_cutoff': 2,
        'lp_order': 5, 't_sal': 40, 't_cvrg': 0.60, 't_threshold': 120, 't_border': -1, 't_cut': 120,
    }
    if use_best_settings:
        crop_params.update({
            't_threshold': 90, 'hdbscan_min': 5, 'hdbscan_min_samples': 3, 'min_d_jump': 1,
            'resize_factor': 4, 'op_close': True, 'value_bias': 1.0, 'select_sum': 1,
            'focus_stability': True, 'foces_stab_t': 60, 'foces_stab_s': 1.5, 't_border': -1,
            'lp_filt': 1, 'lp_cutoff': 1, 'lp_order': 2, 'loess_filt': 0})
    if print_dict:
        for k, v in crop_params.items():
            print(k, ':', v)
    return crop_params


def smart_crop_version():
    return '1.4.0'


def bb_intersection_over_union(boxA, boxB):
    """IoU of two [x1,y1,x2,y2] boxes (inclusive +1 convention) — evaluated by the svc_iou_i32
    kernel; use retargetvid_amd.ops.iou_boxes for batches."""
    from . import ops
    return float(ops.iou_boxes(np.asarray([boxA], np.int32), np.asarray([boxB], np.int32))[0])


# ---- geometry (host) ---------------------------------------------------------------------
def sc_calc_dest_size(vid_data, crop_params, verbose=False):
    """smartVidCrop.py:946-977."""
    a, b = (float(v) for v in crop_params['out_ratio'].split(':'))
    w, h = vid_data['w_orig'], vid_data['h_orig']
    if abs(float(w) / float(h) - a / b) < 0.0000001:
        vid_data['conversion_mode'], vid_data['w_final'], vid_data['h_final'] = 0, w, h
    else:
        vid_data['w_final'], vid_data['h_final'] = int(math.floor((a / b) * h)), h
        vid_data['conversion_mode'] = 1
        if vid_data['w_final'] > w or vid_data['h_final'] > h:
            vid_data['w_final'], vid_data['h_final'] = w, int(math.floor((b / a) * w))
            vid_data['conversion_mode'] = 2
    if verbose:
        print(' orig. (hxw): (%dx%d)' % (h, w))
        print(' final (hxw): (%dx%d)' % (vid_data['h_final'], vid_data['w_final']))
    return vid_data


def sc_compute_bb(vid_data, crop_params, verbose=False):
    """smartVidCrop.py:979-1048: centres (saliency-map pixels) -> [x1,y1,x2,y2] per frame (native: svc_host_boxes)."""
    fc = vid_data['fc']
    borders = tuple(vid_data.get(k, 0) for k in ('border_t', 'border_b', 'border_l', 'border_r'))
    boxes, ctr, fbb_w, fbb_h = temporal.boxes(vid_data['dxs'][:fc], vid_data['dys'][:fc], vid_data['w_orig'], vid_data['h_orig'],
                                              vid_data['w_process'], vid_data['h_process'], vid_data['w_final'],
                                              vid_data['h_final'], borders)
    # full-resolution integer centres, int() = truncation toward zero (:995-999)
    vid_data['dxs'][:fc] = ctr[:, 0].tolist()
    vid_data['dys'][:fc] = ctr[:, 1].tolist()
    vid_data['fbb_w'], vid_data['fbb_h'] = fbb_w, fbb_h
    vid_data['bbs_np'] = boxes                         # int64 [fc, 4]: what the multi-video job gathers (dist.crop_job)
    if isinstance(vid_data, _LazySmaps):
        dict.pop(vid_data, 'bbs', None)                # the list-of-lists form (the reference's VD['bbs']) is made on first access
    else:
        vid_data['bbs'] = boxes.tolist()
    return vid_data


# ---- ingest (host bookkeeping + device saliency) ---------------------------------------------
def _select_frames(n_frames, frame_count, trans_inds, skip, read_batch):
    """Frame selection of ingest_pickle (smartVidCrop.py:621-718), including the
    batch-local after-cut test (:683).  -> true_inds, map2orig, [(first, count)] per read batch."""
    true_inds, map2orig, batches = [], [], []
    total, after_cut = -1, False
    trans = set(int(t) for t in trans_inds)
    for b0 in range(0, n_frames, read_batch):
        first = len(true_inds)
        for i in range(min(read_batch, n_frames - b0)):
            g = b0 + i
            if (g == true_inds[-1] + skip if true_inds else True) or after_cut or g == frame_count - 1:
                total += 1
                true_inds.append(g)
            after_cut = (i - 1) in trans
            map2orig.append(total)
        batches.append((first, len(true_inds) - first))
    return true_inds, map2orig, batches


def _select_frames_video(n_frames, frame_count, trans_probs, trans_threshold, skip, read_batch):
    """Frame selection of the reference's video path (read_and_segment_video, smartVidCrop.py:379-399): the after-cut
    test is the shot network's transition probability of the previous frame.  Same return values as _select_frames."""
    true_inds, map2orig, batches = [], [], []
    total, after_cut = -1, False
    hot = np.asarray(trans_probs) > trans_threshold
    for b0 in range(0, n_frames, read_batch):
        first = len(true_inds)
        for i in range(min(read_batch, n_frames - b0)):
            g = b0 + i
            if (g == true_inds[-1] + skip if true_inds else True) or after_cut or g == frame_count - 1:
                total += 1
                true_inds.append(g)
            after_cut = bool(hot[g])
            map2orig.append(total)
        batches.append((first, len(true_inds) - first))
    return true_inds, map2orig, batches


TRANS_THRESHOLD = 0.1          # smartVidCrop.py:64

_STAGE_BYTES = 96 << 20        # pinned / device staging buffer size of the host-fed ingest (two of each per engine)


class _HostFeed:
    """Host frames -> saliency-size frames on the device, selection applied BEFORE the copy, with the copies off the
    critical path: two pinned host buffers and two device buffers per engine, H2D on a side stream, the down-scale
    (svc_resize_frames_u8) on the caller's stream.  While chunk c is being copied and down-scaled the host gathers
    chunk c+1 into the other pinned buffer; an event per buffer keeps a pinned slot from being refilled before its
    copy has run and a device slot from being overwritten before its down-scale has read it.  Replaces the reference's
    per-frame cv2.resize on the host inside the read loop (smartVidCrop.py:333-335, :633-635) for inputs that live in
    host memory; the 4K stream of BASELINE config 5 is bound by this copy (24.9 MB per frame over PCIe)."""

    def __init__(self, engine):
        import torch
        self.engine = engine
        self.dev = engine.device
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.shape = None

    def _buffers(self, h, w):
        import torch
        if self.shape != (h, w):
            k = max(1, min(32, _STAGE_BYTES // (h * w * 3)))
            self.pinned = [torch.empty((k, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.staged = [torch.empty((k, h, w, 3), dtype=torch.uint8, device=self.dev) for _ in range(2)]
            self.copied = [torch.cuda.Event(), torch.cuda.Event()]       # H2D of the slot has run
            self.consumed = [torch.cuda.Event(), torch.cuda.Event()]     # the down-scale has read the device slot
            self.used = [False, False]
            self.shape, self.k = (h, w), k
        return self.k

    def downscale(self, frames, idx, sal_h, sal_w):
        """frames: host frames [n,h,w,3] u8 -- a numpy array (pageable memory: gathered into the pinned slots by a few
        threads) or a PINNED torch tensor (copied from where it lies, frame by frame); idx: selected frame numbers.
        -> uint8 CUDA tensor [len(idx), sal_h, sal_w, 3], produced on the caller's current stream."""
        import torch
        h, w = int(frames.shape[1]), int(frames.shape[2])
        k = self._buffers(h, w)
        out = torch.empty((len(idx), sal_h, sal_w, 3), dtype=torch.uint8, device=self.dev)
        compute = torch.cuda.current_stream(self.dev)
        direct = torch.is_tensor(frames) and frames.is_pinned()
        src = frames if direct else (frames.numpy() if torch.is_tensor(frames) else frames)
        for c, s in enumerate(range(0, len(idx), k)):
            part = idx[s:s + k]
            slot = c & 1
            if self.used[slot] and not direct:
                self.copied[slot].synchronize()                 # the pinned slot's previous copy has run
            if not direct:                                      # selection before the copy: only these frames cross PCIe
                host = self.pinned[slot].numpy()
                if len(part) > 1 and h * w * 3 >= (1 << 20):    # big frames: the gather itself is the bottleneck (one thread
                    list(self._pool().map(lambda jf: np.copyto(host[jf[0]], src[jf[1]]), enumerate(part)))   # copies ~10 GB/s)
                else:
                    np.take(src, part, axis=0, out=host[:len(part)], mode='clip')
            with torch.cuda.stream(self.copy_stream):
                if self.used[slot]:
                    self.copy_stream.wait_event(self.consumed[slot])    # the device slot's previous reader is done
                if direct:
                    if len(part) > 1 and part[-1] - part[0] == len(part) - 1 and all(part[j + 1] == part[j] + 1 for j in range(len(part) - 1)):
                        self.staged[slot][:len(part)].copy_(src[part[0]:part[0] + len(part)], non_blocking=True)     # a run of consecutive frames: one copy
                    else:
                        for j, f in enumerate(part):
                            self.staged[slot][j].copy_(src[f], non_blocking=True)
                else:
                    self.staged[slot][:len(part)].copy_(self.pinned[slot][:len(part)], non_blocking=True)
                self.copied[slot].record(self.copy_stream)
            compute.wait_event(self.copied[slot])
            out[s:s + len(part)] = self.engine.resize_frames(self.staged[slot][:len(part)], sal_h, sal_w)
            self.consumed[slot].record(compute)
            self.used[slot] = True
        return out

    def _pool(self):
        if getattr(self, '_tp', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._tp = ThreadPoolExecutor(max_workers=4)
        return self._tp


def device_index(engine, idx, dev=None):
    """Frame numbers -> int64 CUDA tensor WITHOUT a host-device synchronisation: torch.as_tensor(list, device=...) copies from
    pageable memory, i.e. waits for everything the stream holds -- once per read batch that is the end of the host's run-ahead
    (and, with several videos in flight, of their overlap).  An arithmetic progression is generated on the device; any
    other list travels through a small ring of pinned slots (an event per slot: a slot is not refilled before its copy ran)."""
    import torch
    dev = dev or engine.device
    n = len(idx)
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    step = int(idx[1]) - int(idx[0]) if n > 1 else 1
    if step > 0 and all(int(idx[i + 1]) - int(idx[i]) == step for i in range(n - 1)):
        return torch.arange(int(idx[0]), int(idx[0]) + step * n, step, dtype=torch.int64, device=dev)
    ring = engine.__dict__.get('_idx_ring')
    if ring is None or ring['cap'] < n or ring['dev'] != dev:
        cap = max(4096, n)
        ring = engine.__dict__['_idx_ring'] = dict(cap=cap, dev=dev, k=0, host=[torch.empty(cap, dtype=torch.int64).pin_memory() for _ in range(4)],
                                                   ev=[None] * 4)
    k = ring['k'] = (ring['k'] + 1) & 3
    if ring['ev'][k] is not None:
        ring['ev'][k].synchronize()
    host = ring['host'][k]
    host[:n] = torch.as_tensor(np.asarray(idx, dtype=np.int64))
    out = host[:n].to(dev, non_blocking=True)
    ev = ring['ev'][k] = ring['ev'][k] or torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return out


def _small_frames(engine, frames, idx, sal_h, sal_w, dev):
    """Frames idx at saliency size on the device, whatever the container (CUDA tensor, on-device generator, host array)."""
    import torch
    if torch.is_tensor(frames) and frames.is_cuda:
        return engine.resize_frames(frames[device_index(engine, idx, frames.device)].to(dev).contiguous(), sal_h, sal_w)
    if not torch.is_tensor(frames) and hasattr(frames, 'select'):             # an on-device generator (synth.LazyBlobVideo)
        if getattr(frames, 'accepts_device_index', False):
            return engine.resize_frames(frames.select(idx, index=device_index(engine, idx, dev)).to(dev).contiguous(), sal_h, sal_w)
        return engine.resize_frames(frames.select(idx).to(dev).contiguous(), sal_h, sal_w)
    feed = getattr(engine, '_host_feed', None)
    if feed is None:
        feed = engine._host_feed = _HostFeed(engine)
    if not torch.is_tensor(frames) and hasattr(frames, 'pinned') and hasattr(frames, 'rows'):   # selected frames in pinned host memory (synth.HostSelectedVideo)
        return feed.downscale(frames.pinned, frames.rows(idx), sal_h, sal_w)
    host = frames if torch.is_tensor(frames) else np.asarray(frames)
    if (host.dtype not in (np.uint8, torch.uint8)) or host.ndim != 4 or host.shape[3] != 3:
        raise TypeError('frames must be uint8 [n,h,w,3] RGB')
    return feed.downscale(host, idx, sal_h, sal_w)


def plan_video(video, crop_params, engine=None, shot_net=None, shots=None):
    """The ingest's host bookkeeping for one video dict (smartVidCrop.py:621-718, :740-758; the video path :379-399,
    :452-456): frame selection, the selected-frame view of the shots, the cut-blend flags and which selected frames keep
    an all-zero map (the last one of every read batch: the reference's off-by-one).  Shot detection runs here when the
    dict has no ``trans_inds`` (device work on the current stream) unless the caller did it already (``shots`` = what
    detect_shots returned for this video: the scheduler's planner thread runs it ahead of the lanes).  Shared by
    ingest_frames (one video at a time) and the multi-video scheduler (retargetvid_amd/scheduler.py)."""
    fr, frame_count, w, h = video['fr'], int(video['frame_count']), int(video['w']), int(video['h'])
    frames = video['frames']              # ndarray / CUDA tensor [n,h,w,3] u8 RGB, or an object with __len__ and .select(idx)
    n_frames = len(frames)
    dsr = float(max(w, h)) / crop_params['max_input_d']
    sal_h, sal_w = int(h / dsr), int(w / dsr)
    trans_probs = None
    if video.get('trans_inds') is None:
        if shots is None:
            if shot_net is None:
                raise ValueError('the video dict has no trans_inds: pass shot_net= (a transnetv1_handler.ShotTransNet) to run shot '
                                 'detection inside the ingest, as the reference\'s video path does')
            shots = detect_shots(frames, fr, crop_params, net=shot_net, engine=engine, trans_threshold=TRANS_THRESHOLD)
        trans_probs = shots['trans_probs']
        true_inds, map2orig, batches = _select_frames_video(n_frames, frame_count, trans_probs, TRANS_THRESHOLD,
                                                            crop_params['skip'], crop_params['read_batch'])
        seg = np.array(shots['segmentation'], dtype=np.int32)
    else:
        trans_inds = [int(v) for v in video['trans_inds']]
        true_inds, map2orig, batches = _select_frames(n_frames, frame_count, trans_inds, crop_params['skip'],
                                                      crop_params['read_batch'])
        scenes = []
        for i in range(len(trans_inds)):
            if frame_count - trans_inds[i] < 2:
                break
            if i + 1 < len(trans_inds):
                scenes.append([trans_inds[i], trans_inds[i + 1] - 1])
        if not scenes:
            raise ValueError('trans_inds %r yields no scenes; pass at least [0, frame_count]' % (trans_inds,))
        seg = np.array(scenes, dtype=np.int32)
    if int(seg[0][0]) != 0:
        # predictions_to_scenes (:211-228) opens the first scene where the transition probability first DROPS below the threshold, so a
        # video that begins inside a transition (a fade-in; a shot network that reports "transition" on the opening frames) leaves its
        # first frames in no scene, and so does a trans_inds list that does not begin with 0.  The reference has no check for it (:799-815)
        # and fails later, in the per-shot interpolation; here it is an error before any device work.
        raise ValueError('the first scene starts at frame %d, not 0: frames 0..%d belong to no scene (%s)' % (
            int(seg[0][0]), int(seg[0][0]) - 1,
            'the transition probability is above the threshold from the first frame on' if trans_probs is not None else 'trans_inds must begin with 0'))
    seg_sel = np.array([[map2orig[v] for v in row] for row in seg], dtype=np.int32)
    n_sel = len(true_inds)
    # the reference's sanity checks (:799-825), as exceptions
    if n_frames > frame_count or len(map2orig) != n_frames or seg[-1][-1] != n_frames - 1 or \
            seg_sel[-1][-1] != n_sel - 1 or map2orig[-1] != n_sel - 1:
        raise ValueError('inconsistent frame / segment bookkeeping (frame_count=%d, frames=%d, segmentation end=%d)'
                         % (frame_count, n_frames, int(seg[-1][-1])))
    zero_map = np.zeros(n_sel, bool)
    for first, cnt in batches:
        if cnt:
            zero_map[first + cnt - 1] = True
    return dict(fr=fr, frame_count=frame_count, w=w, h=h, n_frames=n_frames, sal_h=sal_h, sal_w=sal_w, true_inds=true_inds,
                map2orig=map2orig, batches=batches, seg=seg, seg_sel=seg_sel, n_sel=n_sel, trans_probs=trans_probs,
                zero_map=zero_map, flags=blend_flags(n_sel, seg_sel) if crop_params['clust_filt'] else None)


def _ingest_dict(plan, smaps, xy_stream=None):
    """The dict the ingest hands on (the reference's vid_data after ingest_pickle, smartVidCrop.py:826-836)."""
    vd = dict(smaps_dev=smaps, segmentation=plan['seg'], segmentation_sel=plan['seg_sel'], true_inds=plan['true_inds'],
              inds_to_orig=plan['map2orig'], fr=plan['fr'], fc=plan['n_frames'], fc_sel=plan['n_sel'], h_orig=plan['h'],
              w_orig=plan['w'], h_process=plan['sal_h'], w_process=plan['sal_w'],
              n_net_maps=int(plan['n_sel'] - np.count_nonzero(plan['zero_map'])))     # maps the network writes (the rest stay all-zero: the off-by-one)
    if plan['trans_probs'] is not None:
        vd['trans_probs'] = plan['trans_probs']
    if xy_stream is not None:
        vd['xy_stream'] = xy_stream
    return vd


def ingest_frames(video, crop_params, engine=None, verbose=False, shot_net=None, stream_batch=0):
    """Counterpart of ingest_pickle (smartVidCrop.py:560-836) for an in-memory video dict -- and, when the dict carries no
    ``trans_inds`` and a shot network is given (``shot_net``: transnetv1_handler.ShotTransNet), of the video path
    read_and_segment_video (:234-556): TransNet runs per read batch with the reference's overlap (:248-260, :353-374),
    the after-cut selection follows its transition probabilities (:394-396) and the scenes come from
    predictions_to_scenes with the end-of-segment fix (:452-456).
    stream_batch > 0: threshold, cluster filter, cut blend and centres run INSIDE the ingest, stream_batch maps at a time
    behind each piece of the network (pipeline.StreamPipeline: one tail round per call, chains carried over); the dict
    then also carries ``xy_stream`` (centres) and the filtered maps, and smart_vid_crop skips its own tail.  Results
    are those of the whole-video call.
    The down-scale to saliency size and the UNISAL forward run on the device.  Keeps the
    reference's off-by-one: the last selected frame of each read batch gets an all-zero map.
    Host frames (ndarray) go through the pinned, double-buffered feed above; CUDA tensors and
    on-device generators (``.select``) are used where they are."""
    import torch
    engine = engine or get_engine()
    t = time.perf_counter()
    plan = plan_video(video, crop_params, engine=engine, shot_net=shot_net)
    frames = video['frames']
    true_inds, batches, seg_sel = plan['true_inds'], plan['batches'], plan['seg_sel']
    sal_h, sal_w = plan['sal_h'], plan['sal_w']
    sc_register_time(t, '_read_shot_det')
    t = time.perf_counter()
    n_sel = len(true_inds)
    dev = engine.device
    smaps = torch.zeros((n_sel, sal_h, sal_w), dtype=torch.uint8, device=dev)
    pipe, xy_stream, flags_all = None, None, None
    if stream_batch and crop_params['clust_filt']:
        from . import pipeline as _pl
        flags_all = plan['flags']
        cache = engine.__dict__.setdefault('_pipes', {})        # ring + pinned buffers are re-used from video to video
        key = (sal_h, sal_w, int(stream_batch), torch.cuda.current_stream(dev).cuda_stream)
        pipe = cache.get(key)
        if pipe is None:
            pipe = cache[key] = _pl.StreamPipeline(engine, crop_params, sal_h, sal_w, batch=int(stream_batch))
        pipe.reset()
        pipe.CP, pipe.maps_out = crop_params, smaps
        xy_stream = np.full((n_sel, 2), np.nan)
        fed = 0

        def feed_tail(upto):
            """threshold + one clustering round for the maps [fed, upto) that are complete (in stream order)."""
            nonlocal fed
            while fed < upto:
                k = min(int(stream_batch), upto - fed)
                if len(pipe.calls) >= pipe.depth:
                    for g, x, y in pipe.collect():
                        xy_stream[g] = (x, y)
                chunk = smaps[fed:fed + k]
                engine.threshold_(chunk, crop_params['t_threshold'])
                pipe.submit_maps(chunk, flags_all[fed:fed + k])
                fed += k
    for first, cnt in batches:
        if cnt > 1 and pipe is not None:
            # piece by piece: the tail of a piece follows its network on the stream
            for s0 in range(first, first + cnt - 1, int(stream_batch)):
                idx = true_inds[s0:min(s0 + int(stream_batch), first + cnt - 1)]
                smaps[s0:s0 + len(idx)] = engine.saliency(_small_frames(engine, frames, idx, sal_h, sal_w, dev))
                feed_tail(s0 + len(idx))
            feed_tail(first + cnt)                    # the batch's last selected frame keeps its all-zero map (the off-by-one)
            continue
        if pipe is not None:
            feed_tail(first + cnt)
            continue
        if cnt > 1:
            idx = true_inds[first:first + cnt - 1]
            smaps[first:first + cnt - 1] = engine.saliency(_small_frames(engine, frames, idx, sal_h, sal_w, dev))
    if pipe is not None:
        for g, x, y in pipe.finish():
            xy_stream[g] = (x, y)
    torch.cuda.current_stream(dev).synchronize()      # the caller's stream only: other videos may be in flight on theirs
    sc_register_time(t, '_read_sal_det')
    return _ingest_dict(plan, smaps, xy_stream)


def detect_shots(frames, fr, crop_params=None, net=None, engine=None, trans_threshold=0.1):
    """Shot detection of the reference's video path (smartVidCrop.py:248-372, :452-457) on the device: frames
    [n, h, w, 3] uint8 (CUDA tensor, NumPy / pinned host tensor, or an on-device generator with .select) ->
    dict(trans_probs, segmentation, trans_inds).  `net` is a transnetv1_handler.ShotTransNet (it owns the weights; the
    reference's checkpoint does not ship with it).  The frames are down-scaled to 48 x 27 in read_batch-sized pieces
    (host inputs through the pinned, double-buffered feed), so a long 1080p / 4K video never sits on the device in
    full; `segmentation` carries the reference's end-of-segment fix (every scene ends where the next one starts, the
    last one on the last frame) and trans_inds is consistent with it."""
    import torch
    from . import transnetv1_handler as T
    if net is None:
        raise ValueError('detect_shots needs net= (a transnetv1_handler.ShotTransNet holding the weights)')
    CP = crop_params or sc_init_crop_params()
    eng = engine or net.eng
    n = len(frames)
    th, tw = T.ShotTransNetParams.INPUT_HEIGHT, T.ShotTransNetParams.INPUT_WIDTH
    small = torch.empty((n, th, tw, 3), dtype=torch.uint8, device=eng.device)
    step = max(1, int(CP['read_batch']))
    for s0 in range(0, n, step):
        idx = list(range(s0, min(n, s0 + step)))
        if torch.is_tensor(frames) and frames.is_cuda:
            small[s0:s0 + len(idx)] = eng.resize_frames(frames[s0:s0 + len(idx)].contiguous(), th, tw)
        elif not torch.is_tensor(frames) and hasattr(frames, 'select'):
            small[s0:s0 + len(idx)] = eng.resize_frames(frames.select(idx).to(eng.device).contiguous(), th, tw)
        else:
            feed = getattr(eng, '_host_feed', None)
            if feed is None:
                feed = eng._host_feed = _HostFeed(eng)
            host = frames if torch.is_tensor(frames) else np.asarray(frames)
            small[s0:s0 + len(idx)] = feed.downscale(host, idx, th, tw)
    probs = T.video_transition_probs(net, small, fr, CP['read_batch'])
    seg = np.array(T.predictions_to_scenes(probs, threshold=trans_threshold), dtype=np.int32)
    for i in range(len(seg) - 1):                      # "shot segmentation FIX" (smartVidCrop.py:452-456)
        seg[i][1] = seg[i + 1][0] - 1
    seg[-1][1] = len(probs) - 1
    return dict(trans_probs=probs, segmentation=seg, trans_inds=T.shots_to_trans_inds(seg, len(probs)))


def blend_flags(fc_sel, segmentation_sel):
    """Which maps are blended into their successor (smartVidCrop.py:2324-2327, :2369-2370)."""
    cuts = set(int(s[0]) for s in segmentation_sel)
    cuts.add(int(segmentation_sel[-1][1]))
    flags = np.zeros(fc_sel, np.uint8)
    for i in range(fc_sel):
        if i < fc_sel - 2 and ((i - 1) in cuts or i in cuts or (i + 1) in cuts):
            flags[i] = 1
    return flags


class _LazySmaps(dict):
    """VD dict whose 'smaps' ([H,W,n] u8, the reference's layout) is materialised from the device
    only when somebody asks for it, and whose 'bbs' list is made from the array 'bbs_np' on first access (0.4 ms of
    Python per video and ratio that the multi-video job never needs).  The lazy keys answer `in`, get(), keys(), len() and
    iteration like keys that are present WITHOUT building their values; [], get(), items(), values(), dict(VD) and pickling
    build them: a caller of the reference gets a plain dict with both."""
    _LAZY = {'smaps': 'smaps_dev', 'bbs': 'bbs_np'}

    def __missing__(self, key):
        if key == 'smaps' and dict.__contains__(self, 'smaps_dev'):
            self['smaps'] = np.ascontiguousarray(dict.__getitem__(self, 'smaps_dev').permute(1, 2, 0).cpu().numpy())
            return dict.__getitem__(self, 'smaps')
        if key == 'bbs' and dict.__contains__(self, 'bbs_np'):          # one [x1,y1,x2,y2] list per frame, as the reference returns them
            self['bbs'] = dict.__getitem__(self, 'bbs_np').tolist()
            return dict.__getitem__(self, 'bbs')
        raise KeyError(key)

    def _materialise(self):
        for k, src in self._LAZY.items():
            if not dict.__contains__(self, k) and dict.__contains__(self, src):
                self[k]
        return self

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key in self._LAZY and dict.__contains__(self, self._LAZY[key]))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _pending(self):
        """Lazy key names that would appear on materialisation (no value is built)."""
        return [k for k, src in self._LAZY.items() if not dict.__contains__(self, k) and dict.__contains__(self, src)]

    def keys(self):
        """Key NAMES only: the lazy keys are reported without building their values (a device-to-host copy of every map)."""
        return list(dict.keys(self)) + self._pending()

    def items(self):
        return dict.items(self._materialise())

    def values(self):
        return dict.values(self._materialise())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return dict.__len__(self) + len(self._pending())

    def __bool__(self):
        return dict.__len__(self) > 0

    def copy(self):
        return _LazySmaps(dict.copy(self))

    def __reduce__(self):
        return (dict, (dict(dict.items(self._materialise())),))


def smart_vid_crop(video_path, CP=None, demo_fn='', final_vid_fn='', plots_fn='', frames_dir='',
                   temp_path=None, verbose=False, save_vid=True, callback_progress=None,
                   callback_session=None, callback_status=None, copy_sound=False, engine=None, shot_net=None, stream_batch=0):
    """Saliency -> crop windows for one video.  Returns (VD, smart_crop_results) like the
    reference; VD['bbs'] holds one [x1,y1,x2,y2] per decoded frame.  A video dict without ``trans_inds`` takes the
    reference's video path: ``shot_net`` (transnetv1_handler.ShotTransNet) detects the shots inside the ingest."""
    import torch
    sc_init_time()
    results = {}
    if CP is None:
        CP = sc_init_crop_params()
    if CP['exit_on_spread_sal'] or CP['exit_on_low_cvrg'] or CP['t_border'] != -1:
        raise NotImplementedError('mean-saliency / coverage gates and border detection are disabled in both '
                                  'published parameter sets and are not part of this path')
    if save_vid and (final_vid_fn or demo_fn):
        raise NotImplementedError('video rendering is outside the saliency-to-crop path; pass save_vid=False')
    engine = engine or get_engine()
    if callback_status is not None and callback_session is not None:
        callback_status(callback_session, 'sc', 'SC VIDEO ANALYSIS', 'smart-cropping video analysis')
    if isinstance(video_path, dict):
        video = video_path
    elif isinstance(video_path, str) and video_path.endswith('.pkl'):
        with open(video_path, 'rb') as fp:
            video = pickle.load(fp)
    elif _video_reader is not None:
        video = _video_reader(video_path, CP)
    else:
        raise NotImplementedError('decoding video files and TransNetV1 shot detection stay outside this '
                                  'package: pass the ingest_pickle dict (fr, frame_count, w, h, frames, '
                                  'trans_inds), a .pkl holding it, or install a reader with set_video_reader() '
                                  '(retargetvid_amd/ingest.py)')
    # the reference's feature cache (smartVidCrop.py:2244-2256, :2276-2280): with temp_path, the analysis of a NAMED video
    # (a file path, or a dict with a 'name') -- selection, segmentation and the raw saliency maps -- is pickled to
    # <temp_path>/<name>.pkl after the ingest and read back instead of it the next time (another target ratio, other
    # tail parameters).  The reference builds the file name from an undefined local (it works only through a global of its
    # __main__); here it is the video's own name.  Not combined with stream_batch (that ingest also runs the tail).
    cache_fn = None
    if temp_path is not None and not stream_batch:
        name = video.get('name') if isinstance(video, dict) else None
        if name is None and isinstance(video_path, str):
            name = os.path.splitext(os.path.basename(video_path))[0]
        if name:
            cache_fn = os.path.join(temp_path, str(name) + '.pkl')
    # what the cached analysis depends on besides the video: a file written under other values is ignored and replaced (the
    # reference keys its cache by the name alone, :2244-2256, and would silently re-use it)
    cache_key = None
    if cache_fn is not None:
        cache_key = dict(skip=CP['skip'], read_batch=CP['read_batch'], max_input_d=CP['max_input_d'],
                         frame_count=int(video['frame_count']), shots='net' if video.get('trans_inds') is None else
                         [int(v) for v in video['trans_inds']], weights=getattr(engine, 'weights_id', None))
    cached = None
    if cache_fn is not None and os.path.isfile(cache_fn):
        with open(cache_fn, 'rb') as fp:
            cached = pickle.load(fp)
        if cached.get('cache_key') != cache_key:
            cached = None
    if cached is not None:
        VD = _LazySmaps({k: v for k, v in cached.items() if k not in ('smaps_nhw', 'cache_key')})
        VD['smaps_dev'] = torch.from_numpy(cached['smaps_nhw']).to(engine.device)
    else:
        VD = _LazySmaps(ingest_frames(video, CP, engine, verbose=verbose, shot_net=shot_net, stream_batch=stream_batch))
        if cache_fn is not None:
            os.makedirs(temp_path, exist_ok=True)
            out = {k: v for k, v in dict.items(VD) if k not in ('smaps_dev', 'smaps')}     # (dict.items: nothing lazy is materialised)
            out['smaps_nhw'] = VD['smaps_dev'].cpu().numpy()
            out['cache_key'] = cache_key
            with open(cache_fn, 'wb') as fp:
                pickle.dump(out, fp)
    if callback_status is not None and callback_session is not None:
        callback_status(callback_session, 'sc', 'SC PROCESSING', 'smart-cropping main process')
    VD, results = after_ingest(VD, CP, engine, verbose=verbose)
    if callback_status is not None and callback_session is not None:
        callback_status(callback_session, 'sc', 'SC RENDERING', 'smart-cropping rendering')
    return VD, results


def after_ingest(VD, CP, engine, verbose=False):
    """Everything of smart_vid_crop behind the ingest (smartVidCrop.py:2293-2614 without the rendering): destination
    size, threshold + cluster filter + centres on the device (skipped when the ingest or the multi-video scheduler has
    run them already: VD['xy_stream']), then the host stages -- empty-centre fill, focus stability, interpolation,
    low-pass, LOESS, boxes.  One function for the one-video call and for retargetvid_amd/scheduler.py, so that both give
    the same windows by construction.  -> (VD, smart_crop_results)."""
    results = {}
    VD['segm_backup'] = VD['segmentation'].copy()

    t = time.perf_counter()
    VD = sc_calc_dest_size(VD, CP, verbose=verbose)
    sc_register_time(t, '_calc_dest_size')
    t = time.perf_counter()
    VD['border_t'] = VD['border_b'] = VD['border_l'] = VD['border_r'] = 0
    sc_register_time(t, '_border_det')
    VD['mean_sal_score'] = None
    VD['mean_cvrg_score'] = None

    maps = VD['smaps_dev']
    t = time.perf_counter()
    ppl = VD.get('pixels_per_grey_level_at_threshold')
    near = None
    if 'xy_stream' not in VD:
        # the regime diagnostic (include/svc.h: svc_threshold_census): pixels of the raw maps at t - 1, t, t + 1 per map and level.
        # (maps - (t - 1)) wraps in uint8, so the three levels are exactly the values 0..2: one temporary, no synchronisation here --
        # the count travels to the host with the centres below
        tt = int(CP['t_threshold'])
        if maps.numel() and 2 <= tt <= 254:             # (t = 1: level 0 is also the value of the all-zero rows the off-by-one leaves)
            near = (maps - (tt - 1)).le_(2).sum()
        engine.threshold_(maps, CP['t_threshold'])
    sc_register_time(t, '_thresh')                     # (enqueue time; the stream is synchronised by the D2H below)

    t = time.perf_counter()
    if 'xy_stream' in VD:                              # the streaming ingest has run the tail already (stream_batch)
        xy = VD['xy_stream']
    else:
        flags = blend_flags(VD['fc_sel'], VD['segmentation_sel']) if CP['clust_filt'] else None
        xy_dev = engine.cluster_center_(maps, flags, CP)
        if near is not None:                           # one D2H: n x 2 doubles + the census count
            import torch
            both = torch.cat([xy_dev.reshape(-1), near.to(torch.float64).reshape(1)]).cpu().numpy()
            xy = both[:-1].reshape(-1, 2)
            ppl = float(both[-1]) / (3.0 * max(1, int(VD.get('n_net_maps', VD['fc_sel']))))
        else:
            xy = xy_dev.cpu().numpy()                  # one D2H of n x 2 doubles
    results['cuts_clust'] = 0
    sc_register_time(t, '_clustering')

    t = time.perf_counter()
    xy = np.asarray(xy, np.float64).reshape(-1, 2)
    sc_register_time(t, '_center_of_mass')

    t = time.perf_counter()
    if np.isnan(xy[:, 0]).any():                       # (native: svc_host_fill_empty_centres)
        xy, still_empty = temporal.fill_empty_centres(xy, VD['segmentation_sel'])
        if still_empty:
            raise ValueError('no saliency centre found in any selected frame')
    VD['dx'], VD['dy'] = xy[:, 0].tolist(), xy[:, 1].tolist()
    sc_register_time(t, '_center_empty_handle')
    VD['jumps'] = [255] * len(VD['dx'])
    VD['jumps_inds'] = []
    VD['dxnf'], VD['dynf'] = list(VD['dx']), list(VD['dy'])
    t = time.perf_counter()
    if CP['focus_stability']:            # best settings: hold the focus across short low-saliency jumps (host)
        maps_host = VD['smaps_dev'].cpu().numpy()        # frame-major [n,h,w]: no [H,W,n] transposition (VD['smaps'] stays lazy)
        xy, VD['jumps'], VD['jumps_inds'] = temporal.focus_stability_native(xy, maps_host, VD['fr'], CP)
        VD['dx'], VD['dy'] = xy[:, 0].tolist(), xy[:, 1].tolist()
    sc_register_time(t, '_focus_stability')

    t = time.perf_counter()                            # interpolation + low-pass + LOESS / Savitzky-Golay: one native call
    xi, yi, xs, ys = temporal.centres_to_series(xy, VD['true_inds'], VD['segmentation'], VD['segmentation_sel'], VD['fc'],
                                                VD['fr'], CP)
    VD['dxi'], VD['dyi'] = xi.tolist(), yi.tolist()
    sc_register_time(t, '_interpolation')
    t = time.perf_counter()
    VD['dxs'], VD['dys'] = xs.tolist(), ys.tolist()
    VD['dxs_smooth'], VD['dys_smooth'] = list(VD['dxs']), list(VD['dys'])    # sc_compute_bb overwrites dxs / dys (:995-999)
    sc_register_time(t, '_smooth')
    t = time.perf_counter()
    VD = sc_compute_bb(VD, CP, verbose=verbose)
    sc_register_time(t, '_bb')
    t = time.perf_counter()
    if CP['shift_time'] > 0:
        temporal.shift_time(VD['bbs'], CP['shift_time'])
        VD['bbs_np'] = np.asarray(VD['bbs'], np.int64)
    sc_register_time(t, '_shift')

    results['result'] = 'smart cropped'
    results['info'] = ' (%dx%d)->(%dx%d)->(%dx%d)->(%dx%d)\n' % (
        VD['h_orig'], VD['w_orig'], VD['h_process'], VD['w_process'], VD['h_final'], VD['w_final'],
        VD['fbb_h'], VD['fbb_w'])
    # NOT a key of the reference's results: how many pixels of a raw saliency map sit on each of the three grey levels around
    # t_threshold (mean per map).  Two correct fp32 implementations of the network differ by one grey level on ~0.3 % of the
    # pixels; ~7 - 45 pixels per level keep the crop windows identical to the reference CPU path's, ~500 (a checkpoint whose
    # maps are flat around the threshold) make a fifth of them differ by more than a pixel.  The multi-video
    # job reports the same per-video figure (counted by the network's last kernel: svc_saliency_census_u8).
    results['pixels_per_grey_level_at_threshold'] = None if ppl is None else round(float(ppl), 2)
    results['params'] = ''.join(' %-18s : %s\n' % (k, str(v)) for k, v in CP.items())
    results['mean_sal_score'] = VD['mean_sal_score']
    results['mean_sal_score_t'] = CP['t_sal']
    results['coverage_score'] = VD['mean_cvrg_score']
    results['coverage_score_t'] = CP['t_cvrg']
    t_dict = sc_all_times(VD['fc'] / VD['fr'])
    for k in t_dict:
        if k.startswith('_'):
            results['t_' + k] = t_dict[k]
    for k in t_dict:
        if not k.startswith('_'):
            results['t_' + k] = t_dict[k]
    return VD, results


smartVidCrop = smart_vid_crop      # BASELINE.json's spelling of the entry point


def other_ratio(base, cp):
    """(VD, results) of ANOTHER target ratio from those of a finished run of the same video: nothing before
    sc_calc_dest_size depends on out_ratio, so only destination size and boxes are recomputed."""
    import copy
    VD = _LazySmaps({k: (copy.copy(v) if isinstance(v, list) else v) for k, v in dict.items(base[0]) if k != 'bbs'})
    res = dict(base[1])
    VD = sc_calc_dest_size(VD, cp)
    VD['dxs'], VD['dys'] = list(base[0]['dxs_smooth']), list(base[0]['dys_smooth'])   # smoothing does not depend on the ratio
    VD = sc_compute_bb(VD, cp)
    if cp['shift_time'] > 0:
        temporal.shift_time(VD['bbs'], cp['shift_time'])
        VD['bbs_np'] = np.asarray(VD['bbs'], np.int64)
    res['params'] = ''.join(' %-18s : %s\n' % (k, str(v)) for k, v in cp.items())
    res['info'] = ' (%dx%d)->(%dx%d)->(%dx%d)->(%dx%d)\n' % (
        VD['h_orig'], VD['w_orig'], VD['h_process'], VD['w_process'], VD['h_final'], VD['w_final'],
        VD['fbb_h'], VD['fbb_w'])
    return VD, res


def smart_vid_crop_ratios(video_path, CP, ratios, engine=None, verbose=False, stream_batch=0):
    """Several target aspect ratios for one video with the saliency / clustering work done once
    (the reference's driver re-runs the whole pipeline per ratio, smartVidCrop.py:2722-2775, or
    re-uses its pickled feature cache :2244-2256).  -> {ratio: (VD, smart_crop_results)}; results
    are identical to calling smart_vid_crop once per ratio, because nothing before
    sc_calc_dest_size depends on out_ratio."""
    out = {}
    base = None
    for ratio in ratios:
        cp = dict(CP, out_ratio=ratio)
        if base is None:
            base = smart_vid_crop(video_path, cp, save_vid=False, engine=engine, verbose=verbose, stream_batch=stream_batch)
            out[ratio] = base
        else:
            out[ratio] = other_ratio(base, cp)
    return out


def crop_videos(videos, CP, ratios=None, workers=12, state_dict=None, seed=0, stream_batch=0, packed=None, shot_net=None,
                stats=None):
    """Many videos on one GPU.  Default (``packed``, whenever the cluster filter is on): the job-level scheduler of
    retargetvid_amd/scheduler.py -- ``workers`` lanes (engine + HIP stream), the selected frames of consecutive videos
    packed into full network chunks across video boundaries, one tail round per chunk, host stages on a thread pool;
    ``stats`` (a dict) receives the run's counters.  packed=False: the round-3 form described next.
    ``videos`` may be zero-argument CALLABLES (videos built on demand).  Without ``shot_net`` the feeder thread calls them one at a
    time as lanes take them; WITH ``shot_net`` the scheduler's planner threads call them -- up to three concurrently, on threads of
    their own, with a planner's HIP stream current: such callables must be thread-safe -- and at most ``lanes + 2 per planner``
    videos are materialised ahead of the lanes (scheduler.JobScheduler(plan_ahead=)), however long the job is.

    ``workers`` videos in flight, one per worker thread: every worker thread owns an engine
    (weights + workspace) and a HIP stream and runs smart_vid_crop_ratios on its share, so the
    low-occupancy clustering tail and the host-side temporal stages of one video overlap the network of
    the next ones (what bench.py does with its batches).  ``videos``: a sequence of ingest_pickle dicts or
    of zero-argument callables producing them (built on the worker's stream).  stream_batch > 0: every video's tail runs
    inside its ingest, stream_batch maps at a time (pipeline.StreamPipeline).  Returns a list, in input
    order, of {ratio: (VD, smart_crop_results)}; each entry equals a sequential smart_vid_crop_ratios call."""
    import torch
    ratios = tuple(ratios) if ratios else (CP['out_ratio'],)
    videos = list(videos)
    if packed is None:
        packed = bool(CP['clust_filt'])
    if packed:
        from . import scheduler as _sched
        # no more lanes than videos (a lane = an engine: weights, workspace, 0.6 GB of frame and map rows), and a lane's rows
        # sized by the job where that is known without building the videos
        lanes = max(1, min(int(workers), len(videos) or 1))
        js = _sched.JobScheduler(CP, ratios, lanes=lanes, state_dict=state_dict, seed=seed, shot_net=shot_net,
                                 lane_rows=_sched.lane_rows_for(videos, CP, lanes))
        try:
            out = js.run(videos)
            if stats is not None:
                stats.update(js.stats)
        finally:
            js.close()
        return out
    workers = max(1, min(int(workers), len(videos) or 1))
    dev = torch.device('cuda', torch.cuda.current_device())
    from . import ops as _ops
    if state_dict is None:                                   # built once, not once per worker (33 of an engine's 43 ms)
        from . import weights as _weights
        state_dict = _weights.make_synthetic_state_dict(seed)
    engines = [_ops.Engine(state_dict, device=dev.index, seed=seed) for _ in range(workers)]
    out, errors = [None] * len(videos), []

    def run(k):
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(device=dev)
        try:
            with torch.cuda.stream(stream):
                for i in range(k, len(videos), workers):
                    v = videos[i]() if callable(videos[i]) else videos[i]
                    out[i] = smart_vid_crop_ratios(v, CP, ratios, engine=engines[k], stream_batch=stream_batch)
                stream.synchronize()
        except BaseException as e:                       # surfaced in the caller's thread
            errors.append(e)

    # a worker coming back from a launch or a wait must get the interpreter back quickly to keep its stream fed: the default
    # 5 ms switch interval lets another worker's NumPy-free Python stretch (bookkeeping, SciPy set-up) hold it that long
    import sys
    si = sys.getswitchinterval()
    if os.environ.get('SVC_SWITCH_INTERVAL_US'):
        sys.setswitchinterval(float(os.environ['SVC_SWITCH_INTERVAL_US']) * 1e-6)
    threads = [threading.Thread(target=run, args=(k,)) for k in range(workers)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        sys.setswitchinterval(si)
    for e in engines:
        e.close()
    if errors:
        raise errors[0]
    return out


def write_results(results_out, vid_fn, out_ratio, vid_data, info_dict):
    """The two files the evaluator reads (smartVidCrop.py:2730-2731, :2777-2785):
    <vid>_<w>-<h>.txt with one 'x1,y1,x2,y2' line per frame, and <same>_info.txt with key:value lines."""
    import os
    os.makedirs(results_out, exist_ok=True)
    suffix = vid_fn + '_' + str(out_ratio.replace(':', '-'))
    with open(os.path.join(results_out, suffix + '_info.txt'), 'w') as fp:
        for k in info_dict:
            fp.write(k + ':' + str(info_dict[k]) + '\n')
    with open(os.path.join(results_out, suffix + '.txt'), 'w') as fp:
        for bb in vid_data['bbs']:
            fp.write('%d,%d,%d,%d\n' % (bb[0], bb[1], bb[2], bb[3]))
    return os.path.join(results_out, suffix + '.txt')
