"""TransNet V1 shot-boundary detection on the device -- the counterpart of the reference's
3rd_party_libs/transnetv1/transnetv1_handler.py (same names, argument meaning and return values):

  ShotTransNetParams                  :8-15    (F, L, S, D, INPUT_WIDTH, INPUT_HEIGHT, CHECKPOINT_PATH)
  ShotTransNet(params, ...)           :17-24   the network; here `weights=` takes the TensorFlow-layout arrays
      .predict_raw(frames)            :93-97   [batch, frames, 27, 48, 3] uint8 -> [batch, frames] P(transition)
      .predict_frames(frames)         :99-100
      .predict_video(frames)          :102-130 100-frame windows, stride 50, edge frames repeated
  shot_preprocess_frame(s)(_list)     :133-150 cv2.resize to 48 x 27 (here: the library's OpenCV-exact down-scale)
  shots_from_predictions, assert_segmentation  :232-262
  scenes_from_predictions             transnet_utils.py:5-19;  predictions_to_scenes  smartVidCrop.py:214-230

The forward pass is hand-written HIP behind the C ABI (svc_transnet_load / svc_transnet_predict, csrc/svc_shot.hip); there
is no CPU fallback.  The reference restores a TensorFlow checkpoint that does not ship with it (note.txt:1) and that
this build cannot read (no TensorFlow): pass `weights=` (weights.make_transnet_state_dict for the synthetic network of the
tests, or arrays exported from the checkpoint under the reference's variable names)."""
import ctypes

import numpy as np
import torch

from . import _lib, ops, weights as _weights


class ShotTransNetParams:
    F = 16
    L = 3
    S = 2
    D = 256
    INPUT_WIDTH = 48
    INPUT_HEIGHT = 27
    CHECKPOINT_PATH = None


class ShotTransNet:
    def __init__(self, params=None, session=None, weights=None, engine=None, windows_per_call=16, _blob=None):
        self.params = params or ShotTransNetParams()
        p = self.params
        if (p.F, p.L, p.S, p.D, p.INPUT_WIDTH, p.INPUT_HEIGHT) != (16, 3, 2, 256, 48, 27):
            raise ValueError('only the F16 L3 S2 D256 network on 48x27 frames (the reference\'s configuration) is built')
        if weights is None and _blob is None:
            raise ValueError('ShotTransNet needs weights= (TensorFlow-layout arrays under the reference\'s variable names): the '
                             'checkpoint %r cannot be read without TensorFlow' % (p.CHECKPOINT_PATH,))
        self._own = engine is None
        self.eng = engine or ops.Engine(seed=0)
        self.windows_per_call = int(windows_per_call)
        blob = _blob if _blob is not None else np.ascontiguousarray(_weights.pack_transnet_blob(weights), np.float32)
        self._blob = blob
        _lib.check(self.eng.lib.svc_transnet_load(self.eng._h, blob.ctypes.data_as(ctypes.c_void_p), blob.size))

    def config(self):
        """The engine's TransNet knobs (svc_transnet_config_get): [matrix pipe (-1 = the engine's SVC_MX), 16-position tiles per
        wavefront, XCD-aware tile order]."""
        cfg = (ctypes.c_int32 * 3)()
        _lib.check(self.eng.lib.svc_transnet_config_get(self.eng._h, cfg))
        return list(cfg)

    def clone(self):
        """A second network with the same weights, matrix pipe and kernel knobs on an engine of its own (a handle's workspace
        serves one call at a time): what the job scheduler's planner threads run.  The knobs are copied handle to handle
        (svc_transnet_config_get / _set), not through the process environment.  Memory: the copy owns an engine (the saliency
        network's weights, ~15 MB; its workspace is only allocated by a saliency call) and, after its first prediction, a TransNet
        workspace of up to 1.6 GB (two activation buffers for windows_per_call windows)."""
        cfg = self.config()
        if cfg[0] < 0:                                        # "follow SVC_MX": pin what THIS engine resolved it to
            cfg[0] = int(self.eng.lib.svc_transnet_matrix_pipe(self.eng._h))
        net = ShotTransNet(self.params, windows_per_call=self.windows_per_call, _blob=self._blob)
        _lib.check(net.eng.lib.svc_transnet_config_set(net.eng._h, (ctypes.c_int32 * 3)(*cfg)))
        return net

    def matrix_pipe(self):
        """'f32', 'bf16x6' or 'bf16x3': what the convolution cells run on (svc_transnet_matrix_pipe; environment SVC_SHOT_MX
        when the engine is created, default = SVC_MX)."""
        return {0: 'f32', 6: 'bf16x6', 3: 'bf16x3'}[int(self.eng.lib.svc_transnet_matrix_pipe(self.eng._h))]

    def close(self):
        if self._own and self.eng is not None:
            self.eng.close()
        self.eng = None

    # -- the reference's methods --------------------------------------------------------------------------------
    def predict_raw_device(self, frames, rows=None):
        """CUDA uint8 [batch, frames, 27, 48, 3] -> CUDA float32 [batch, frames].  rows=(a, b): only columns a .. b - 1 of the
        result are wanted (svc_transnet_predict_rows: the layers are computed on the frames those depend on and nothing else;
        the kept columns are bit for bit those of the full pass, the others are unspecified)."""
        if not (torch.is_tensor(frames) and frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous()):
            raise TypeError('frames must be a contiguous CUDA uint8 tensor')
        assert frames.dim() == 5 and tuple(frames.shape[2:]) == (self.params.INPUT_HEIGHT, self.params.INPUT_WIDTH, 3), \
            ' [ShotTransNet] Input shape must be [batch, frames, height, width, 3].'
        nb, nt = int(frames.shape[0]), int(frames.shape[1])
        if rows is not None and (int(rows[0]), int(rows[1])) != (0, nt):
            out = torch.zeros((nb, nt), dtype=torch.float32, device=frames.device)
            _lib.check(self.eng.lib.svc_transnet_predict_rows(self.eng._h, ctypes.c_void_p(frames.data_ptr()), nb, nt, int(rows[0]), int(rows[1]),
                                                              ctypes.c_void_p(out.data_ptr()),
                                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return out
        out = torch.empty((nb, nt), dtype=torch.float32, device=frames.device)
        _lib.check(self.eng.lib.svc_transnet_predict(self.eng._h, ctypes.c_void_p(frames.data_ptr()), nb, nt,
                                                     ctypes.c_void_p(out.data_ptr()),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def predict_raw(self, frames):
        dev = torch.device('cuda', torch.cuda.current_device())
        t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames, np.uint8))
        return self.predict_raw_device(t.to(dev).contiguous()).cpu().numpy()

    def predict_frames(self, frames):
        return self.predict_video(frames)

    def predict_video(self, frames, keep=None):
        """[frames, 27, 48, 3] uint8 (NumPy or CUDA tensor) -> NumPy float32 [frames].
        keep=(a, b): only rows a .. b - 1 of the result are wanted (the caller drops the rest: the reference's call site predicts
        an array of read_batch + overlap rows, most of them the zero tail behind a 600-frame video, and keeps the video's rows,
        smartVidCrop.py:353-374).  A window's 50 outputs depend on that window's 100 frames alone, so only the windows whose
        outputs fall into [a, b) are computed -- the kept rows are bit for bit those of the full computation, the others read 0."""
        assert len(frames.shape) == 4 and tuple(frames.shape[1:]) == (self.params.INPUT_HEIGHT, self.params.INPUT_WIDTH, 3), \
            ' [ShotTransNet] Input shape must be [frames, height, width, 3].'
        dev = torch.device('cuda', torch.cuda.current_device())
        t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames, np.uint8))
        t = t.to(dev).contiguous()
        n = int(t.shape[0])
        wi_all = window_indices(n)
        k0, k1 = 0, len(wi_all)
        if keep is not None:                                  # window k yields rows 50 k .. 50 k + 49
            a, b = max(0, int(keep[0])), min(n, int(keep[1]))
            k0, k1 = (a // 50, (b - 1) // 50 + 1) if b > a else (0, 0)
        wi = torch.from_numpy(wi_all[k0:k1]).to(dev)
        out = torch.zeros(len(wi_all) * 50, dtype=torch.float32, device=dev)
        for i in range(0, len(wi), self.windows_per_call):
            win = t[wi[i:i + self.windows_per_call].reshape(-1)].reshape(-1, 100, *t.shape[1:]).contiguous()
            res = self.predict_raw_device(win, rows=(25, 75))[:, 25:75].reshape(-1)       # the reference keeps the middle 50 (:117-121)
            out[50 * (k0 + i):50 * (k0 + i) + res.numel()] = res
        return out[:n].cpu().numpy()


def window_indices(n):
    """Frame index of every slot of every window: windows of 100 where the first / last 25 frames belong to the previous /
    next window; 25 copies of the first frame in front, 25 + 50 - (n % 50 or 50) copies of the last behind (:104-121)."""
    pad_end = 25 + 50 - (n % 50 if n % 50 != 0 else 50)
    idx = np.concatenate([np.zeros(25, np.int64), np.arange(n, dtype=np.int64), np.full(pad_end, n - 1, np.int64)])
    wins, ptr = [], 0
    while ptr + 100 <= len(idx):
        wins.append(idx[ptr:ptr + 100])
        ptr += 50
    return np.stack(wins)


def shot_preprocess_frames(frames, engine=None):
    """[n, h, w, 3] uint8 -> [n, 27, 48, 3] with the library's OpenCV-exact INTER_LINEAR down-scale (cv2.resize(frame,
    (48, 27)) in the reference, :133-142).  CUDA in -> CUDA out; NumPy in -> NumPy out."""
    own = engine is None
    eng = engine or ops.Engine(seed=0)
    try:
        t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames, np.uint8))
        out = eng.resize_frames(t.cuda().contiguous(), ShotTransNetParams.INPUT_HEIGHT, ShotTransNetParams.INPUT_WIDTH)
        return out if torch.is_tensor(frames) else out.cpu().numpy()
    finally:
        if own:
            eng.close()


def shot_preprocess_frame(frame, engine=None):
    return shot_preprocess_frames(np.asarray(frame)[None], engine)[0]


def shot_preprocess_frames_list(frames, engine=None):
    return shot_preprocess_frames(np.stack([np.asarray(f) for f in frames]), engine)


def scenes_from_predictions(predictions, threshold=0.1):
    pred = (np.asarray(predictions) > threshold).astype(np.uint8)
    scenes, t, tp, start, i = [], -1, 0, 0, 0
    for i, t in enumerate(pred):
        if tp == 1 and t == 0:
            start = i
        if tp == 0 and t == 1 and i != 0:
            scenes.append([start, i])
        tp = t
    if t == 0:
        scenes.append([start, i])
    return np.array(scenes, dtype=np.int32)


def predictions_to_scenes(predictions, threshold=0.5):
    """smartVidCrop.py:214-230: scenes_from_predictions plus the "all frames are transitions" fix."""
    scenes = scenes_from_predictions(predictions, threshold)
    if len(scenes) == 0:
        return np.array([[0, len(predictions) - 1]], dtype=np.int32)
    return scenes


def assert_segmentation(shots, l, min_frames=12):
    """:209-230 (the reference compares with the literal 12, not min_frames; kept)."""
    shots = [list(s) for s in shots]
    shots = [s for s in shots if not (s[1] - s[0] < 12)]
    if len(shots) == 0:
        shots.append([0, l - 1])
    for i in range(len(shots) - 1):
        if shots[i][1] != (shots[i + 1][0] - 1):
            shots[i][1] = shots[i + 1][0] - 1
    if shots[-1][1] < l - 1:
        shots[-1][1] = l - 1
    return shots


def shots_from_predictions(predictions, threshold=0.1):
    shots = [list(s) for s in scenes_from_predictions(predictions, threshold)]
    return np.array(assert_segmentation(shots, len(predictions), min_frames=12), dtype=np.int32)


def shot_trans_net_handler_version():
    return '1.0'


# ---- the reference's call site (smartVidCrop.py:248-372): read batches with an overlap ---------------------------------
def video_transition_probs(net, frames_small, fr, read_batch=2000, predict=None):
    """Transition probability of every frame of a video the way read_and_segment_video obtains it: the video is cut in
    read batches of `read_batch` frames; each batch is predicted inside an array of read_batch + overlap frames
    (overlap = int(fr - 5)) whose head holds the last `overlap` rows of the previous batch's array -- ZEROS for the first
    batch -- and whose unused tail is zeros (smartVidCrop.py:258-260, :353-358, :369-374); rows overlap .. overlap + len of
    predict_frames' output are kept.  frames_small: [n, 27, 48, 3] uint8, CUDA tensor or NumPy.  -> NumPy float32 [n].
    `predict` replaces net.predict_frames (the tests pass the oracle)."""
    own = predict is None                      # the device network: only the windows whose rows are kept are computed (predict_video's keep)
    predict = predict or net.predict_frames
    is_t = torch.is_tensor(frames_small)
    n = int(frames_small.shape[0])
    overlap = int(fr - 5)
    size = read_batch + overlap
    probs, prev = [], None
    for si in range(0, n, read_batch):
        cur = frames_small[si:si + read_batch]
        ln = int(cur.shape[0])
        arr = torch.zeros((size,) + tuple(frames_small.shape[1:]), dtype=torch.uint8, device=frames_small.device) if is_t \
            else np.zeros((size,) + tuple(frames_small.shape[1:]), np.uint8)
        arr[overlap:overlap + ln] = cur
        if prev is not None and overlap > 0:
            arr[:overlap] = prev[size - overlap:]
        prev = arr
        full = net.predict_video(arr, keep=(overlap, overlap + ln)) if own else predict(arr)
        probs.append(np.asarray(full)[overlap:overlap + ln])
    return np.concatenate(probs).astype(np.float32) if probs else np.zeros(0, np.float32)


def shots_to_trans_inds(scenes, frame_count):
    """Scene list -> the `trans_inds` of the package's input dict (smartVidCrop.ingest_frames: scene i = trans_inds[i] ..
    trans_inds[i + 1] - 1): the scene starts plus the frame count.  NB: the reference's video path keeps the scenes
    themselves ([start, first transition frame]); for the one-frame hard cuts both describe the same partition."""
    return [int(s[0]) for s in scenes] + [int(frame_count)]
