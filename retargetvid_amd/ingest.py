"""Optional decode hand-off in front of the path.

The saliency-to-crop path starts at decoded RGB frames (the reference's ``ingest_pickle`` door,
smartVidCrop.py:560-836).  Decoding and TransNetV1 shot detection stay on the host and outside
this package; this module only adapts a decoder to that door:

    from retargetvid_amd import ingest, smartVidCrop as S
    S.set_video_reader(lambda path, CP: ingest.read_video_cv2(path, shot_detector=my_transnet))
    VD, res = S.smart_vid_crop('clip.mp4', CP, save_vid=False)

``read_video_cv2`` needs OpenCV (``cv2``), which is not part of this image; ``shot_detector`` is any
callable ``frames[n,h,w,3] u8 RGB -> iterable of transition frame indices`` (the reference runs
TransNetV1 at 27x48, transnetv1_handler.py:91-130, and thresholds its output with ``t_cut``);
without one the video is a single shot.
"""
import numpy as np


def video_dict(frames, fr, trans_inds=None, frame_count=None):
    """The ingest_pickle dict (smartVidCrop.py:568-573) for decoded RGB frames [n,h,w,3] u8."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    fc = int(frame_count) if frame_count is not None else n
    ti = sorted(set(int(t) for t in (trans_inds or []) if 0 < int(t) < n))
    return dict(fr=float(fr), frame_count=fc, w=w, h=h, frames=frames, trans_inds=[0] + ti + [n])


def read_video_cv2(path, shot_detector=None, max_frames=None):
    """Decode a file with OpenCV (BGR -> RGB, like smartVidCrop.py:330-331) into the ingest_pickle dict."""
    try:
        import cv2
    except ImportError as e:
        raise ImportError('read_video_cv2 needs OpenCV (cv2); install it or pass your own reader to '
                          'smartVidCrop.set_video_reader()') from e
    cap = cv2.VideoCapture(path)
    if not cap.isOpened():
        raise IOError('cannot open video %r' % (path,))
    fr = cap.get(cv2.CAP_PROP_FPS)
    frame_count = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
    out = []
    while max_frames is None or len(out) < max_frames:
        ok, bgr = cap.read()
        if not ok:
            break
        out.append(np.ascontiguousarray(bgr[:, :, ::-1]))
    cap.release()
    if not out:
        raise IOError('no frames decoded from %r' % (path,))
    frames = np.stack(out)
    trans = list(shot_detector(frames)) if shot_detector is not None else []
    return video_dict(frames, fr, trans, frame_count=max(frame_count, len(out)))
