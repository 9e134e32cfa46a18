"""Optional decode hand-off in front of the path.

The saliency-to-crop path starts at decoded RGB frames (the reference's ``ingest_pickle`` door,
smartVidCrop.py:560-836).  Decoding and TransNetV1 shot detection stay on the host and outside
this package; this module only adapts a decoder to that door:

    from retargetvid_amd import ingest, smartVidCrop as S
    S.set_video_reader(lambda path, CP: ingest.read_video_cv2(path, shot_detector=my_transnet))
    VD, res = S.smart_vid_crop('clip.mp4', CP, save_vid=False)

``read_video_cv2`` needs OpenCV (``cv2``), which is not part of this image (it has never run here); ``shot_detector`` is
any callable ``frames[n,h,w,3] u8 RGB -> iterable of transition frame indices`` (the reference runs TransNetV1 at 27x48,
transnetv1_handler.py:91-130, and thresholds its output with ``t_cut``); without one the video is a single shot.

``read_frames_pillow`` is the decoder that DOES run in this image (Pillow is installed; tests/test_host_logic.py and
tests/test_gpu_pipeline.py execute it): a directory of frame images (a video extracted to PNG / JPEG files, sorted by name) or
one multi-frame image file (GIF, APNG, WebP, TIFF).  ``shots=None`` leaves ``trans_inds`` out of the dict, so that
``smart_vid_crop(path, CP, shot_net=net)`` takes the reference's video path (TransNet V1 on the device decides the shots):

    S.set_video_reader(lambda path, CP: ingest.read_frames_pillow(path, fr=25.0, shots=None))
    VD, res = S.smart_vid_crop('frames_of_clip_017/', CP, save_vid=False, shot_net=net)
"""
import os

import numpy as np


def video_dict(frames, fr, trans_inds=None, frame_count=None, shots='given'):
    """The ingest_pickle dict (smartVidCrop.py:568-573) for decoded RGB frames [n,h,w,3] u8.  shots=None: no ``trans_inds``
    key (the entry point then runs shot detection itself: smart_vid_crop(..., shot_net=))."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    fc = int(frame_count) if frame_count is not None else n
    out = dict(fr=float(fr), frame_count=fc, w=w, h=h, frames=frames)
    if shots is not None:
        ti = sorted(set(int(t) for t in (trans_inds or []) if 0 < int(t) < n))
        out['trans_inds'] = [0] + ti + [n]
    return out


_IMAGE_EXT = ('.png', '.jpg', '.jpeg', '.bmp', '.tif', '.tiff', '.webp', '.ppm')


def read_frames_pillow(path, fr=None, shot_detector=None, max_frames=None, shots='given'):
    """Decode with Pillow into the ingest_pickle dict: ``path`` = a directory of frame images (sorted by file name; all of one
    size) or one multi-frame image file (GIF / APNG / WebP / TIFF).  Frames are converted to RGB like the reference's
    cv2.cvtColor(frame, COLOR_BGR2RGB) (smartVidCrop.py:330-331).  ``fr``: frames per second; None = the file's own frame
    duration where it has one (GIF / APNG / WebP ``duration`` in ms), else 25.0.  ``shot_detector`` / ``shots`` as in video_dict."""
    from PIL import Image
    out, rate = [], fr
    if os.path.isdir(path):
        names = sorted(f for f in os.listdir(path) if f.lower().endswith(_IMAGE_EXT))
        if not names:
            raise IOError('no frame images in %r' % (path,))
        for f in names[:max_frames]:
            with Image.open(os.path.join(path, f)) as im:
                out.append(np.asarray(im.convert('RGB'), np.uint8))
    else:
        with Image.open(path) as im:
            n = int(getattr(im, 'n_frames', 1))
            if rate is None and im.info.get('duration'):
                rate = 1000.0 / float(im.info['duration'])
            for i in range(n if max_frames is None else min(n, max_frames)):
                im.seek(i)
                out.append(np.asarray(im.convert('RGB'), np.uint8))
    if any(f.shape != out[0].shape for f in out):
        raise IOError('frames of %r differ in size' % (path,))
    frames = np.ascontiguousarray(np.stack(out))
    trans = list(shot_detector(frames)) if shot_detector is not None else []
    return video_dict(frames, 25.0 if rate is None else rate, trans, shots=shots)


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
