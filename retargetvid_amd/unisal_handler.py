"""Counterpart of the reference's saliency operator boundary
(3rd_party_libs/unisal/unisal_handler.py:68-71, :85-86).

  trainer = init_unisal_for_images()                      # builds the device engine
  smaps = predictions_from_memory_nuint8_np(trainer, images, [], '')

``images`` is the reference's ``frames[:n]`` buffer: uint8 [n,H,W,3] RGB at saliency size;
the result is uint8 [H,W,n] with the frame index as the fastest axis, exactly what
smartVidCrop.py:420-421 stores into ``vid_data['smaps'][:, :, si:ei]``.  The whole batch is
one device pass (the reference loops one image at a time, train.py:1265-1266)."""
import numpy as np


def init_unisal_for_images(state_dict=None, seed=0):
    """state_dict: a UNISAL checkpoint in the reference's key layout (e.g.
    torch.load('weights_best.pth')); None builds the deterministic synthetic one."""
    from . import ops
    return ops.Engine(state_dict, seed=seed)


def predictions_from_memory_nuint8_np(trainer, images, out_names=(), out_dir=''):
    import torch
    if len(out_names) > 0:
        raise NotImplementedError('writing prediction images (cv2.imwrite, train.py:1276-1278) is not part of this path')
    images = np.ascontiguousarray(np.asarray(images), dtype=np.uint8)
    if images.ndim != 4 or images.shape[3] != 3:
        raise ValueError('images must be uint8 [n,H,W,3]')
    if images.shape[0] == 0:
        raise IndexError('empty image batch')           # the reference fails on images[0]
    maps = trainer.saliency(torch.from_numpy(images).to(trainer.device))
    return np.ascontiguousarray(maps.permute(1, 2, 0).cpu().numpy())
