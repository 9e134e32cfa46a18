"""CPU restatement of TransNet V1 (shot-boundary network) as the reference builds it -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and tools/ only; the product path is retargetvid_amd/transnetv1_handler.py -> libsvc_hip.so).

PARITY UNPINNED: the reference runs this network in TensorFlow 1.x (3rd_party_libs/transnetv1/transnetv1_handler.py:17-130)
and neither TensorFlow nor the pre-trained checkpoint (note.txt:1) is available in the build container, so no output of the
reference itself could be recorded.  What is restated here is the graph the reference constructs, layer by layer:

  _build (:25-84)      uint8 [B, T, 27, 48, 3] / 255 -> L = 3 SDDCNN blocks of S = 2 DDCNN cells; a cell = four
                       Conv3D(filters, kernel 3x3x3, dilation (d, 1, 1), d = 1, 2, 4, 8, padding SAME, bias, ReLU) on the SAME
                       input, concatenated on the channel axis (:33-36, :55-59); filters = 16 * 2**block (:47);
                       MaxPool3D (1, 2, 2) after each block (:64); flatten (h, w, c) per frame (:68-69); Dense 256 ReLU
                       (:72); Dense 2 (:76); softmax, class 1 (:79)
  predict_video (:100-130)  windows of 100 frames, stride 50, the middle 50 kept; 25 copies of the first frame in front,
                       25 + 50 - (n % 50 or 50) copies of the last frame behind
Weights are TensorFlow-layout arrays: conv kernels [kt, kh, kw, cin, cout], dense kernels [in, out]
(retargetvid_amd/weights.make_transnet_state_dict names them after the reference's variable scopes)."""
import numpy as np
import torch
import torch.nn.functional as F

F0, L, S, D = 16, 3, 2, 256
H, W = 27, 48
DILATIONS = (1, 2, 4, 8)


def conv_name(block, cell, d):
    return 'TransNet/SDDCNN_%d/DDCNN_%d/Conv3D_%d' % (block + 1, cell + 1, d)


def forward(sd, frames_u8):
    """frames_u8 [B, T, 27, 48, 3] uint8 -> P(transition) float32 [B, T]  (ShotTransNet.predict_raw, :93-97)."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(torch.float32) / 255.0
    x = x.permute(0, 4, 1, 2, 3).contiguous()                         # NDHWC -> NCDHW
    with torch.no_grad():
        for b in range(L):
            for c in range(S):
                outs = []
                for d in DILATIONS:
                    k = torch.from_numpy(sd[conv_name(b, c, d) + '/kernel']).permute(4, 3, 0, 1, 2).contiguous()
                    bias = torch.from_numpy(sd[conv_name(b, c, d) + '/bias'])
                    # kernel 3, dilation (d, 1, 1), SAME: symmetric zero padding of (d, 1, 1)
                    outs.append(F.relu(F.conv3d(x, k, bias, padding=(d, 1, 1), dilation=(d, 1, 1))))
                x = torch.cat(outs, 1)
            x = F.max_pool3d(x, (1, 2, 2))                            # VALID: floor
        Bn, C, T, h, w = x.shape
        x = x.permute(0, 2, 3, 4, 1).reshape(Bn, T, h * w * C)        # flatten (h, w, c) per frame
        x = F.relu(x @ torch.from_numpy(sd['TransNet/dense/kernel']) + torch.from_numpy(sd['TransNet/dense/bias']))
        logits = x @ torch.from_numpy(sd['TransNet/dense_1/kernel']) + torch.from_numpy(sd['TransNet/dense_1/bias'])
        return torch.softmax(logits, -1)[:, :, 1].numpy()


def window_indices(n):
    """Frame index of every slot of every 100-frame window of predict_video (:104-121) for a video of n frames."""
    pad_end = 25 + 50 - (n % 50 if n % 50 != 0 else 50)
    idx = np.concatenate([np.zeros(25, np.int64), np.arange(n), np.full(pad_end, n - 1, np.int64)])
    wins, ptr = [], 0
    while ptr + 100 <= len(idx):
        wins.append(idx[ptr:ptr + 100])
        ptr += 50
    return np.stack(wins)


def predict_video(sd, frames_u8, batch=4):
    """[n, 27, 48, 3] uint8 -> [n] float32 (:100-130)."""
    n = len(frames_u8)
    wi = window_indices(n)
    res = []
    for i in range(0, len(wi), batch):
        p = forward(sd, frames_u8[wi[i:i + batch]])
        res.append(p[:, 25:75].reshape(-1))
    return np.concatenate(res)[:n]


def predictions_to_scenes(predictions, threshold=0.5):
    """smartVidCrop.py:214-230 (same walk as transnet_utils.scenes_from_predictions + the all-ones fix)."""
    pred = (np.asarray(predictions) > threshold).astype(np.uint8)
    scenes, t, t_prev, start, i = [], -1, 0, 0, 0
    for i, t in enumerate(pred):
        if t_prev == 1 and t == 0:
            start = i
        if t_prev == 0 and t == 1 and i != 0:
            scenes.append([start, i])
        t_prev = t
    if t == 0:
        scenes.append([start, i])
    if len(scenes) == 0:
        return np.array([[0, len(pred) - 1]], dtype=np.int32)
    return np.array(scenes, dtype=np.int32)
