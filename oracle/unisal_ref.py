"""Oracle: UNISAL static (SALICON-domain) saliency forward on the CPU, fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  A functional PyTorch-CPU
restatement of the reference network, driven directly by a state-dict that
uses the reference's key layout, so the real ``weights_best.pth`` could be
dropped in.  BN is applied as a separate eval-mode op (NOT folded) so that
the product's folded weights are checked against an independent formulation.

Follows (reference file:line, relative to /root/reference):
  3rd_party_libs/unisal/unisal/models/MobileNetV2.py:10-23   conv_bn / conv_1x1_bn
  3rd_party_libs/unisal/unisal/models/MobileNetV2.py:26-83   InvertedResidual
  3rd_party_libs/unisal/unisal/models/MobileNetV2.py:111-173 stage table, omit-stride
                                                             sub-sampling, feature taps
  3rd_party_libs/unisal/unisal/model.py:348-378              Gaussian prior maps
  3rd_party_libs/unisal/unisal/model.py:388-409              skip connections
  3rd_party_libs/unisal/unisal/model.py:411-506              forward (static, bypass RNN)
  3rd_party_libs/unisal/unisal/utils.py:132-136              log_softmax over all pixels
  3rd_party_libs/unisal/unisal/train.py:1255-1279            exp, /amax, *255, uint8 cast
  3rd_party_libs/unisal/unisal/data.py:1086-1103             get_optimal_out_size
  3rd_party_libs/unisal/unisal/data.py:1281-1302             preprocess (LANCZOS, /255, normalise)

Pinned: tests/golden/unisal_*.npz hold outputs of the reference's own
model.py / MobileNetV2.py imported in the build container (script:
tools/make_golden_unisal.py); tests/test_oracle_unisal.py checks this module
against them.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import lanczos_ref

BN_EPS = 1e-5
RGB_MEAN = (0.485, 0.456, 0.406)
RGB_STD = (0.229, 0.224, 0.225)
SOURCE = 'SALICON'

# MobileNetV2.py:111-120  (t, c, n, s)
_STAGES = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
           (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def backbone_blocks():
    """[(idx, inp, oup, stride, expand)] for features.1 .. features.17."""
    blocks = []
    inp = 32
    idx = 1
    for t, c, n, s in _STAGES:
        for i in range(n):
            blocks.append((idx, inp, c, s if i == 0 else 1, t))
            inp = c
            idx += 1
    return blocks


def get_optimal_out_size(img_size):
    """data.py:1086-1103 — network input (H, W) for an image of (h, w)."""
    ar = img_size[0] / img_size[1]
    best, best_size = -1.0, None
    for n1 in range(7, 14):
        for n2 in range(7, 14):
            if 100 <= n1 * n2 <= 120:
                this_ar = n1 / n2
                ratio = min(ar, this_ar) / max(ar, this_ar)
                if ratio > best:          # np.argmax keeps the first maximum
                    best, best_size = ratio, (n1, n2)
    return best_size[0] * 32, best_size[1] * 32


def preprocess(img_u8, out_size=None):
    """data.py:1281-1294: HWC u8 RGB -> LANCZOS resize -> /255 -> normalise -> CHW fp32."""
    if out_size is None:
        out_size = get_optimal_out_size(img_u8.shape[:2])
    r = lanczos_ref.resize_lanczos_u8(img_u8, out_size[0], out_size[1])
    t = torch.from_numpy(r).permute(2, 0, 1).contiguous()
    # torchvision ToTensor: byte -> float32 then div(255); Normalize: sub mean, div std
    t = t.to(torch.float32).div(255)
    mean = torch.tensor(RGB_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(RGB_STD, dtype=torch.float32).view(3, 1, 1)
    return (t - mean) / std


class _SD:
    def __init__(self, sd):
        self.sd = {k: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)))
                   for k, v in sd.items()}

    def __call__(self, key):
        return self.sd[key].to(torch.float32)

    def bn(self, x, prefix):
        """Eval-mode BatchNorm2d; resolves the SALICON branch of a DSBN module."""
        if prefix + '.weight' not in self.sd:
            prefix = prefix + '.bn_' + SOURCE        # model.py:100-101
        return F.batch_norm(x, self(prefix + '.running_mean'), self(prefix + '.running_var'),
                            self(prefix + '.weight'), self(prefix + '.bias'),
                            False, 0.0, BN_EPS)


def _relu6(x):
    return torch.clamp(x, 0.0, 6.0)


def _inverted_residual(P, x, prefix, inp, oup, stride, expand, omit_stride):
    """MobileNetV2.py:26-83.  Omit-stride blocks run their depthwise conv at stride 1."""
    actual_stride = 1 if omit_stride else stride
    use_res = stride == 1 and inp == oup
    hidden = round(inp * expand)
    y = x
    if expand == 1:
        y = F.conv2d(y, P(prefix + '.0.weight'), None, actual_stride, 1, 1, hidden)
        y = _relu6(P.bn(y, prefix + '.1'))
        y = F.conv2d(y, P(prefix + '.3.weight'))
        y = P.bn(y, prefix + '.4')
    else:
        y = F.conv2d(y, P(prefix + '.0.weight'))
        y = _relu6(P.bn(y, prefix + '.1'))
        y = F.conv2d(y, P(prefix + '.3.weight'), None, actual_stride, 1, 1, hidden)
        y = _relu6(P.bn(y, prefix + '.4'))
        y = F.conv2d(y, P(prefix + '.6.weight'))
        y = P.bn(y, prefix + '.7')
    return x + y if use_res else y


def backbone(P, x):
    """MobileNetV2.py:161-173 -> (feat_1x[1280], feat_2x[160], feat_4x[64])."""
    x = F.conv2d(x, P('cnn.features.0.0.weight'), None, 2, 1)
    x = _relu6(P.bn(x, 'cnn.features.0.1'))
    feat_2x = feat_4x = None
    for idx, inp, oup, stride, expand in backbone_blocks():
        # every first block of a stage is built with omit_stride=True (MobileNetV2.py:130-132)
        x = _inverted_residual(P, x, 'cnn.features.%d.conv' % idx, inp, oup, stride, expand, True)
        if idx == 7:
            feat_4x = x.clone()
        elif idx == 14:
            feat_2x = x.clone()
        if stride != 1:                                  # MobileNetV2.py:170-171
            x = x[..., ::2, ::2]
    x = F.conv2d(x, P('cnn.features.18.0.weight'))
    x = _relu6(P.bn(x, 'cnn.features.18.1'))
    return x, feat_2x, feat_4x


def gaussian_maps(gaussians, h, w, scaling=6.0):
    """model.py:348-378: 16 separable Gaussian priors on a linspace(0,1) grid -> [16,h,w]."""
    ys = torch.linspace(0, 1, h, dtype=torch.float32)
    xs = torch.linspace(0, 1, w, dtype=torch.float32)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    maps = []
    for g in torch.unbind(gaussians.to(torch.float32)):
        m = torch.ones(h, w, dtype=torch.float32)
        for mu_logstd, grid in zip(g, (gy, gx)):
            mu = mu_logstd[0]
            std = torch.exp(mu_logstd[1])
            m = m * torch.exp(-((grid - mu) / std) ** 2 / 2)
        maps.append(m * scaling)
    return torch.stack(maps)


def _skip(P, x, name):
    """model.py:388-409: 1x1+BN+ReLU6, (dropout = identity), 1x1(+bias)+BN."""
    y = F.conv2d(x, P(name + '.expansion.0.weight'))
    y = _relu6(P.bn(y, name + '.expansion.1'))
    y = F.conv2d(y, P(name + '.reduction.0.weight'), P(name + '.reduction.0.bias'))
    return P.bn(y, name + '.reduction.1')


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def forward_logits(sd, x, target_size, taps=None):
    """model.py:411-497 for T=1, static=True, source='SALICON'.

    x: [B,3,H,W] fp32 normalised.  Returns the bilinear-resized pre-softmax map
    [B, th, tw] (log_softmax is applied by the caller).  ``taps`` (dict) receives
    intermediate tensors for golden comparison.
    """
    P = _SD(sd)
    with torch.no_grad():
        f1, f2, f4 = backbone(P, x)
        s2 = _skip(P, f2, 'skip_2x')
        s4 = _skip(P, f4, 'skip_4x')
        g = gaussian_maps(P('coarse_gaussians_salicon'), f1.shape[-2], f1.shape[-1])
        f = torch.cat((f1, g.unsqueeze(0).expand(f1.shape[0], -1, -1, -1)), dim=1)
        f = _inverted_residual(P, f, 'post_cnn.inv_res.conv', 1296, 256, 1, 1, False)
        post_cnn = f
        f = _up2(f)
        f = torch.cat((f, s2), dim=1)
        f = _inverted_residual(P, f, 'upsampling_2.inv_res.conv', 384, 128, 1, 2, False)
        f = _up2(f)
        f = torch.cat((f, s4), dim=1)
        f = _inverted_residual(P, f, 'post_upsampling_2.inv_res.conv', 192, 64, 1, 2, False)
        a = F.conv2d(f, P('adaptation_salicon.0.weight'), P('adaptation_salicon.0.bias'))
        y = F.interpolate(a, size=x.shape[-2:], mode='nearest')
        y = F.pad(y, [20] * 4, mode='replicate')
        y = F.conv2d(y, P('smoothing_salicon.weight'))
        y = F.interpolate(y, size=tuple(target_size), mode='bilinear', align_corners=False)
        if taps is not None:
            taps.update(feat_1x=f1, feat_2x=f2, feat_4x=f4, skip_2x=s2, skip_4x=s4,
                        post_cnn=post_cnn, dec=f, adapt=a[:, 0])
        return y[:, 0]


def quantise_u8(pre):
    """utils.py:132-136 + train.py:1267-1274: log-softmax over all pixels, exp,
    divide by the max, *255.0, truncate to u8.  pre: [B,h,w] fp32 -> [B,h,w] u8."""
    b = pre.shape[0]
    ls = F.log_softmax(pre.reshape(b, -1), dim=1).reshape(pre.shape)
    p = ls.exp().numpy()
    out = np.empty(p.shape, np.uint8)
    for i in range(b):
        s = (p[i] / np.amax(p[i])) * 255.0
        out[i] = s.astype('uint8')
    return out


def saliency_u8(sd, images_u8, taps=None):
    """Oracle of predictions_from_memory_nuint8_np (unisal_handler.py:85-86,
    train.py:1255-1279).  images_u8: [n,H,W,3] u8 RGB -> [H,W,n] u8.
    Frames are processed one at a time like the reference."""
    n, h, w = images_u8.shape[:3]
    out = np.zeros((h, w, n), np.uint8)
    for i in range(n):
        x = preprocess(images_u8[i]).unsqueeze(0)
        t = {} if taps is not None else None
        pre = forward_logits(sd, x, (h, w), t)
        out[:, :, i] = quantise_u8(pre)[0]
        if taps is not None:
            t['input'] = x
            t['pre'] = pre
            taps.setdefault('frames', []).append(t)
    return out
