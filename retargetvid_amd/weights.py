"""UNISAL weights: synthetic checkpoints, BN folding and the packed blob for the HIP library.

The reference loads ``weights_best.pth`` (3rd_party_libs/unisal/unisal/model.py:32-33,
train.py:1200-1209); that file is not distributed with the reference
(.MISSING_LARGE_BLOBS), so this module can (a) accept any state-dict that uses the
reference's key layout (the static SALICON slice listed in SURVEY.md §8 A2; extra
``rnn.*`` / other-domain keys are ignored) and (b) build a deterministic synthetic
one from a NumPy seed so that tests, goldens and the benchmark share identical
weights without shipping a 13 MB file.

``fold_state_dict`` folds every eval-mode BatchNorm (eps 1e-5, model.py:65-101,
MobileNetV2.py:10-23) into the preceding convolution and lays tensors out the way
the kernels in ``csrc/`` read them; ``pack_blob`` serialises the result for
``svc_create`` (include/svc.h).
"""
import struct

import numpy as np

BN_EPS = 1e-5
_STAGES = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
           (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]       # MobileNetV2.py:111-120


def backbone_blocks():
    """[(idx, inp, oup, stride, expand)] for cnn.features.1 .. cnn.features.17."""
    blocks, inp, idx = [], 32, 1
    for t, c, n, s in _STAGES:
        for i in range(n):
            blocks.append((idx, inp, c, s if i == 0 else 1, t))
            inp = c
            idx += 1
    return blocks


# --------------------------------------------------------------------------------------
# synthetic checkpoint
# --------------------------------------------------------------------------------------
def _bn(rng, sd, prefix, c, dsbn):
    p = prefix + ('.bn_SALICON' if dsbn else '')
    sd[p + '.weight'] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sd[p + '.bias'] = rng.normal(0, 0.2, c).astype(np.float32)
    sd[p + '.running_mean'] = rng.normal(0, 0.2, c).astype(np.float32)
    sd[p + '.running_var'] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return p


def _bn_identity(sd, p, ch=0, beta=0.0):
    sd[p + '.weight'][ch] = 1.0
    sd[p + '.bias'][ch] = beta
    sd[p + '.running_mean'][ch] = 0.0
    sd[p + '.running_var'][ch] = 1.0


def _conv(rng, sd, key, cout, cin, k, gain=1.0):
    fan = cin * k * k
    sd[key] = (rng.normal(0, 1, (cout, cin, k, k)) * gain * np.sqrt(2.0 / fan)).astype(np.float32)


def _inv_res(rng, sd, prefix, inp, oup, expand, dsbn, carrier_in=None, carrier_out=None,
             residual=False):
    """Random InvertedResidual weights; optionally routes a pass-through 'carrier'
    (input channel carrier_in -> hidden channel 0 -> output channel carrier_out)."""
    hidden = round(inp * expand)
    if expand == 1:
        names = ('.0', '.1', None, None, '.3', '.4')
    else:
        names = ('.3', '.4', '.0', '.1', '.6', '.7')
    dw_w, dw_bn, ex_w, ex_bn, pj_w, pj_bn = names
    if ex_w:
        _conv(rng, sd, prefix + ex_w + '.weight', hidden, inp, 1)
        ex_p = _bn(rng, sd, prefix + ex_bn, hidden, dsbn)
    _conv(rng, sd, prefix + dw_w + '.weight', hidden, 1, 3)
    dw_p = _bn(rng, sd, prefix + dw_bn, hidden, dsbn)
    _conv(rng, sd, prefix + pj_w + '.weight', oup, hidden, 1, gain=0.7)
    pj_p = _bn(rng, sd, prefix + pj_bn, oup, dsbn)
    if carrier_in is None:
        return
    hc = carrier_in if expand == 1 else 0            # hidden channel that carries the signal
    if ex_w:
        w = sd[prefix + ex_w + '.weight']
        w[0] = 0.0
        w[0, carrier_in] = 1.0
        _bn_identity(sd, ex_p)
    w = sd[prefix + dw_w + '.weight']
    w[hc] = 0.0
    w[hc, 0, 1, 1] = 1.0
    _bn_identity(sd, dw_p, hc)
    w = sd[prefix + pj_w + '.weight']
    w[carrier_out] = 0.0
    if not residual:                                  # residual blocks pass x through by themselves
        w[carrier_out, hc] = 1.0
    _bn_identity(sd, pj_p, carrier_out)


def make_synthetic_state_dict(seed=0, carrier=True, gain=0.5, noise=0.02):
    """Deterministic synthetic UNISAL checkpoint (static SALICON slice, reference keys).

    All tensors are random (He-scaled convs, non-trivial BN statistics, a
    non-Gaussian 41x41 smoothing kernel) so that BN folding and every kernel are
    exercised.  With ``carrier=True`` channel 0 additionally carries the input
    luminance through the backbone, Skip-4x and the decoder so that the saliency map
    follows bright blobs of the input: saliency ~ exp(gain * 5 * luminance).  That
    makes thresholded maps look like real ones (a few compact regions) instead of
    salt-and-pepper, which is what the clustering stage's cost depends on.
    """
    rng = np.random.RandomState(seed)
    sd = {}
    c = 0 if carrier else None
    # stem (MobileNetV2.py:124)
    _conv(rng, sd, 'cnn.features.0.0.weight', 32, 3, 3)
    p = _bn(rng, sd, 'cnn.features.0.1', 32, False)
    if carrier:
        std = np.array([0.229, 0.224, 0.225], np.float32)
        sd['cnn.features.0.0.weight'][0] = (5.0 * std / 27.0).reshape(3, 1, 1)
        _bn_identity(sd, p, 0, beta=2.6)
    for idx, inp, oup, stride, expand in backbone_blocks():
        use_c = carrier and idx <= 7
        _inv_res(rng, sd, 'cnn.features.%d.conv' % idx, inp, oup, expand, False,
                 c if use_c else None, c if use_c else None,
                 residual=(stride == 1 and inp == oup))
    _conv(rng, sd, 'cnn.features.18.0.weight', 1280, 320, 1)
    _bn(rng, sd, 'cnn.features.18.1', 1280, False)
    # decoder (model.py:192-246)
    _inv_res(rng, sd, 'post_cnn.inv_res.conv', 1296, 256, 1, False)
    for name, cin, cout in (('skip_2x', 160, 128), ('skip_4x', 64, 64)):
        hid = cin * 2
        _conv(rng, sd, name + '.expansion.0.weight', hid, cin, 1)
        pe = _bn(rng, sd, name + '.expansion.1', hid, True)
        _conv(rng, sd, name + '.reduction.0.weight', cout, hid, 1, gain=0.7)
        sd[name + '.reduction.0.bias'] = rng.normal(0, 0.1, cout).astype(np.float32)
        pr = _bn(rng, sd, name + '.reduction.1', cout, True)
        if carrier and name == 'skip_4x':
            w = sd[name + '.expansion.0.weight']
            w[0] = 0.0
            w[0, 0] = 1.0
            _bn_identity(sd, pe)
            w = sd[name + '.reduction.0.weight']
            w[0] = 0.0
            w[0, 0] = 1.0
            sd[name + '.reduction.0.bias'][0] = 0.0
            _bn_identity(sd, pr)
    _inv_res(rng, sd, 'upsampling_2.inv_res.conv', 384, 128, 2, True)
    _inv_res(rng, sd, 'post_upsampling_2.inv_res.conv', 192, 64, 2, True,
             128 if carrier else None, 0 if carrier else None)
    w = (rng.normal(0, 1, (1, 64, 1, 1)) * (noise if carrier else 0.05)).astype(np.float32)
    if carrier:
        w[0, 0] = gain
    sd['adaptation_salicon.0.weight'] = w
    sd['adaptation_salicon.0.bias'] = np.array([0.1], np.float32)
    # smoothing: Gaussian (model.py:264-272) times a random modulation, normalised
    ax = np.linspace(0, 1, 41)
    g1 = np.exp(-((ax - 0.5) / np.exp(-2.0)) ** 2 / 2)
    k = np.outer(g1, g1) * rng.uniform(0.6, 1.4, (41, 41))
    sd['smoothing_salicon.weight'] = (k / k.sum()).astype(np.float32).reshape(1, 1, 41, 41)
    # Gaussian priors: manual init (model.py:323-331) plus jitter
    mus = ([(a, b) for a in (0.25, 0.5, 0.75) for b in (0.25, 0.5, 0.75)] +
           [(0.5, 0.25), (0.5, 0.5), (0.5, 0.75), (0.25, 0.5), (0.5, 0.5), (0.75, 0.5), (0.5, 0.5)])
    ls = [(-1.5, -1.5)] * 9 + [(0, -1.5)] * 3 + [(-1.5, 0)] * 3 + [(0, 0)]
    g = np.stack([np.array(mus, np.float32), np.array(ls, np.float32)], axis=2)   # [16, y/x, mu/logstd]
    sd['coarse_gaussians_salicon'] = (g + rng.normal(0, 0.02, g.shape)).astype(np.float32)
    return sd


def make_reference_init_state_dict(seed=7, bn_stats=None):
    """A checkpoint initialised the way the reference class initialises itself, from a NumPy seed.

    Same keys and the same DISTRIBUTIONS as a freshly constructed ``UNISAL`` (no carrier channel, no structure):
    backbone and decoder InvertedResidual convs N(0, sqrt(2 / (k*k*out_channels))) (MobileNetV2.py:85-98,:175-188),
    BatchNorm gamma 1 / beta 0, the skip / adaptation convs PyTorch's default kaiming-uniform
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias (model.py:208-246,:257-262), the smoothing kernel the
    normalised Gaussian of model.py:264-272 and the 16 manual Gaussian priors of model.py:323-331.  The values come
    from NumPy (torch's own RNG stream cannot be reproduced without the reference's constructor), so the big
    tensors are reproducible anywhere from the seed.

    ``bn_stats``: {'<bn prefix>.running_mean' / '.running_var': array} -- running statistics calibrated by the
    REFERENCE model itself in the build container (BatchNorm layers in train mode over seeded frames,
    tools/make_golden_unisal.py) and stored with the golden vectors (small); without it the statistics are the
    constructor's (0, 1)."""
    rng = np.random.RandomState(seed)
    sd = {}

    def conv_he(key, cout, cin, k):
        sd[key] = (rng.normal(0, 1, (cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cout))).astype(np.float32)

    def conv_default(key, cout, cin, k, bias):
        bound = 1.0 / np.sqrt(cin * k * k)
        sd[key + '.weight'] = rng.uniform(-bound, bound, (cout, cin, k, k)).astype(np.float32)
        if bias:
            sd[key + '.bias'] = rng.uniform(-bound, bound, cout).astype(np.float32)

    def bn(prefix, c, dsbn):
        p = prefix + ('.bn_SALICON' if dsbn else '')
        sd[p + '.weight'] = np.ones(c, np.float32)
        sd[p + '.bias'] = np.zeros(c, np.float32)
        sd[p + '.running_mean'] = np.zeros(c, np.float32)
        sd[p + '.running_var'] = np.ones(c, np.float32)

    def inv_res(prefix, inp, oup, expand, dsbn):
        hidden = round(inp * expand)
        if expand == 1:
            conv_he(prefix + '.0.weight', hidden, 1, 3); bn(prefix + '.1', hidden, dsbn)
            conv_he(prefix + '.3.weight', oup, hidden, 1); bn(prefix + '.4', oup, dsbn)
        else:
            conv_he(prefix + '.0.weight', hidden, inp, 1); bn(prefix + '.1', hidden, dsbn)
            conv_he(prefix + '.3.weight', hidden, 1, 3); bn(prefix + '.4', hidden, dsbn)
            conv_he(prefix + '.6.weight', oup, hidden, 1); bn(prefix + '.7', oup, dsbn)

    conv_he('cnn.features.0.0.weight', 32, 3, 3)
    bn('cnn.features.0.1', 32, False)
    for idx, inp, oup, stride, expand in backbone_blocks():
        inv_res('cnn.features.%d.conv' % idx, inp, oup, expand, False)
    conv_he('cnn.features.18.0.weight', 1280, 320, 1)
    bn('cnn.features.18.1', 1280, False)
    inv_res('post_cnn.inv_res.conv', 1296, 256, 1, False)      # plain BatchNorm (model.py:192-200)
    for name, cin, cout in (('skip_2x', 160, 128), ('skip_4x', 64, 64)):
        conv_default(name + '.expansion.0', cin * 2, cin, 1, False)
        bn(name + '.expansion.1', cin * 2, True)
        conv_default(name + '.reduction.0', cout, cin * 2, 1, True)
        bn(name + '.reduction.1', cout, True)
    inv_res('upsampling_2.inv_res.conv', 384, 128, 2, True)
    inv_res('post_upsampling_2.inv_res.conv', 192, 64, 2, True)
    conv_default('adaptation_salicon.0', 1, 64, 1, True)
    ax = np.linspace(0, 1, 41)
    g1 = np.exp(-((ax - 0.5) / np.exp(-2.0)) ** 2 / 2)
    k = np.outer(g1, g1)
    sd['smoothing_salicon.weight'] = (k / k.sum()).astype(np.float32).reshape(1, 1, 41, 41)
    mus = ([(a, b) for a in (0.25, 0.5, 0.75) for b in (0.25, 0.5, 0.75)] +
           [(0.5, 0.25), (0.5, 0.5), (0.5, 0.75), (0.25, 0.5), (0.5, 0.5), (0.75, 0.5), (0.5, 0.5)])
    ls = [(-1.5, -1.5)] * 9 + [(0, -1.5)] * 3 + [(-1.5, 0)] * 3 + [(0, 0)]
    sd['coarse_gaussians_salicon'] = np.stack([np.array(mus, np.float32), np.array(ls, np.float32)], axis=2)
    if bn_stats is not None:
        for kname, v in bn_stats.items():
            if kname not in sd:
                raise KeyError('bn_stats key %s is not a BatchNorm statistic of the static SALICON slice' % kname)
            sd[kname] = np.asarray(v, np.float32)
    return sd


def make_trained_like_state_dict(golden_dir, variant=1):
    """The TRAINED-LIKE checkpoint: the reference-initialised one (seed 7, BatchNorm statistics calibrated by the reference
    model: tests/golden/unisal_golden2.npz) with its last decoder stage (skip_4x, post_upsampling_2, adaptation_salicon,
    smoothing_salicon) replaced by tensors the REFERENCE model was fitted to in the build container
    (tools/make_golden_unisal3.py -> tests/golden/unisal_golden3.npz, keys tl/<name>): peaky maps like a trained saliency
    network's -- ~440 points above the threshold, ~7 pixels per grey level next to it (the luminance-carrier checkpoint: ~45,
    the reference-initialised one: ~500).  variant=2: a second fit (`tl2`, tests/golden/unisal_golden4.npz: another seed, skip_2x and
    upsampling_2 fitted as well, narrower targets), so that the parity claims on peaky maps do not rest on one checkpoint."""
    import os
    name = 'unisal_golden3.npz' if variant == 1 else 'unisal_golden4.npz'
    g2 = np.load(os.path.join(golden_dir, 'unisal_golden2.npz'))
    g3 = np.load(os.path.join(golden_dir, name))
    sd = make_reference_init_state_dict(7, {k[3:]: g2[k] for k in g2.files if k.startswith('bn/')})
    n = 0
    for k in g3.files:
        if k.startswith('tl/'):
            if k[3:] not in sd or sd[k[3:]].shape != g3[k].shape:
                raise KeyError('%s: %s is not a tensor of the static SALICON slice' % (name, k))
            sd[k[3:]] = np.asarray(g3[k], np.float32)
            n += 1
    if n == 0:
        raise KeyError('%s holds no tl/ tensors' % name)
    return sd


def to_numpy_state_dict(sd):
    """Accept a torch or numpy state-dict (e.g. torch.load('weights_best.pth'))."""
    out = {}
    for k, v in sd.items():
        if hasattr(v, 'detach'):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


# --------------------------------------------------------------------------------------
# folding
# --------------------------------------------------------------------------------------
def _bn_params(sd, prefix):
    if prefix + '.weight' not in sd:
        prefix += '.bn_SALICON'                      # DSBN: model.py:100-101 with source SALICON
    g = sd[prefix + '.weight'].astype(np.float64)
    b = sd[prefix + '.bias'].astype(np.float64)
    m = sd[prefix + '.running_mean'].astype(np.float64)
    v = sd[prefix + '.running_var'].astype(np.float64)
    s = g / np.sqrt(v + BN_EPS)
    return s, b - m * s


def _fold(sd, wkey, bnprefix, bias_key=None):
    """conv (+bias) followed by eval BN -> (W', b') with W'[o] = W[o]*s[o]."""
    w = sd[wkey].astype(np.float64)
    s, t = _bn_params(sd, bnprefix)
    wf = w * s.reshape(-1, 1, 1, 1)
    bf = t.copy()
    if bias_key is not None:
        bf = bf + sd[bias_key].astype(np.float64) * s
    return wf, bf


def _pw(wf, bf, relu6, name):
    cout, cin = wf.shape[:2]
    return dict(kind='pw', name=name, cin=cin, cout=cout, relu6=relu6,
                w=wf.reshape(cout, cin).astype(np.float32), b=bf.astype(np.float32))


def _dw(wf, bf, stride, name):
    c = wf.shape[0]
    # kernel layout [9][C]: tap-major so a thread's float4 over channels is contiguous
    return dict(kind='dw', name=name, c=c, stride=stride, relu6=True,
                w=np.ascontiguousarray(wf.reshape(c, 9).T).astype(np.float32),
                b=bf.astype(np.float32))


def _fold_inv_res(sd, prefix, expand, stride, name):
    layers = []
    if expand == 1:
        layers.append(_dw(*_fold(sd, prefix + '.0.weight', prefix + '.1'), stride, name + '.dw'))
        layers.append(_pw(*_fold(sd, prefix + '.3.weight', prefix + '.4'), False, name + '.project'))
    else:
        layers.append(_pw(*_fold(sd, prefix + '.0.weight', prefix + '.1'), True, name + '.expand'))
        layers.append(_dw(*_fold(sd, prefix + '.3.weight', prefix + '.4'), stride, name + '.dw'))
        layers.append(_pw(*_fold(sd, prefix + '.6.weight', prefix + '.7'), False, name + '.project'))
    return layers


def gaussian_maps_np(gaussians, h, w, scaling=6.0):
    """model.py:348-378 in float32 NumPy -> [16,h,w]."""
    ys = np.linspace(0, 1, h, dtype=np.float32)
    xs = np.linspace(0, 1, w, dtype=np.float32)
    out = np.empty((gaussians.shape[0], h, w), np.float32)
    for i, g in enumerate(gaussians.astype(np.float32)):
        my = np.exp(-((ys - g[0, 0]) / np.exp(g[0, 1])) ** 2 / 2).astype(np.float32)
        mx = np.exp(-((xs - g[1, 0]) / np.exp(g[1, 1])) ** 2 / 2).astype(np.float32)
        out[i] = np.outer(my, mx) * np.float32(scaling)
    return out


def smoothing_phase_table(k41):
    """Rewrite nearest-x8 upsample -> replicate-pad 20 -> 41x41 conv (model.py:485-492)
    as 64 phase kernels of 7x7 taps over the low-res map.

    out[y,x] = sum_{dy,dx} k[dy,dx] * L[clamp((y+dy-20) // 8), clamp((x+dx-20) // 8)];
    with y = 8*cy + py the low-res row offset is floor((py+dy-20)/8) in [-3, 3].
    Returns float32 [8(py)][8(px)][7][7] (row offset -3..3, col offset -3..3)."""
    k = k41.reshape(41, 41).astype(np.float64)
    sel = np.zeros((8, 7, 41))
    for p in range(8):
        for d in range(41):
            sel[p, (p + d - 20) // 8 + 3, d] = 1.0
    return np.einsum('pad,qbe,de->pqab', sel, sel, k).astype(np.float32)


def fold_state_dict(sd):
    """-> ordered list of layer dicts in execution order (see csrc/svc_net.cpp)."""
    sd = to_numpy_state_dict(sd)
    L = []
    wf, bf = _fold(sd, 'cnn.features.0.0.weight', 'cnn.features.0.1')
    # stem layout [ky][kx][cin][cout]
    L.append(dict(kind='stem', name='stem', w=np.ascontiguousarray(wf.transpose(2, 3, 1, 0)).astype(np.float32),
                  b=bf.astype(np.float32)))
    for idx, inp, oup, stride, expand in backbone_blocks():
        L += _fold_inv_res(sd, 'cnn.features.%d.conv' % idx, expand, stride, 'f%d' % idx)
    L.append(_pw(*_fold(sd, 'cnn.features.18.0.weight', 'cnn.features.18.1'), True, 'f18'))
    for name in ('skip_2x', 'skip_4x'):
        L.append(_pw(*_fold(sd, name + '.expansion.0.weight', name + '.expansion.1'), True, name + '.expand'))
        L.append(_pw(*_fold(sd, name + '.reduction.0.weight', name + '.reduction.1',
                            name + '.reduction.0.bias'), False, name + '.reduce'))
    # raw Gaussian parameters [16][y/x][mu/logstd]; the library evaluates the prior maps
    # (model.py:348-378) for whatever feature size the input aspect ratio selects
    L.append(dict(kind='const', name='gaussians',
                  w=sd['coarse_gaussians_salicon'].astype(np.float32).reshape(64)))
    L += _fold_inv_res(sd, 'post_cnn.inv_res.conv', 1, 1, 'post_cnn')
    L += _fold_inv_res(sd, 'upsampling_2.inv_res.conv', 2, 1, 'us2')
    L += _fold_inv_res(sd, 'post_upsampling_2.inv_res.conv', 2, 1, 'post_us2')
    L.append(dict(kind='adapt', name='adapt',
                  w=sd['adaptation_salicon.0.weight'].reshape(64).astype(np.float32),
                  b=sd['adaptation_salicon.0.bias'].reshape(1).astype(np.float32)))
    L.append(dict(kind='const', name='smooth_phase', w=smoothing_phase_table(sd['smoothing_salicon.weight'])))
    return L


# --------------------------------------------------------------------------------------
# blob
# --------------------------------------------------------------------------------------
BLOB_MAGIC = 0x53564331          # 'SVC1'


def pack_blob(layers):
    """Serialise folded layers: header {magic, n_tensors}, then per tensor
    {offset_floats, n_floats} (u64 each), then the fp32 payload (each tensor 64-B
    aligned).  Tensor order is fixed: for every layer its ``w`` then (if any) ``b``;
    csrc/svc_net.cpp walks the same order."""
    tensors = []
    for l in layers:
        w = np.ascontiguousarray(l['w'], np.float32)
        if l['kind'] == 'pw':                       # rows padded to a multiple of the 32-wide MFMA tile
            pad = (-w.shape[0]) % 32
            if pad:
                w = np.concatenate([w, np.zeros((pad, w.shape[1]), np.float32)])
        tensors.append(w.ravel())
        if 'b' in l:
            tensors.append(np.ascontiguousarray(l['b'], np.float32).ravel())
    head = 16 + 16 * len(tensors)
    head = (head + 63) // 64 * 64
    offs, cur = [], 0
    for t in tensors:
        offs.append(cur)
        cur += (t.size + 15) // 16 * 16
    out = bytearray(head + cur * 4)
    struct.pack_into('<QQ', out, 0, BLOB_MAGIC, len(tensors))
    for i, (t, o) in enumerate(zip(tensors, offs)):
        struct.pack_into('<QQ', out, 16 + 16 * i, head // 4 + o, t.size)
        out[head + o * 4: head + o * 4 + t.size * 4] = t.tobytes()
    return bytes(out)


# ------------------------------------------------------------------------------------------------------
# TransNet V1 (shot boundaries): variable names and layouts of the reference's TensorFlow graph
# (3rd_party_libs/transnetv1/transnetv1_handler.py:25-84); no pre-trained checkpoint ships with the reference
# (note.txt:1), so the tests and the bench use this seeded initialisation.
# ------------------------------------------------------------------------------------------------------
TRANSNET_F, TRANSNET_L, TRANSNET_S, TRANSNET_D = 16, 3, 2, 256
TRANSNET_DILATIONS = (1, 2, 4, 8)


def transnet_cells():
    """(block, cell, cin, filters) of the six DDCNN cells."""
    cells, cin = [], 3
    for b in range(TRANSNET_L):
        f = TRANSNET_F << b
        for c in range(TRANSNET_S):
            cells.append((b, c, cin, f))
            cin = 4 * f
    return cells


def make_transnet_state_dict(seed=0):
    """TensorFlow-layout float32 arrays: conv kernels [3, 3, 3, cin, f] (He-scaled so activations stay O(1) through the
    six cells), biases, Dense(256) on the 3 x 6 x 256 flattened map, Dense(2)."""
    rng = np.random.RandomState(seed)
    sd = {}
    for b, c, cin, f in transnet_cells():
        for d in TRANSNET_DILATIONS:
            p = 'TransNet/SDDCNN_%d/DDCNN_%d/Conv3D_%d' % (b + 1, c + 1, d)
            sd[p + '/kernel'] = (rng.randn(3, 3, 3, cin, f) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
            sd[p + '/bias'] = (rng.randn(f) * 0.05).astype(np.float32)
    nflat = 3 * 6 * 4 * (TRANSNET_F << (TRANSNET_L - 1))
    sd['TransNet/dense/kernel'] = (rng.randn(nflat, TRANSNET_D) * np.sqrt(2.0 / nflat)).astype(np.float32)
    sd['TransNet/dense/bias'] = (rng.randn(TRANSNET_D) * 0.05).astype(np.float32)
    sd['TransNet/dense_1/kernel'] = (rng.randn(TRANSNET_D, 2) * np.sqrt(4.0 / TRANSNET_D)).astype(np.float32)
    sd['TransNet/dense_1/bias'] = (rng.randn(2) * 0.5).astype(np.float32)
    return sd


def transnet_from_tf_variables(variables):
    """A TensorFlow checkpoint's variables -> the state dict make_transnet_state_dict / pack_transnet_blob use.

    `variables`: {name: array} as exported from the reference's graph (3rd_party_libs/transnetv1/transnetv1_handler.py:25-91),
    e.g. ``{v.name: sess.run(v) for v in tf.trainable_variables()}`` or ``tf.train.load_checkpoint(path)`` read variable
    by variable.  Accepted spellings: with or without the ``:0`` tensor suffix; optimiser slots (``.../Adam``, ``.../Adam_1``,
    ``beta1_power``, ``beta2_power``, ``global_step``) and anything outside the ``TransNet/`` scope are ignored.
    Layouts are TensorFlow's own: Conv3D kernels [kt, kh, kw, cin, filters], Dense kernels [in, out].  Raises KeyError
    for a missing variable and ValueError for a wrong shape, naming it."""
    clean = {}
    for name, arr in variables.items():
        n = str(name)
        if n.endswith(':0'):
            n = n[:-2]
        if not n.startswith('TransNet/') or n.rsplit('/', 1)[-1] not in ('kernel', 'bias'):
            continue                                   # optimiser slots, counters, other scopes
        clean[n] = np.asarray(arr, np.float32)
    want = {}
    for b, c, cin, f in transnet_cells():
        for d in TRANSNET_DILATIONS:
            p = 'TransNet/SDDCNN_%d/DDCNN_%d/Conv3D_%d' % (b + 1, c + 1, d)
            want[p + '/kernel'] = (3, 3, 3, cin, f)
            want[p + '/bias'] = (f,)
    nflat = 3 * 6 * 4 * (TRANSNET_F << (TRANSNET_L - 1))
    want['TransNet/dense/kernel'] = (nflat, TRANSNET_D)
    want['TransNet/dense/bias'] = (TRANSNET_D,)
    want['TransNet/dense_1/kernel'] = (TRANSNET_D, 2)
    want['TransNet/dense_1/bias'] = (2,)
    sd = {}
    for name, shape in want.items():
        if name not in clean:
            raise KeyError('TransNet checkpoint: variable %r is missing (have %d TransNet variables)' % (name, len(clean)))
        if tuple(clean[name].shape) != shape:
            raise ValueError('TransNet checkpoint: %r has shape %r, the F16 L3 S2 D256 graph needs %r'
                             % (name, tuple(clean[name].shape), shape))
        sd[name] = np.ascontiguousarray(clean[name])
    return sd


def pack_transnet_blob(sd):
    """The device layout (csrc/svc_shot.hip): per cell, per dilation a GEMM weight [rows = out channel, padded to a multiple
    of 32][27 taps x cpad] (cpad = max(4, cin): the three input channels get a zero fourth), then the cell's 4f biases;
    Dense(256) transposed to [256][4608] + bias; Dense(2) transposed to [2][256] + bias.  One flat float32 array."""
    parts = []
    for b, c, cin, f in transnet_cells():
        cpad, rows = max(4, cin), (f + 31) // 32 * 32
        kpad = (27 * cpad + 7) // 8 * 8                                        # rows padded to whole 8-deep MFMA k steps
        for d in TRANSNET_DILATIONS:
            k = np.asarray(sd['TransNet/SDDCNN_%d/DDCNN_%d/Conv3D_%d/kernel' % (b + 1, c + 1, d)], np.float32)
            w = np.zeros((rows, 27, cpad), np.float32)
            w[:f, :, :cin] = k.reshape(27, cin, f).transpose(2, 0, 1)          # tap = (kt * 3 + kh) * 3 + kw
            wp = np.zeros((rows, kpad), np.float32)
            wp[:, :27 * cpad] = w.reshape(rows, -1)
            parts.append(wp.reshape(-1))
        parts.append(np.concatenate([np.asarray(sd['TransNet/SDDCNN_%d/DDCNN_%d/Conv3D_%d/bias' % (b + 1, c + 1, d)], np.float32)
                                     for d in TRANSNET_DILATIONS]))
    parts.append(np.ascontiguousarray(np.asarray(sd['TransNet/dense/kernel'], np.float32).T).reshape(-1))
    parts.append(np.asarray(sd['TransNet/dense/bias'], np.float32))
    parts.append(np.ascontiguousarray(np.asarray(sd['TransNet/dense_1/kernel'], np.float32).T).reshape(-1))
    parts.append(np.asarray(sd['TransNet/dense_1/bias'], np.float32))
    return np.concatenate(parts)
