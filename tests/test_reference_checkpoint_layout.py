"""Build-container-only check (skipped wherever /root/reference is absent, e.g. on the GPU box):
a state-dict produced by the REFERENCE model class itself — all 834 entries, including rnn.*,
post_rnn.* and the three other domains — goes through the product's loader/folder, and the oracle
driven by that same dict reproduces the reference forward.  This is what dropping in the real
weights_best.pth exercises."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/3rd_party_libs/unisal/unisal'),
                                reason='reference tree not present')


def test_reference_state_dict_loads_and_matches():
    sys.path.insert(0, ROOT)
    from tools.ref_import import load_reference_unisal
    from oracle import unisal_ref as U
    from retargetvid_amd import weights
    torch.manual_seed(3)
    torch.set_num_threads(1)
    net, _ = load_reference_unisal()
    for m in net.modules():                                  # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.7, 1.3)
    sd = net.state_dict()
    assert len(sd) == 834 and any(k.startswith('rnn.') for k in sd)
    layers = weights.fold_state_dict(sd)                     # torch tensors, extra keys ignored
    blob = weights.pack_blob(layers)
    assert len(layers) == 67 and len(blob) > 12_000_000         # stem + 50 backbone + f18 + 4 skip + gaussians + 8 decoder + adapt + smoothing
    x = torch.randn(1, 3, 256, 416)
    with torch.no_grad():
        ref = net(x[:, None], target_size=(140, 250), source='SALICON', static=True)[0, 0, 0]
    pre = U.forward_logits(weights.to_numpy_state_dict(sd), x, (140, 250))
    got = torch.log_softmax(pre.reshape(1, -1), 1).reshape(140, 250)
    assert (got - ref).abs().max() < 1e-5
