"""Import the reference UNISAL model code in the BUILD CONTAINER ONLY (golden generation).

/root/reference does not exist on the GPU box and nothing at test/bench/smoke time
imports this file.  The package import (unisal/__init__.py) pulls cv2/torchvision, which
are absent here, so the four needed modules are loaded individually under stub packages
(SURVEY.md Appendix D)."""
import importlib.util
import sys
import types

U = '/root/reference/3rd_party_libs/unisal/unisal'


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference_unisal():
    pk = types.ModuleType('unisal'); pk.__path__ = [U]; sys.modules['unisal'] = pk
    pm = types.ModuleType('unisal.models'); pm.__path__ = [U + '/models']; sys.modules['unisal.models'] = pm
    pk.utils = _load('unisal.utils', U + '/utils.py')
    _load('unisal.models.cgru', U + '/models/cgru.py')
    _load('unisal.models.MobileNetV2', U + '/models/MobileNetV2.py')
    model = _load('unisal.model', U + '/model.py')
    # cnn_cfg: avoid the missing mobilenet_v2.pth.tar; rnn_cfg: avoid .cuda() in cgru.py:221
    net = model.UNISAL(cnn_cfg={'pretrained': False}, rnn_cfg={'dropout': (False, False, False)})
    return net.eval(), pk.utils
