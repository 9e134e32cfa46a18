import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the fused front kernel keeps the normalised network input on chip; the tests read it back through svc_debug_tap
os.environ.setdefault('SVC_KEEP_INPUT', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def synthetic_sd():
    from retargetvid_amd import weights
    return weights.make_synthetic_state_dict(0)


@pytest.fixture(scope='session')
def engine(synthetic_sd):
    """One device engine per test session (GPU tests only)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no GPU is visible')
    from retargetvid_amd import ops
    eng = ops.Engine(synthetic_sd)
    yield eng
    eng.close()
