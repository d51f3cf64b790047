import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the test-only shallow topology (6 bottleneck blocks, same code paths as ResNet-101): the product's table holds the
    # reference's topologies only; oracle/model.py carries its own entry
    try:
        from regda_amd.models import Encoder
        Encoder.LAYERS.setdefault('resnet17t', (2, 1, 1, 2))
    except (ImportError, OSError):      # library not built: the tests that need it say so themselves
        pass


@pytest.fixture(scope='session')
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name), allow_pickle=False)
    return load
