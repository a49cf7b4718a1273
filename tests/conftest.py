import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the experiment switches some tests flip (NVDR_DEBUG, NVDR_LG_MODE, NVDR_SHADE_QUEUE: alternative formulations used as cross-checks)
# are only honoured by the library when NVDR_TUNING=1 (csrc/core.hip nvdr_tuning_env)
os.environ.setdefault('NVDR_TUNING', '1')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with -m gpu on the MI355X box)")


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
