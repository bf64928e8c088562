import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from edvr_amd import _lib
    lib = _lib.lib()  # raises loudly if the HIP extension is missing: never fall back
    assert lib.edvr_check_device() == 0, lib.edvr_last_error().decode()
    ver = lib.edvr_version().decode()
    assert 'variant:' not in ver, f'{_lib.LIB_PATH} is an experiment build ({ver}): the GPU suite only runs on the product library'
    return torch.device('cuda:0')
