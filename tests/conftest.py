import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _fresh_matrix_mode():
    """ops._MMA is process-wide state an engine sets on entry (the mode of the LAST engine that ran); kernel-level tests that build
    mt_conv3d_t structs directly must not inherit it from whichever test ran before them."""
    mod = sys.modules.get('multitalent_amd.ops')
    if mod is not None:
        mod.set_mma(0)
    yield
    # library options a test may have turned (cout tiles per workgroup of the tiled backward-weight kernels: 104 = also on small volumes)
    mod = sys.modules.get('multitalent_amd.ops')
    lib = sys.modules.get('multitalent_amd._lib')
    if mod is not None and lib is not None and getattr(lib, '_lib', None) is not None:
        mod.set_option('bwdw_cw', 4)
        mod.set_option('conv_tapsplit', 1)
