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
def _selection_state_does_not_leak():
    """The C ABI has no process-wide state (ABI 4); what remains is the DEFAULT selection / matrix mode ops.py writes into the problem
    structs it builds (ops.set_option, ops.set_mma: host-side conveniences of kernel-level tests).  A test that changes them restores
    them itself; this fixture only CHECKS that — a leak fails the test that leaked (and is undone, so that it fails alone).  Together
    with MT_TEST_SHUFFLE=<seed> (below) this is how the suite is shown to be order-independent."""
    yield
    mod = sys.modules.get('multitalent_amd.ops')
    if mod is None:
        return
    leaked = not mod.options_are_default()
    if leaked:
        state = (hex(mod._select), dict(mod._caps), mod._MMA)
        mod._select, mod._MMA = mod._select_env, 0
        mod._caps.clear()
        pytest.fail("the test left a non-default kernel selection / matrix mode behind: select, caps, mma = %r" % (state,))


def pytest_collection_modifyitems(config, items):
    """MT_TEST_SHUFFLE=<seed>: run the collected tests in a seeded random order (no plugin needed)."""
    seed = os.environ.get('MT_TEST_SHUFFLE')
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)
