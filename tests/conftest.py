import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return _load


@pytest.fixture(autouse=True)
def _same_random_state_for_every_test():
    """Tests must not depend on which tests ran before them.  Modules built inside a test (projector heads of the hooks,
    ...) draw their initial weights from the GLOBAL generators -- seeded here -- and the meter batching mode is global
    state -- ended below."""
    import random

    import numpy as np
    import torch
    random.seed(20260101)
    np.random.seed(20260101)
    torch.manual_seed(20260101)
    yield
    # a test that drives step_compute() by hand leaves the meters in batching mode (step_update() ends it): the next
    # test's plain meter adds would silently be queued
    try:
        from spcl_amd.contrastyou import meters
        meters.flush_batch()
    except ImportError:
        pass
