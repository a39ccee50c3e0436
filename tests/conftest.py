import os
import sys

import pytest

# torch first: its wheel bundles its own HIP runtime under the same soname as /opt/rocm's, and the process ends up
# with whichever is loaded first.  Loaded after libohxgb.so has pulled in the system one, torch.cuda reports no
# device (seen on the GPU box when a test that only uses ctypes ran before the first torch test).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the libraries once if they are missing (they travel prebuilt to the GPU box)."""
    from tests import helpers
    helpers.ensure_built()


@pytest.fixture(scope="session")
def oracle_lib():
    from tests import helpers
    return helpers.oracle_lib()


@pytest.fixture(scope="session")
def small_model():
    """20 trees, depth <= 10, grown on the C12 grid: fast everywhere."""
    from quickchem_amd import synth
    return synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"])


@pytest.fixture(scope="session")
def deep_model():
    """The config #1 golden booster: 100 trees, depth <= 18."""
    from quickchem_amd import synth
    return synth.make_model(num_trees=100, max_depth=18, sample_log2=16, min_leaf=2, grid=synth.GRIDS["C12"])
