import os
import sys

import pytest

# Resident batches take the deferred compaction unless the recent churn (spawned + deleted surfels per keyframe) is high, in which case the library
# switches to the classic two-launch chain (msl_surfel.hip, run_batch).  The parity tests are exactly the heavy-churn cases, so they pin the deferred
# path (read once per process); tests/test_surfel_gpu.py::test_classic_chain_gives_identical_maps runs them again with MSL_SF_DEFER=0.
os.environ.setdefault("MSL_SF_DEFER", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()
