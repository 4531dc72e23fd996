import os
import sys

import pytest

# Which map chain a resident batch takes is the library's choice (msl_surfel.hip, run_batch: deferred windows for a handle on one caller stream under low
# churn, the classic k_fuse + k_compact pair otherwise -- a handle with its own two streams always).  The parity tests pin the deferred chain (read once
# per process) because it is the more intricate one; tests/test_surfel_gpu.py::test_classic_chain_gives_identical_maps runs them again with MSL_SF_DEFER=0,
# and ::test_default_policy_picks_the_chain_and_keeps_parity runs the unset-environment policy itself in a child process.
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
