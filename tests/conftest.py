import os
import sys

import pytest

os.environ.setdefault("LFSR_LAB", "1")   # the library's A/B selectors (LFSR_* environment variables) are live only in a process started with LFSR_LAB set
# ... and none of them is inherited from the shell: with no selector set every operator runs the product's default kernel, and a test that wants another
# form sets its selector itself (monkeypatch).  LFSR_HIP_LIB (which library file) and the bench rehearsal knobs are not kernel selectors and stay.
_KEEP = {"LFSR_LAB", "LFSR_HIP_LIB", "LFSR_BENCH_ONE_DEVICE", "LFSR_BENCH_BACKEND", "LFSR_BENCH_TEST_RANK"}
for _k in [k for k in os.environ if k.startswith("LFSR_") and k not in _KEEP]:
    del os.environ[_k]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
