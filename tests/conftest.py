import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; ignored when the plugin is absent)")


def pem_to_der(path):
    import base64
    lines = [l.strip() for l in open(path) if l.strip() and not l.startswith("-----")]
    return base64.b64decode("".join(lines))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_certs(golden_dir):
    return {n: pem_to_der(os.path.join(golden_dir, n + ".pem"))
            for n in ("kLeadingZeroes", "kEmptySPKI", "kRealSPKI")}
