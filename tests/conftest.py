import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


def load_golden(name):
    import yaml
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return yaml.safe_load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import kqo
    kqo.build()
    return kqo
