import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def parse_regression_file(path):
    """[section] / key = value files written by regression_mod%WriteData
    (reference: src/driver/standalone/util/regression.F90:76-124)."""
    out, cur = {}, None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith("["):
                cur = line.strip("[]")
                out[cur] = {}
            else:
                k, v = [s.strip() for s in line.split("=")]
                out[cur][" ".join(k.split())] = v if k == "category" else float(v)
    return out


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
