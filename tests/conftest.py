"""pytest configuration: markers, import path, shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, ABI surface.
`-m gpu` runs on the MI355X box: parity of the HIP path against the oracle through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build_native()
    return p


@pytest.fixture(scope="session")
def L(pkg):
    return pkg.lib()


@pytest.fixture(scope="session")
def O():
    o = entry.load_oracle()
    if not os.path.exists(o.LIB_PATH):
        o.build()
    return o


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "blur_golden.json")) as f:
        return json.load(f)
