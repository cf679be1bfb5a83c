import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def jsg():
    """The product package with libjsg.so built in-tree (hipcc cross-compiles without a GPU)."""
    import jadespectrogram_amd
    from jadespectrogram_amd import _build
    if not os.path.exists(_build.LIB):
        _build.build_lib()
    jadespectrogram_amd.capi.lib()
    return jadespectrogram_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import jsg_oracle
    return jsg_oracle


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "colormap_ref.json")) as fh:
        cm = json.load(fh)
    with open(os.path.join(ROOT, "tests", "golden", "survey_kats.json")) as fh:
        kats = json.load(fh)
    return {"colormap": cm, "kats": kats}
