import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def adac():
    """The product package (directory name is not a Python identifier)."""
    return importlib.import_module("duckdb-adaptive-compression_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "survey_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_ctx(adac):
    """A context on cuda:0; the GPU tests go through the C ABI of libadacodec.so only."""
    adac.build()
    ctx = adac.Context(0)
    yield ctx
    ctx.close()
