import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_primitives():
    return load_golden("primitives.json")


@pytest.fixture(scope="session")
def golden_msm():
    return load_golden("msm.json")


@pytest.fixture(scope="session")
def golden_ipp():
    return load_golden("ipp.json")


@pytest.fixture(scope="session")
def golden_r1cs():
    return load_golden("r1cs.json")


@pytest.fixture(scope="session")
def golden_codec():
    return load_golden("codec.json")
