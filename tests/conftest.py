import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def fuzz_case(po, row):
    """Rebuild (P, T) of a fuzz_vectors.json row (see gen_golden.make_case)."""
    if row["kind"] == "periodic":
        unit = po.gen_text(row["seed"], row["sigma"], 0, row["unit"])
        T = np.resize(unit, row["n"])
    else:
        T = po.gen_text(row["seed"], row["sigma"], 0, row["n"])
    P = T[row["k"]:row["k"] + row["m"]].copy()
    if row["kind"] == "mutated":
        P[row["mut"]] = (int(P[row["mut"]]) + 1) % row["sigma"]
    return P, T


@pytest.fixture
def ab_library():
    """The A/B build of the library (the product plus the superseded kernels of kernels_ab.inc, which
    smartgpu_tune selects) for the duration of one test; everything the test creates goes through it."""
    from smart_amd import engine
    L = engine.use_library(engine.AB_LIB_PATH)
    try:
        yield L
    finally:
        for key in range(8):
            L.smartgpu_tune(key, 0)
        engine.use_library()
