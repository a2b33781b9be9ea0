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
