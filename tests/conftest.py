import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the checker (tests only)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name), allow_pickle=False)
    return load


def record_margin(name, measured, tolerance):
    """append (test quantity, measured error, asserted tolerance) to gpurun_out/test_margins.jsonl: the record behind the tolerances the
    parity tests state (DESIGN.md section 5 quotes it); never fails a test"""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_margins.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "measured": float(measured), "tolerance": float(tolerance)}) + "\n")
    except Exception:  # noqa: BLE001
        pass
