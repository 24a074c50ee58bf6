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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kernel_cases():
    with open(os.path.join(GOLDEN, "kernel_cases.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "kernel_cases.npz"))
    return meta, data


@pytest.fixture(scope="session")
def end_to_end():
    with open(os.path.join(GOLDEN, "end_to_end.json")) as f:
        info = json.load(f)
    data = np.load(os.path.join(GOLDEN, "end_to_end.npz"))
    return info, data


@pytest.fixture(scope="session")
def appendix_b():
    with open(os.path.join(GOLDEN, "appendix_b.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def params_plumbing():
    with open(os.path.join(GOLDEN, "params_plumbing.json")) as f:
        return json.load(f)


def case_inputs(meta_row, data):
    pre = "k%04d_" % meta_row["id"]
    g = lambda k: data[pre + k]
    has = lambda k: (pre + k) in data.files
    return dict(Cs=g("Cs"), LE=g("LE"), ds=g("ds"), Fs=g("Fs"), T=g("T"), LPC=g("LPC"),
                LP=g("LP") if has("LP") else None, preds=g("preds") if has("preds") else None)
