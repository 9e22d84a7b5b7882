import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---------------------------------------------------------------- measured parity margins
# Every GPU parity test reports what it MEASURED next to the bar it asserted (heatmap error, worst gradient ratio against the
# reference's own fp32 error, argmax disagreements, PCK delta ...).  `pytest -q` swallows prints, so the numbers are collected
# here and written to gpurun_out/parity_<round>.json at the end of the session (copied to profiles/ for the record).
_PARITY = {}


def parity_record(tag, **numbers):
    e = _PARITY.setdefault(tag, {})
    for k, v in numbers.items():
        e[k] = v if isinstance(v, (str, list, dict, bool)) else float(v)


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    meta = {"deterministic": os.environ.get("LHN_DETERMINISTIC", "0"), "strict_bars": os.environ.get("LHN_STRICT_BARS", "0"),
            "exitstatus": int(exitstatus), "entries": len(_PARITY)}
    name = os.environ.get("LHN_PARITY_FILE", "parity_r03.json")
    path = os.path.join(out, name)
    old = {}
    if os.path.exists(path) and os.environ.get("LHN_PARITY_APPEND") == "1":
        try:
            old = json.load(open(path)).get("tests", {})
        except (OSError, ValueError):
            old = {}
    old.update(_PARITY)
    json.dump({"meta": meta, "tests": old}, open(path, "w"), indent=1, sort_keys=True)
