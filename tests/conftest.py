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
    if _BENCH["proc"] is not None and _BENCH["proc"].poll() is None:
        _BENCH["proc"].terminate()              # (the exact child this session started)
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


# ---------------------------------------------------------------- CPU oracle of the full-size bench configuration, in the background
_BENCH = {"proc": None, "dir": None}


def _start_bench_oracle():
    import subprocess
    import tempfile
    if _BENCH["proc"] is None:
        _BENCH["dir"] = tempfile.mkdtemp(prefix="lhn_bench_oracle_")
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        log = open(os.path.join(ROOT, "gpurun_out", "bench_config_oracle.log"), "w")
        env = {k: v for k, v in os.environ.items() if not k.startswith("LHN_")}
        _BENCH["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_config_oracle.py"), _BENCH["dir"]],
                                          stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)


def pytest_collection_modifyitems(config, items):
    """A GPU session that collected the full-size bench-configuration test starts its CPU oracle right away (child process,
    half of the host cores, no GPU): by the time that test -- sorted last -- runs, the oracle's two minutes per variant are done."""
    import torch
    if any("test_zz_bench_config_gpu" in it.nodeid for it in items) and torch.cuda.device_count() > 0:
        _start_bench_oracle()


def bench_oracle(variant, timeout=900):
    """The npz tests/bench_config_oracle.py wrote for `variant` (waits for the child; fails loudly if it died)."""
    import time

    import numpy as np
    _start_bench_oracle()
    path = os.path.join(_BENCH["dir"], f"bench_oracle_{variant}.npz")
    t0 = time.time()
    while not os.path.exists(path):
        rc = _BENCH["proc"].poll()
        if rc is not None and not os.path.exists(path):
            raise RuntimeError(f"tests/bench_config_oracle.py exited with {rc} before writing {variant}: see gpurun_out/bench_config_oracle.log")
        if time.time() - t0 > timeout:
            raise RuntimeError("tests/bench_config_oracle.py did not finish in time")
        time.sleep(1.0)
    return np.load(path)
