import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run with -m gpu on the MI355X box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def pytest_sessionfinish(session, exitstatus):
    """how often the fp32-oracle allowance of tests/_tol.py was the binding term: gpurun_out/tol_stats.json"""
    try:
        import json
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _tol
        if not _tol.STATS:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        calls = len(_tol.STATS)
        rows = sum(s["rows"] for s in _tol.STATS)
        needed = sum(s["needed_ref32"] for s in _tol.STATS)
        worst = sorted((s for s in _tol.STATS if s["needed_ref32"]), key=lambda s: -s["frac"])[:40]
        with open(os.path.join(out, "tol_stats.json"), "w") as f:
            json.dump({"calls": calls, "calls_needing_ref32": sum(1 for s in _tol.STATS if s["needed_ref32"]),
                       "rows": rows, "rows_needing_ref32": needed,
                       "straggler_rows": sum(s.get("stragglers", 0) for s in _tol.STATS),
                       "calls_with_stragglers": [s["what"] for s in _tol.STATS if s.get("stragglers")],
                       # rule (c) and the straggler clause exist under deep=True only (whole-model comparisons)
                       "deep_calls": sum(1 for s in _tol.STATS if s.get("deep")),
                       "deep_call_names": sorted({s["what"] for s in _tol.STATS if s.get("deep")}),
                       "straggler_rows_outside_deep_calls": sum(s.get("stragglers", 0) for s in _tol.STATS
                                                                if not s.get("deep")),
                       "rule_c_rows": sum(s.get("rule_c_rows", 0) for s in _tol.STATS),
                       "rule_c_rows_outside_deep_calls": sum(s.get("rule_c_rows", 0) for s in _tol.STATS
                                                             if not s.get("deep")),
                       "largest_fractions": worst}, f, indent=1)
    except Exception:
        pass
