import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def exact_forward_for_the_suite():
    """The suite's bit-exactness statements about forward images (oracle == HIP, bit for bit) are statements about the
    exact-summation-order kernels: every test runs with ops.EXACT_FORWARD = True.  The product's default — the batched
    pair forward with fp16 two-piece products, images to fp32 rounding — has its own tests (tests/test_fast_forward.py),
    which switch it on themselves."""
    from gaussiangrasper_amd import ops
    prev = ops.set_exact_forward(True)
    yield
    ops.set_exact_forward(prev)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on first use."""
    from oracle import oracle as O
    O.build()
    return O


def pytest_sessionfinish(session, exitstatus):
    """Achieved-error statistics of the toleranced GPU comparisons (tests/test_gpu_parity.py
    assert_close): printed, and written where a gpurun call brings them back."""
    mod = sys.modules.get("test_gpu_parity")
    stats = getattr(mod, "ERROR_STATS", None) if mod else None
    if not stats:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "grad_errors.json"), "w") as f:
            json.dump(stats, f, indent=0)
    except OSError:
        pass
    worst = {}
    for s in stats:
        key = s["what"].split("[")[0]
        w = worst.setdefault(key, {"max_abs_over_scale": 0.0, "rel_p999": 0.0, "rel_max": 0.0, "rtol": s["rtol"],
                                   "atol_frac": s["atol_frac"]})
        for k in ("max_abs_over_scale", "rel_p999", "rel_max"):
            w[k] = max(w[k], s[k])
    print("\nachieved errors of the toleranced comparisons (worst over all cases):")
    for key, w in sorted(worst.items()):
        print(f"  {key:40s} max|err|/max|ref| {w['max_abs_over_scale']:.2e}  rel p99.9 {w['rel_p999']:.2e}  "
              f"rel max {w['rel_max']:.2e}   (allowed rtol {w['rtol']:g} + {w['atol_frac']:g} max|ref|)")
