"""CPU tests of bench.py's host-side pieces (no GPU): the line is strict JSON whatever the extras return, the contract's flags
parse, and without a GPU the script says so and exits non-zero instead of printing a number."""
import importlib.util
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_non_finite_numbers_leave_the_line_as_null():
    b = _bench()
    line = {"value": 1.5, "nested": {"a": float("nan"), "b": [1.0, float("inf"), {"c": -float("inf")}], "d": "text", "e": None, "f": 3}}
    out = json.loads(json.dumps(b._finite(line)), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert out == {"value": 1.5, "nested": {"a": None, "b": [1.0, None, {"c": None}], "d": "text", "e": None, "f": 3}}
    assert b._median_ms([0.001, 0.003, 0.002]) == 2.0 and math.isclose(b.F64_MFMA_PEAK_TFLOPS, 78.6)


def test_workloads_are_the_baseline_configurations():
    b = _bench()
    g, l = b.WORKLOADS["global"], b.WORKLOADS["local"]
    assert (g["n_cams"], g["n_pts"], g["n_obs"]) == (1000, 100000, 1000000)      # BASELINE.json configs[3]: the configuration the metric is quoted on
    assert (l["n_cams"], l["n_pts"], l["n_obs"]) == (20, 5000, 50000)            # configs[2]


def test_without_a_gpu_the_bench_refuses_to_print_a_number():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs an MI355X" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
