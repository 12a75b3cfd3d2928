"""bench.py's one-line JSON contract (the driver depends on it): keys, types and the relations between them, on a short
run of the small fp64 workload with every optional leg enabled."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "n200_f64", "--steps", "6", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "EKF updates/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) <= 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    pv = cb["parity_vs_oracle"]
    assert pv["decisions_identical"] is True and pv["P_max_rel"] <= 1e-9 and pv["state_max_rel_componentwise"] <= 1e-9
    # round 3: the #2 kernel of the frame is on the line too, and every launch class of the downdate carries its fraction
    sw = d["roofline_sweep"]
    assert sw["us_per_panel"] > 0 and sw["panels_per_frame"] > 0 and sw["updates"] > 0 and sw["achieved"] > 0
    assert abs(sw["ms_per_frame"] - 1e-3 * sw["us_per_panel"] * sw["panels_per_frame"]) <= 1e-9
    for name, c in roof["by_launch_class"].items():
        assert name in ("m>=512", "m<512") and 0 < c["frac"] <= 1.0 and "traffic" in c
    assert d["parity_ok"] is True


def _walk(node, path=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _walk(v, f"{path}.{k}" if path else k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _walk(v, f"{path}[{i}]")
    else:
        yield path, node


def test_default_line_the_driver_runs_has_no_fraction_above_one():
    """The line `python bench.py` prints with no workload flag (n1000_f32x: int8 units, fast_fp32_configuration, all_matched):
    every `frac` anywhere in the JSON is a fraction of a peak the timed kernel actually runs against, so none exceeds 1;
    roofline, cpu_baseline and parity_ok are present; the bound labels come from the three-way rule."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["config"]["name"] == "n1000_f32x" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["roofline"] is not None and d["cpu_baseline"] is not None and d["parity_ok"] is True
    fracs = [(p, v) for p, v in _walk(d) if p.split(".")[-1].endswith("frac")]
    assert fracs, "no frac on the line"
    for p, v in fracs:
        assert v is None or 0.0 <= v <= 1.0, (p, v)
    roof = d["roofline"]
    assert roof["unit"] == "TOP/s (int8)" and roof["peak"] == 5033.0 and roof["bound"] in ("hbm", "mfma")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) <= 1e-12
    # both int8 figures of the guides beside the derived peak (round-5 review item 8)
    assert roof["guide_i8_peak_tops"] == 3944.0 and 0.0 < roof["frac_of_guide_i8_peak"] <= 1.0
    assert abs(roof["frac_of_guide_i8_peak"] - roof["achieved"] / 3944.0) <= 1e-12
    # the PCIe-inclusive secondary line (host keypoints through ekf_step) and the parity gate the configuration is held to
    pi = d["pcie_inclusive"]
    assert pi["value"] > 0 and pi["host_bytes_per_frame"] > 0 and pi["value"] <= 1.05 * d["value"]
    assert "1e-5" in d["config"]["parity_gate"]
    live = [p for p in d["cpu_baseline"]["literal_points"] if p.get("measured")]
    assert sorted(p["N"] for p in live) == [200, 350]
    for name, c in roof["by_launch_class"].items():
        assert c.get("bound", "mfma") in ("mfma", "hbm", "fixed-cost"), (name, c.get("bound"))
    am = d["all_matched"]["p_update"]
    assert am["unit"] == "TOP/s (int8)" and abs(am["frac"] - am["achieved"] / am["peak"]) <= 1e-12
    assert d["fast_fp32_configuration"]["value"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["parity_vs_oracle"]["parity_ok"] is True


def test_sweep_timing_through_the_abi():
    """ekf_timing_sweep: HIP-event time, panels and flops of the Cholesky sweep of every update since the last reset."""
    import numpy as np  # noqa: F401

    from openekfmonoslam_amd import engine
    from openekfmonoslam_amd.synth import SyntheticSequence

    seq = SyntheticSequence(50, 3)
    e = engine.EkfEngine(seq.cam, seq.par, 50, max_keypoints=264, precision=1)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    e.timing(True)
    e.timing_reset()
    infos = [e.step(*seq.frames[t]) for t in range(3)]
    sw = e.sweep_timing()
    updates = sum((1 if i.n_inliers > 0 else 0) + (1 if i.n_rescued > 0 else 0) for i in infos)
    panels = sum(-(-2 * i.n_inliers // 32) + -(-2 * i.n_rescued // 32) for i in infos)
    assert sw["updates"] == updates and sw["panels"] == panels and sw["ms"] > 0
    n = 13 + 6 * 50
    fl = sum((2.0 * M) ** 3 / 3.0 for i in infos for M in (i.n_inliers, i.n_rescued) if M > 0)
    assert abs(sw["flops_fp64"] - fl) <= 1e-9 * fl
    flb = sum((2.0 * M) ** 2 * n for i in infos for M in (i.n_inliers, i.n_rescued) if M > 0)
    assert abs(sw["flops_b"] - flb) <= 1e-9 * flb
    e.timing_reset()
    assert e.sweep_timing()["panels"] == 0
    e.close()
