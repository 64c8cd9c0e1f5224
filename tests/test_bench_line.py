"""The line bench.py prints last is what the driver parses: strict JSON, bounded size, the contract's keys + roofline + cpu_baseline.
(Round 5's line had grown to 20.7 KB and the driver recorded `parsed: null`.)  CPU only: the record of a real run
(profiles/r05_h_bench_default_flags.json, the full object bench.py built in round 5) is pushed through bench.compact_line, then
inflated the way a longer run inflates it (more windows, longer notes, NaN in a side measurement)."""
import importlib.util
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "profiles", "r05_h_bench_default_flags.json")


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _strict(line):
    def no_const(c):
        raise ValueError(f"non-finite constant {c} in the bench line")
    return json.loads(line, parse_constant=no_const)


def _check_contract(d, bench):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert isinstance(d["config"]["workload"], str) and d["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "algorithmic_bytes"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["full_record"] == bench.FULL_RECORD


def test_compact_line_of_a_real_record(bench):
    full = json.load(open(FULL))
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 6144
    d = _strict(line)
    _check_contract(d, bench)
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["roofline"]["frac"] == full["roofline"]["frac"] and d["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert d["mask_iou_vs_reference"]["windows_at_0.99"] == full["mask_iou_vs_reference"]["windows_at_0.99"]
    assert d["full_schedule"]["value"] == full["full_schedule"]["value"]
    assert d["secondary"]["value"] == full["secondary"]["value"]


def test_compact_line_stays_bounded_when_the_record_grows(bench):
    full = json.load(open(FULL))
    full["mask_iou_vs_reference"]["windows"] = [{"window": i, "iou": 1.0, "identical_fraction": 1.0, "restarts_in_place": 10} for i in range(512)]
    full["config"]["workload"] *= 20
    full["metric"] *= 10
    full["cpu_baseline"]["sample"] *= 30
    full["roofline"]["traffic_by_kernel"] = {f"k{i}": {"launches": i, "bytes_per_launch": 1 << 30} for i in range(300)}
    full["roofline_post_unet"]["kernels"] *= 40
    full["secondary"]["mask_iou_vs_reference"]["windows"] = full["mask_iou_vs_reference"]["windows"]
    full["secondary"]["config"]["workload"] *= 20
    full["secondary_fp8"]["roofline"]["note"] = "x" * 50000
    full["step45"] = {"sd_ms_per_window": 1.0, "note": "y" * 10000, "per_label": list(range(1000))}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    _check_contract(_strict(line), bench)


def test_emit_prints_one_strict_line_last_and_writes_the_full_record(bench, tmp_path, monkeypatch):
    full = json.load(open(FULL))
    full["roofline_post_unet"]["kernels"][0]["GB/s"] = float("nan")       # a side measurement that went wrong must not poison the line
    full["fast_mode"]["value"] = float("inf")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("earlier noise on stdout")
        bench.emit(full)
    lines = [x for x in buf.getvalue().split("\n") if x]
    d = _strict(lines[-1])
    _check_contract(d, bench)
    assert len(lines[-1]) < bench.LINE_LIMIT
    rec = _strict(open(tmp_path / bench.FULL_RECORD).read())
    assert rec["roofline_post_unet"]["kernels"][0]["GB/s"] is None and rec["fast_mode"]["value"] is None
    assert len(rec["mask_iou_vs_reference"]["windows"]) == len(full["mask_iou_vs_reference"]["windows"])     # nothing is dropped from the record


def test_multi_gpu_secondary_summary_fits(bench):
    """The N > 1 line: the SVD configs[3] leg is summarised with its own n_gpus / rccl_ranks."""
    full = json.load(open(FULL))
    full["n_gpus"], full["rccl_ranks"], full["dist_backend"] = 8, 8, "nccl (RCCL)"
    for k in ("cpu_baseline", "full_schedule", "fast_mode", "secondary_fp8", "roofline_post_unet"):
        full.pop(k, None)
    full["secondary"].update({"n_gpus": 8, "rccl_ranks": 8, "scaling": "weak"})
    d = _strict(bench.compact_line(full))
    assert d["secondary"]["n_gpus"] == 8 and d["secondary"]["rccl_ranks"] == 8 and d["n_gpus"] == 8
