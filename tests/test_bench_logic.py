"""CPU tests of bench.py's own arithmetic (no GPU, no engine): the per-frame statistics of the tracking lines."""
import numpy as np

import bench


def test_gap_over_same_hop_separates_workload_from_jitter():
    """A sequence of period 8 whose frames take very different times by design (3 ... 40 iterations) has a raw p99 / p50 far above 1, a
    `gap_over_same_k` that still mixes frames with different neighbours, and a `gap_over_same_hop` of 1.0 — until one frame is late."""
    period = [90.0, 380.0, 140.0, 143.0, 83.0, 382.0, 134.0, 140.0]       # (the warm-start pass of the bench, microseconds)
    ks = [5, 40, 14, 3, 3, 40, 13, 4]
    hops = 256
    gaps = np.array([period[i % 8] for i in range(hops)])
    k = [ks[i % 8] for i in range(hops)]
    r = bench._track_report(hops, gaps.sum() * 1e-6, gaps, gaps, k, period=8)
    assert r["completion_gap_us"]["p99"] / r["completion_gap_us"]["p50"] > 2.5
    assert abs(r["gap_over_same_hop"]["p99"] - 1.0) < 1e-12 and r["gap_over_same_hop"]["frames_above_1.25x"] == 0
    assert r["gap_over_same_k"]["p99"] > 1.2                                 # k = 3 twice per period, 143 and 83 us: not jitter
    assert abs(r["frames_per_s"] - hops / (gaps.sum() * 1e-6)) < 1e-6
    late = gaps.copy(); late[100] *= 3.0
    r2 = bench._track_report(hops, late.sum() * 1e-6, late, late, k, period=8)
    assert r2["gap_over_same_hop"]["frames_above_1.25x"] == 1 and abs(r2["gap_over_same_hop"]["max"] - 3.0) < 1e-12


def test_dist_reports_the_percentiles_of_the_pass():
    d = bench._dist(np.arange(1, 101, dtype=float))
    assert d["p50"] == 50.5 and d["max"] == 100.0 and abs(d["mean"] - 50.5) < 1e-12 and 98.0 < d["p99"] <= 100.0


def test_cpu_share_is_at_least_one_core_and_at_most_the_host():
    import os
    s = bench.cpu_share()
    assert 1.0 <= s <= float(os.cpu_count() or 1)


def _canned_full(n_gpus=1):
    """A full measurement record of the size a real run produces (round 5's line was 23.7 KB and the driver's record, which keeps 8 KB of
    stdout, lost it), with long notes everywhere a real run has them."""
    long = "x" * 3000
    cfgs = {k: {"us_per_iteration": 17.912345678, "hbm_frac": 0.033123456, "valu_issue_frac": 0.6012345, "k_search_avg_launch_us": 11.8912345,
                "build_rbc_ms": 0.04212345, "kernel_us": {"a" * 60: 1.0}, "note": long} for k in ("A_x64", "B", "C")}
    track = {w: {v: {"frames_per_s": 6009.68, "gap_over_same_hop": {"p99": 1.2134}, "completion_gap_us": {"p50": 1.0}, "note": long}
                 for v in ("blocking", "pipelined_pageable", "pipelined_registered")} for w in ("cold_start", "warm_start")}
    holes = {"A_holes": {n: {"us_per_iteration": 8.91234567, "N_max": 1653} for n in ("clean", "scattered10", "blobs30", "blobs10_rgb0", "blobs30_rgb0")},
             "B_holes": {n: {"us_per_iteration": 22.0912345} for n in ("blobs30", "blobs30_rgb0")},
             "A_x64_holes": {"blobs30": {"us_per_iteration": 2.2912345}}, "A_wall": {"a_2e2": {"us_per_iteration": 9.312345}}}
    full = {"metric": "ICP iterations/sec at |F|=|M|=16384, |R|=256", "value": 114206.123456789, "unit": "iterations/s", "n_gpus": n_gpus,
            "steps": 20, "warmup": 5, "ms_per_step": 0.350245123456, "us_per_iteration": 8.7561234567, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic kg-like pair, |F|=|M|=16384, |R|=256, power method, weighted, a=2e2 c=1e-6; step = 40 fixed "
                                   "iterations (one hipGraph), RBC prebuilt", "parallelism": "single", "registrations_per_gpu": 1, "power_start": "squared",
                       "setup": long, "reduce_mode": "fused", "launches_per_iteration": 1},
            "roofline": {"bound": "hbm", "kernel": "k_search<chained> (finalize of the previous iteration in its prologue)", "achieved": 135.6678901,
                         "peak": 8000.0, "unit": "GB/s", "frac": 0.016958486, "frac_moved": 0.0152123456, "traffic": 2160000.0, "traffic_source": {"how": long},
                         "algorithmic_bytes_per_launch": 1187904, "avg_launch_us": 8.7561234, "kernel_us": {"b" * 80: 1.0},
                         "valu_beside_it": {"executed": {"valu_issue_frac": 0.401234567, "how": long}}, "note": long},
            "value_right_after_start": {"iterations_per_s": 108000.123, "note": long}, "per_gpu_iterations_per_s": [114206.123456789] * n_gpus,
            "config4_per_gpu_value": 541000.123456, "registration_latency": {"build_rbc_ms": 0.0153123},
            "other_configs": dict(cfgs, track=track, holes=holes), "reference_order_us_per_iteration": 50.4123456,
            "reference_order_squared_us_per_iteration": 12.3123456,
            "mode_note": {"identical_ids_frac": 0.99993896, "vs_float64": {"between_the_modes_fp32": {"dt_over_t": 2.4e-5}, "benchmarked": {"dt_over_t": 3.7e-7}, "reading": long}},
            "cpu_baseline": {"value": 1103.7123, "unit": "iterations/s", "cores": 16, "threads": 16, "host_cores": 256, "cpu_share": 16.0, "kind": "port",
                             "sweep": [{"threads": t, "iterations_per_s": 1.0, "note": long} for t in (1, 8, 16)], "sample": long,
                             "sample_short": "5000 iterations of the same pair in 12.0 s, OpenMP thread sweep [1, 8, 16], best count reported"},
            "git_head": "abc1234"}
    return full


def test_result_line_is_short_and_last_on_stdout(tmp_path):
    """The driver's record keeps the last 8 KB of stdout: the ONE line must fit (< 4 KB), be the last stdout line, carry the contract's
    fields with `roofline` and `cpu_baseline`, and the full record must land in bench_extra.json / stderr instead."""
    import io
    import json
    for n in (1, 8):
        full = _canned_full(n)
        if n > 1:
            del full["cpu_baseline"]                 # (rank 0 at N = 1 only)
            full["config"]["parallelism"] = "replicas: one rank per GPU (torch.distributed.run; gloo for the barrier and the reductions of the report, no collective on the data path)"
        assert len(json.dumps(full)) > 20000
        out, err = io.StringIO(), io.StringIO()
        out.write("some earlier chatter\n")
        text = bench.emit(full, out=out, err=err, extra_dirs=[str(tmp_path)])
        lines = out.getvalue().splitlines()
        assert lines[-1] == text and len(text.encode()) < bench.LINE_LIMIT == 4096
        line = json.loads(lines[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in line, k
        assert line["n_gpus"] == n and line["steps"] == 20 and line["warmup"] == 5 and line["config"]["workload"].startswith("configs[1]")
        assert abs(line["value"] - full["value"]) < 1e-2 and abs(line["ms_per_step"] - full["ms_per_step"]) < 1e-6
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_moved", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "valu_issue_frac"):
            assert k in line["roofline"], k
        assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-6
        if n == 1:
            for k in ("value", "unit", "cores", "threads", "host_cores", "kind", "sample"):
                assert k in line["cpu_baseline"], k
            for k in ("B_us_per_iteration", "C_us_per_iteration", "A_x64_us_per_iteration", "B_hbm_frac", "C_hbm_frac", "A_x64_hbm_frac",
                      "reference_order_us_per_iteration", "reference_order_squared_us_per_iteration"):
                assert isinstance(line[k], float), k
        else:
            assert "cpu_baseline" not in line
        assert json.load(open(tmp_path / bench.EXTRA_FILE)) == full
        assert len(err.getvalue()) < 300 and bench.EXTRA_FILE in err.getvalue()      # (a pointer: the two streams' tails together still hold the line)
        assert "other_configs" not in line and "mode_note" not in line


def test_result_line_drops_optional_scalars_before_it_grows(monkeypatch):
    """Whatever a run adds beside the metric, the line stays under the limit and keeps the contract's fields."""
    import json
    full = _canned_full()
    monkeypatch.setattr(bench, "LINE_LIMIT", 2400)
    line = json.loads(bench.compact_line(full))
    assert len(json.dumps(line)) < 2400 and "roofline" in line and "cpu_baseline" in line and "value" in line
    assert "A_x64_blobs30_us_per_iteration" not in line


def test_cpu_sweep_never_oversubscribes_the_share():
    assert bench.sweep_counts(16.0, 256) == [1, 8, 16]
    assert bench.sweep_counts(8.0, 8) == [1, 8]
    assert bench.sweep_counts(96.0, 256) == [1, 8, 16, 64, 96]
    assert bench.sweep_counts(0.5, 4) == [1]
