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
