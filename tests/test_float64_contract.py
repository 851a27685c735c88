"""The "final R|t within 1e-5 relative" contract (BASELINE.json north_star) under a float64 solution (tests/float64_ref.py).

Both fp32 formulations of the iteration — reference order + literal power method, and the engine's default (single-pass double moments
+ squared power start) — run to convergence on the benchmark pair; each run's per-iteration correspondences drive a float64 restatement
of the same iterations.  Asserted: (1) each mode is within 1e-5 of the float64 solution of its own correspondences in EVERY component in
the strict norm (|dq|, |dt| / |t|, |ds| / s); (2) the default mode is at least as close as the reference-order mode in every component;
(3) the two modes differ from each other by what their correspondence sets differ by: the float64 solutions of the two id sequences are
as far apart as the two fp32 results — one near-tie correspondence out of 16384 moves the least-squares translation by ~2e-5 |t|.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import float64_ref as R64   # noqa: E402


def run_mode(oracle, F, M, m, nr, **kw):
    o = oracle.OracleICP(m, nr, 2e2, 1e-6, threads=8, **kw)
    o.write_f(F); o.write_m(M); o.build_rbc()
    ids = []
    while True:
        o.step()
        ids.append(o.nn_id["id"].copy())
        if o.k >= 40 or o.converged:
            break
    f = R64.Float64ICP(F, M, 2e2, 1e-6)
    for i in ids:
        f.step(i)
    return o.T.copy(), f.T.copy(), ids


def test_float64_restatement_solves_a_known_motion():
    """The float64 iteration itself: exact correspondences of a rigidly moved, scaled set give the motion back in one step."""
    r = np.random.default_rng(4)
    n = 500
    F = np.zeros((n, 8)); F[:, :3] = r.uniform(-800, 800, (n, 3)); F[:, 4:7] = r.uniform(0, 1, (n, 3)); F[:, 3] = F[:, 7] = 1
    q = np.array([0.05, -0.02, 0.03, 0.0]); q[3] = np.sqrt(1 - (q[:3] ** 2).sum())
    Rm, t, s = R64.quat_to_rot(q), np.array([25.0, -10.0, 15.0]), 1.01
    M = F.copy(); M[:, :3] = ((F[:, :3] - t) @ Rm) / s                 # F = s R M + t
    f = R64.Float64ICP(F, M, 2e2, 1e-6)
    f.step(np.arange(n))
    e = R64.errors_against(np.concatenate([q, t, [s]]), f.T, 800.0)
    assert e["dq"] < 1e-12 and e["dt_over_t"] < 1e-12 and e["ds_over_s"] < 1e-12, e


def test_both_modes_against_the_float64_solution(oracle, engine):
    side, nr = 128, 256
    F, M = engine.synth_pair(side)
    scene = float(np.abs(F[:, :3]).max())
    Tf, Tf64, ids_f = run_mode(oracle, F, M, side * side, nr, power_fast=True, fused=True)
    Tr, Tr64, ids_r = run_mode(oracle, F, M, side * side, nr)
    ef, er = R64.errors_against(Tf, Tf64, scene), R64.errors_against(Tr, Tr64, scene)
    # (1) the north star's tolerance, every block against its OWN magnitude
    for e in (ef, er):
        assert e["dq"] < 1e-5 and e["dt_over_t"] < 1e-5 and e["ds_over_s"] < 1e-5, e
    # (2) the benchmarked mode is the closer one (measured: dq 1.8e-8 / 3.8e-8, dt 9.6e-6 / 7.7e-5 mm, ds 4e-9 / 5e-8)
    for key in ("dq", "dt_mm", "dt_over_t", "dt_over_scene", "ds_over_s"):
        assert ef[key] <= er[key], (key, ef, er)
    assert ef["dt_over_t"] < 2e-6 and er["dt_over_t"] < 1e-5          # pinned a few x above what is measured
    # (3) what separates the two modes is which correspondences they ended with, not the arithmetic of the reductions
    between32, between64 = R64.errors_against(Tf, Tr, scene), R64.errors_against(Tf64, Tr64, scene)
    assert len(ids_f) == len(ids_r)
    differing = int((ids_f[-1] != ids_r[-1]).sum())
    assert 0 < differing <= 8                                          # (1 of 16384 on this pair)
    assert abs(between32["dt_mm"] - between64["dt_mm"]) < 0.25 * between64["dt_mm"], (between32, between64)
    assert between64["dt_over_t"] > 1e-5                               # the strict 1e-5 between MODES is out of reach of any arithmetic
