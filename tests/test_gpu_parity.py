"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle.

Bar: bit-exact for indices AND for every float the iteration produces (the canonical arithmetic of
DESIGN.md §3 makes the fp32 reductions reproducible), so comparisons are on the raw bit patterns.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

A, C_ = 2e2, 1e-6


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a.view(np.uint32)


def assert_bits(got, want, what):
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = np.nonzero(bits(got).reshape(-1) != bits(want).reshape(-1))[0]
    assert bad.size == 0, "%s: %d of %d words differ, first at %d: got %r want %r" % (
        what, bad.size, got.size, bad[0], got.reshape(-1)[bad[0]], want.reshape(-1)[bad[0]])


def set_modes(engine, g, power_fast=False, fused=False):
    """Explicit modes (the handle's defaults are the benchmarked ones: squared + fused)."""
    g.setPowerMode(engine.PowerMode.SQUARED if power_fast else engine.PowerMode.LITERAL)
    g.setReduceMode(engine.ReduceMode.FUSED if fused else engine.ReduceMode.REFERENCE_ORDER)


def make(engine, oracle, side, nr, rot=1, weighted=1, power_fast=False, zero_fraction=0.0, seed=0x1C9D5EED,
         max_iterations=40, fused=False):
    m = side * side
    F, M = engine.synth_pair(side, seed=seed, zero_fraction=zero_fraction)
    g = engine.ICP(0, rot, weighted)
    g.init(m, nr, A, C_, max_iterations=max_iterations)
    set_modes(engine, g, power_fast, fused)
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, rot=rot, weighted=weighted, power_fast=power_fast, threads=8,
                         max_iterations=max_iterations, fused=fused)
    o.write_f(F)
    o.write_m(M)
    return g, o, F, M


def check_rbc(engine, g, o):
    Mem = engine.Memory
    assert_bits(g.read(Mem.REPS), o.reps, "representatives")
    assert np.array_equal(g.read(Mem.RBC_OWNER), o.rbc_owner), "owner"
    assert np.array_equal(g.read(Mem.RBC_N), o.rbc_N), "N"
    assert np.array_equal(g.read(Mem.RBC_O), o.rbc_O), "O"
    assert np.array_equal(g.read(Mem.RBC_PERM), o.rbc_perm), "perm"


def check_step(engine, g, o, weighted=True):
    Mem = engine.Memory
    assert np.array_equal(g.read(Mem.RID), o.rid), "nearest representative"
    gn, on = g.read(Mem.NN_ID), o.nn_id
    assert np.array_equal(gn["id"], on["id"]), "correspondence ids: %d differ" % np.count_nonzero(gn["id"] != on["id"])
    assert_bits(gn["dist"], on["dist"], "correspondence distances")
    if weighted:
        assert_bits(g.read(Mem.W), o.W, "weights")
        assert_bits(g.read(Mem.SUM_W), np.array([o.sum_w]), "sum of weights")
    assert_bits(g.read(Mem.MEANS), o.means, "means")
    assert_bits(g.read(Mem.S), o.S, "S")
    assert_bits(g.read(Mem.TK), o.Tk, "Tk")
    assert_bits(g.read(Mem.RK).reshape(3, 3), o.Rk, "Rk")
    assert_bits(g.read(Mem.R).reshape(3, 3), o.R, "R")
    assert_bits(g.read(Mem.T), o.T, "T")


@pytest.mark.parametrize("side,nr", [(128, 256), (32, 16), (64, 64), (30, 4), (6, 4), (16, 256), (192, 2048), (96, 1024)])
def test_build_rbc(engine, oracle, side, nr):
    g, o, F, M = make(engine, oracle, side, nr)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)
    XP = g.read(engine.Memory.RBC_XP)
    assert_bits(XP, F[o.rbc_perm], "permuted database")
    g.close()


@pytest.mark.parametrize("side,nr", [(128, 256), (32, 16), (30, 4), (6, 4)])
def test_steps_bit_exact(engine, oracle, side, nr):
    """config 2 (kg-like pair, power method, weighted) and small / ragged sizes: 4 free-running steps."""
    g, o, F, M = make(engine, oracle, side, nr)
    g.buildRBC()
    o.build_rbc()
    for it in range(4):
        g.step()
        o.step()
        check_step(engine, g, o)
        assert g.state().power_iterations == o.power_iters
    assert g.k == 4
    g.close()


@pytest.mark.parametrize("rot,weighted,fast", [(1, 0, False), (0, 1, False), (0, 0, False), (1, 1, True)])
def test_variants(engine, oracle, rot, weighted, fast):
    """The other ICPStep specialisations (REGULAR weighting, EIGEN rotation) and the squared power start."""
    g, o, F, M = make(engine, oracle, 64, 64, rot=rot, weighted=weighted, power_fast=fast)
    g.buildRBC()
    o.build_rbc()
    for it in range(3):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=bool(weighted))
    g.close()


def test_zero_points(engine, oracle):
    """10 % invalid (all-zero) points collapse onto one representative list (kernels/icp_kernels.cl:50-51)."""
    g, o, F, M = make(engine, oracle, 64, 64, zero_fraction=0.1)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)
    assert o.rbc_N.max() > 300
    for it in range(2):
        g.step()
        o.step()
        check_step(engine, g, o)
    g.close()


@pytest.mark.parametrize("fast", [False, True])
def test_run_to_convergence(engine, oracle, fast):
    """ICP::run: same iteration count, same convergence decision, final R|t identical (required: 1e-5 relative)."""
    g, o, F, M = make(engine, oracle, 128, 256, power_fast=fast)
    g.buildRBC()
    o.build_rbc()
    kg = g.run()
    ko = o.run()
    assert kg == ko and kg == g.k
    assert bool(g.state().converged) == o.converged
    T, To = g.read(engine.Memory.T), o.T
    assert np.allclose(T, To, rtol=1e-5, atol=0)          # the north-star tolerance
    assert_bits(T, To, "final T")                          # what the canonical arithmetic actually gives
    assert_bits(g.R, o.R, "final R")
    check_step(engine, g, o)
    g.close()


def test_run_fixed_matches_steps(engine, oracle):
    g, o, F, M = make(engine, oracle, 64, 64)
    g.buildRBC()
    o.build_rbc()
    g.run_fixed(7)
    for _ in range(7):
        o.step()
    check_step(engine, g, o)
    assert g.k == 7
    g.close()


def test_teacher_forced(engine, oracle):
    """Feed the oracle's T_k before every step: correspondences must match exactly at every k."""
    g, o, F, M = make(engine, oracle, 64, 64)
    g.buildRBC()
    o.build_rbc()
    for it in range(6):
        g.write(engine.Memory.T, o.T, block=True)
        g.step()
        o.step()
        gn, on = g.read(engine.Memory.NN_ID), o.nn_id
        assert np.array_equal(gn["id"], on["id"])
        assert_bits(gn["dist"], on["dist"], "dist")
    g.close()


def test_max_iterations_and_thresholds(engine, oracle):
    g, o, F, M = make(engine, oracle, 64, 64, max_iterations=5)
    g.buildRBC()
    o.build_rbc()
    assert g.run() == o.run() == 5
    assert not g.state().converged
    assert_bits(g.read(engine.Memory.T), o.T, "T after max_iterations")
    # loose thresholds: stops after the first iteration that meets them
    g2, o2, _, _ = make(engine, oracle, 64, 64)
    g2.setAngleThreshold(1.0)
    g2.setTranslationThreshold(5.0)
    o2 = oracle.OracleICP(64 * 64, 64, A, C_, angle_threshold=1.0, translation_threshold=5.0, threads=8)
    o2.write_f(F)
    o2.write_m(M)
    g2.buildRBC()
    o2.build_rbc()
    assert g2.run() == o2.run()
    assert g2.state().converged and o2.converged
    assert_bits(g2.read(engine.Memory.T), o2.T, "T at loose thresholds")
    g.close()
    g2.close()


def test_batched(engine, oracle):
    """config 4 in miniature: independent registrations in one launch set, each equal to its own oracle."""
    B, side, nr = 3, 64, 64
    g = engine.ICP(0)
    g.init(side * side, nr, A, C_, batch=B)
    set_modes(engine, g)
    oracles = []
    for b in range(B):
        F, M = engine.synth_pair(side, seed=0x1C9D5EED + b, rot_deg=2.0 + b)
        g.write(engine.Memory.F, F, batch_index=b)
        g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(side * side, nr, A, C_, threads=8)
        o.write_f(F)
        o.write_m(M)
        o.build_rbc()
        oracles.append(o)
    g.buildRBC()
    k0 = g.run()
    for b, o in enumerate(oracles):
        ko = o.run()
        st = g.state(b)
        assert st.k == ko, (b, st.k, ko)
        assert_bits(g.read(engine.Memory.T, b), o.T, "T of registration %d" % b)
        assert np.array_equal(g.read(engine.Memory.NN_ID, b)["id"], o.nn_id["id"])
    assert k0 == oracles[0].k
    g.close()


@pytest.mark.parametrize("side,nr,zero_fraction,steps", [(128, 1024, 0.1, 4), (256, 4096, 0.0, 2), (256, 2048, 0.05, 2)])
def test_stage1_pruning_exact(engine, oracle, side, nr, zero_fraction, steps):
    """The dense search variant prunes stage 1 (seed bound + bounding boxes of representative groups): ids, distances
    and nearest representatives stay bit-exact, also with exact ties (10 % identical zero points), several tiles of
    representatives (nr > 1024) and a seed that changes from one iteration to the next."""
    g, o, F, M = make(engine, oracle, side, nr, zero_fraction=zero_fraction)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)                          # the owner search of the construction is the same pruned stage 1
    g.buildRBC()                                     # (cached graph: a second construction gives the same structure)
    check_rbc(engine, g, o)
    for _ in range(steps):
        g.step()
        o.step()
        check_step(engine, g, o)
    g.close()


def test_stage1_pruning_batched_ties(engine, oracle):
    """Dense by batch (9 x 64 blocks > 512): every registration equals its own oracle, zero points included."""
    B, side, nr = 9, 64, 64
    g = engine.ICP(0)
    g.init(side * side, nr, A, C_, batch=B)
    set_modes(engine, g)
    oracles = []
    for b in range(B):
        F, M = engine.synth_pair(side, seed=0x1C9D5EED + 7 * b, rot_deg=1.0 + 0.5 * b, zero_fraction=0.1 if b % 2 else 0.0)
        g.write(engine.Memory.F, F, batch_index=b)
        g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(side * side, nr, A, C_, threads=8)
        o.write_f(F)
        o.write_m(M)
        o.build_rbc()
        oracles.append(o)
    g.buildRBC()
    for it in range(3):
        g.step()
        for b, o in enumerate(oracles):
            o.step()
            assert np.array_equal(g.read(engine.Memory.RID, b), o.rid), (it, b)
            gn = g.read(engine.Memory.NN_ID, b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (it, b)
            assert_bits(gn["dist"], o.nn_id["dist"], "distances of registration %d" % b)
            assert_bits(g.read(engine.Memory.T, b), o.T, "T of registration %d" % b)
    g.close()


def test_config3_one_step(engine, oracle):
    """config 3: |F|=|M|=65536, |R|=1024 (the reference itself cannot run it: SURVEY §0.7)."""
    g, o, F, M = make(engine, oracle, 256, 1024)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)
    for _ in range(2):
        g.step()
        o.step()
        check_step(engine, g, o)
    g.close()


def test_config5_properties(engine, oracle):
    """config 5: |F|=|M|=2^20, |R|=4096 — full oracle pass is minutes of CPU, so: exact parity of the RBC
    search on a random subset of queries, exact sum of weights / means / S from the GPU's own
    correspondences (the oracle's reduction code on the GPU's nn_id), and list-structure invariants."""
    side, nr = 1024, 4096
    m = side * side
    F, M = engine.synth_pair(side)
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    set_modes(engine, g)
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    g.buildRBC()
    Mem = engine.Memory
    N, O, perm = g.read(Mem.RBC_N), g.read(Mem.RBC_O), g.read(Mem.RBC_PERM)
    assert int(N.sum()) == m
    assert np.array_equal(O, np.concatenate([[0], np.cumsum(N[:-1], dtype=np.uint64)]).astype(np.uint32))
    assert np.array_equal(np.sort(perm), np.arange(m, dtype=np.uint32))           # a permutation
    owner = g.read(Mem.RBC_OWNER)
    assert np.array_equal(owner[perm], np.repeat(np.arange(nr, dtype=np.uint32), N))  # grouped by owner
    starts = np.zeros(m, bool)
    starts[O[N > 0]] = True
    assert np.all((np.diff(perm.astype(np.int64)) > 0) | starts[1:])              # stable inside every list
    R, src = oracle.get_reps(F, nr)
    assert_bits(g.read(Mem.REPS), R, "representatives")
    rng = np.random.default_rng(7)
    sub = np.sort(rng.choice(m, 4096, replace=False))
    rbc = dict(owner=owner, N=N, O=O, perm=perm, XP=np.ascontiguousarray(F[perm]))
    own_o = oracle.rbc_search(F[sub], R, rbc, src, A)[2]                           # nearest rep of fixed points = owner
    assert np.array_equal(own_o, owner[sub])
    g.step()
    nn = g.read(Mem.NN_ID)
    nn_o, NN_o, rid_o = oracle.rbc_search(M[sub], R, rbc, src, A)                  # T = identity at k = 0
    assert np.array_equal(g.read(Mem.RID)[sub], rid_o)
    assert np.array_equal(nn["id"][sub], nn_o["id"])
    assert_bits(nn["dist"][sub], nn_o["dist"], "dist")
    # reductions: oracle code on the GPU's correspondences
    Wt, sw = oracle.weights(nn)
    assert_bits(g.read(Mem.W), Wt, "weights")
    assert_bits(g.read(Mem.SUM_W), np.array([sw]), "sum_w")
    NNp = np.ascontiguousarray(F[nn["id"]])
    means = oracle.mean_weighted(NNp, M, Wt, sw)
    assert_bits(g.read(Mem.MEANS), means, "means")
    DF, DM = oracle.devs(NNp, M, means)
    S = oracle.sij(DM, DF, Wt, C_)
    assert_bits(g.read(Mem.S), S, "S")
    Tk, _ = oracle.power_method(S, means)
    assert_bits(g.read(Mem.TK), Tk, "Tk")
    g.close()


# ---- fused reduction mode (docs/HISTORY.md §3.11): bit-exact against the oracle's fused restatement, and within the
# ---- north-star tolerance of the reference-order mode

@pytest.mark.parametrize("side,nr,rot,weighted", [(128, 256, 1, 1), (32, 16, 1, 1), (30, 4, 1, 1), (6, 4, 1, 1),
                                                   (64, 64, 1, 0), (64, 64, 0, 1), (192, 256, 1, 1), (144, 64, 1, 1)])
def test_fused_steps_bit_exact(engine, oracle, side, nr, rot, weighted):
    g, o, F, M = make(engine, oracle, side, nr, rot=rot, weighted=weighted, power_fast=True, fused=True)
    g.buildRBC()
    o.build_rbc()
    for it in range(4):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=False)
        if weighted:
            assert_bits(g.read(engine.Memory.W), o.W, "weights")
            assert_bits(g.read(engine.Memory.SUM_W), np.array([o.sum_w]), "sum of weights")
    g.close()


def test_fused_run_and_cross_mode_tolerance(engine, oracle):
    """config 2 in fused mode: identical to the fused oracle; against the reference-order run: same k, the
    same correspondences at the end, final [q | t, s] within 1e-5 relative (north star)."""
    g, o, F, M = make(engine, oracle, 128, 256, power_fast=True, fused=True)
    g.buildRBC()
    o.build_rbc()
    kg, ko = g.run(), o.run()
    assert kg == ko
    T = g.read(engine.Memory.T)
    assert_bits(T, o.T, "final T (fused)")
    check_step(engine, g, o, weighted=False)
    r, orf, _, _ = make(engine, oracle, 128, 256, power_fast=True, fused=False)
    r.buildRBC()
    kr = r.run()
    Tr = r.read(engine.Memory.T)
    assert abs(kr - kg) <= 1
    # The contract ("final R|t within 1e-5 relative", north star; docs/HISTORY.md §3.11), under a float64 solution: every mode's result against the
    # float64 restatement of its own iterations (tests/float64_ref.py, fed the correspondences the ENGINE found in each iteration), each
    # block of [q | t, s] against its own magnitude — |q| = 1, |t| (25 mm here), s.
    import float64_ref as R64
    scene = float(np.abs(F[:, :3]).max())

    def against_float64(h, k, Tend):
        h.reset_transform(); h.buildRBC()
        f = R64.Float64ICP(F, M, 2e2, 1e-6)
        for _ in range(k):
            h.step()
            f.step(h.read(engine.Memory.NN_ID)["id"])
        assert_bits(h.read(engine.Memory.T), Tend, "step by step = run ()")
        return R64.errors_against(Tend, f.T, scene), f.T

    ef, T64f = against_float64(g, kg, T)
    er, T64r = against_float64(r, kr, Tr)
    for e in (ef, er):
        assert e["dq"] < 1e-5 and e["dt_over_t"] < 1e-5 and e["ds_over_s"] < 1e-5, e
    for key in ("dq", "dt_mm", "ds_over_s"):                             # the default mode is the closer one, in every component
        assert ef[key] <= er[key], (key, ef, er)
    # BETWEEN two free-running modes: when they find the same correspondences all the way (these two do: both with the squared power start),
    # what separates them is arithmetic alone and the strict 1e-5 holds between them as well (measured: |dt| / |t| = 6e-7); when a
    # near-tie correspondence flips (the literal power method against this one: 1 id of 16384, tests/test_float64_contract.py), the
    # float64 solutions of the two correspondence sets are as far apart as the two fp32 results
    b32, b64 = R64.errors_against(T, Tr, scene), R64.errors_against(T64f, T64r, scene)
    if b64["dt_over_t"] < 1e-7:
        assert b32["dq"] < 1e-5 and b32["dt_over_t"] < 1e-5 and b32["ds_over_s"] < 1e-5, b32
    else:
        assert abs(b32["dt_mm"] - b64["dt_mm"]) < 0.25 * b64["dt_mm"], (b32, b64)
    # What is measured between the modes, pinned ~2x above it so that a regression shows
    assert np.abs(T[:4] - Tr[:4]).max() < 2e-6, np.abs(T[:4] - Tr[:4]).max()
    assert np.abs(T[4:7] - Tr[4:7]).max() < 2e-3, np.abs(T[4:7] - Tr[4:7]).max()          # mm
    assert abs(T[7] - Tr[7]) < 2e-6
    g.reset_transform(); g.buildRBC(); g.run()                           # (the final correspondences of the two runs)
    r.reset_transform(); r.buildRBC(); r.run()
    ids_f, ids_r = g.read(engine.Memory.NN_ID)["id"], r.read(engine.Memory.NN_ID)["id"]
    assert np.mean(ids_f == ids_r) > 0.999
    g.close()
    r.close()


def test_fused_first_iteration_correspondences_equal_reference_order(engine, oracle):
    """Same T => the search is the same code: ids and distances are bit-identical across the two modes."""
    a, _, F, M = make(engine, oracle, 64, 64, fused=True)
    b, _, _, _ = make(engine, oracle, 64, 64, fused=False)
    for x in (a, b):
        x.buildRBC()
        x.step()
    na, nb = a.read(engine.Memory.NN_ID), b.read(engine.Memory.NN_ID)
    assert np.array_equal(na["id"], nb["id"])
    assert_bits(na["dist"], nb["dist"], "dist")
    a.close()
    b.close()


def test_fused_batched_and_large(engine, oracle):
    g, o, F, M = make(engine, oracle, 256, 1024, power_fast=True, fused=True)          # config 3
    g.buildRBC()
    o.build_rbc()
    for _ in range(2):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=False)
    g.close()


def test_fused_chain_fixed_run_and_limits(engine, oracle):
    """The chained graph (one launch per iteration): fixed runs, max_iterations, repeated runs, batches."""
    g, o, F, M = make(engine, oracle, 64, 64, power_fast=True, fused=True)
    g.buildRBC()
    o.build_rbc()
    g.run_fixed(7)
    for _ in range(7):
        o.step()
    check_step(engine, g, o, weighted=False)
    assert g.k == 7
    g.run_fixed(1)                                  # odd / even launch counts both end in the visible state
    o.step()
    check_step(engine, g, o, weighted=False)
    g.step()                                        # the two-kernel path continues from the chain's state
    o.step()
    check_step(engine, g, o, weighted=False)
    g.close()
    g, o, F, M = make(engine, oracle, 64, 64, power_fast=True, fused=True, max_iterations=5)
    g.buildRBC()
    o.build_rbc()
    assert g.run() == o.run() == 5
    assert_bits(g.read(engine.Memory.T), o.T, "T after max_iterations (chain)")
    g.buildRBC()                                    # second registration on the same handle: k resets, T persists
    o.build_rbc()
    assert g.run() == o.run()
    assert_bits(g.read(engine.Memory.T), o.T, "T after the second run")
    g.close()


def test_chain_is_automatic_only_for_latency_bound_sizes(engine, monkeypatch):
    monkeypatch.delenv("ICP_AMD_CHAIN", raising=False)
    for side, nr, batch, fused, want in ((128, 256, 1, True, 1), (128, 256, 4, True, 2), (256, 1024, 1, True, 3),
                                         (128, 256, 1, False, 4), (64, 64, 8, True, 1), (64, 64, 9, True, 2)):
        g = engine.ICP(0)
        g.init(side * side, nr, A, C_, batch=batch)
        g.setReduceMode(engine.ReduceMode.FUSED if fused else engine.ReduceMode.REFERENCE_ORDER)
        assert g.launches_per_iteration() == want, (side, nr, batch, fused)
        g.close()


def test_fused_chain_batched(engine, oracle):
    B, side, nr = 3, 64, 64
    g = engine.ICP(0)
    g.init(side * side, nr, A, C_, batch=B)
    g.setPowerMode(engine.PowerMode.SQUARED)
    g.setReduceMode(engine.ReduceMode.FUSED)
    oracles = []
    for b in range(B):
        F, M = engine.synth_pair(side, seed=77 + b, rot_deg=1.0 + 2 * b)
        g.write(engine.Memory.F, F, batch_index=b)
        g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(side * side, nr, A, C_, threads=8, power_fast=True, fused=True)
        o.write_f(F); o.write_m(M); o.build_rbc()
        oracles.append(o)
    g.buildRBC()
    g.run()
    ks = []
    for b, o in enumerate(oracles):
        ko = o.run()
        ks.append(ko)
        st = g.state(b)
        assert st.k == ko and bool(st.converged) == o.converged, (b, st.k, ko)
        assert_bits(g.read(engine.Memory.T, b), o.T, "T of registration %d" % b)
    assert len(set(ks)) > 1                          # registrations really stop at different iterations
    g.close()


@pytest.mark.parametrize("chain,launches", [(None, 1), ("1", 1), ("0", 2)])
def test_chain_modes(engine, oracle, monkeypatch, chain, launches):
    """Fused graphs run one launch per iteration (finalize in the next search's prologue, double-buffered state and
    moments) where the size is latency-bound; ICP_AMD_CHAIN=0 / 1 (read at icp_create) forces the two-launch / chained
    form.  Same bits either way, and the same as the oracle."""
    if chain is None:
        monkeypatch.delenv("ICP_AMD_CHAIN", raising=False)
    else:
        monkeypatch.setenv("ICP_AMD_CHAIN", chain)
    g, o, F, M = make(engine, oracle, 128, 256, power_fast=True, fused=True)
    assert g.launches_per_iteration() == launches
    g.buildRBC()
    o.build_rbc()
    assert g.run() == o.run()
    assert_bits(g.read(engine.Memory.T), o.T, "final T (chain)")
    check_step(engine, g, o, weighted=False)
    g.run_fixed(3)
    for _ in range(3):
        o.step()
    check_step(engine, g, o, weighted=False)
    g.close()


@pytest.mark.parametrize("side,nr", [(2, 1), (2, 4), (4, 2), (8, 64), (10, 4), (14, 4)])
@pytest.mark.parametrize("fused", [False, True])
def test_tiny_and_odd_shapes(engine, oracle, side, nr, fused):
    """Smallest sets, one representative (odd LDS pair tile), every point a representative, sides that are not
    multiples of 8 (fused mode falls back to linear 64-query blocks), partially filled blocks."""
    g, o, F, M = make(engine, oracle, side, nr, power_fast=True, fused=fused)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)
    for _ in range(3):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=not fused)
    g.close()


def _random_cases(n, seed=20261003):
    rng = np.random.default_rng(seed)
    cases = []
    sides = [8, 12, 16, 24, 32, 40, 48, 64, 96, 128]
    while len(cases) < n:
        side = int(rng.choice(sides))
        nr = int(2 ** rng.integers(0, 11))
        p = nr.bit_length() - 1
        nrx, nry = 1 << (p - p // 2), 1 << (p // 2)
        if nr > side * side or side % nrx or side % nry:
            continue
        batch = int(rng.choice([1, 1, 1, 2, 5, 9])) if side <= 64 else 1
        cases.append((side, nr, batch, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(0, 2)),
                      int(rng.integers(0, 2)), float(rng.choice([0.0, 0.0, 0.05, 0.3])), int(rng.integers(1, 5)),
                      int(rng.integers(1, 1 << 30))))
    return cases


@pytest.mark.parametrize("side,nr,batch,fused,squared,weighted,rot,zero_fraction,steps,seed", _random_cases(28))
def test_randomized_configurations(engine, oracle, side, nr, batch, fused, squared, weighted, rot, zero_fraction, steps, seed):
    """Seeded random sweep over sizes, representative counts (1 .. 1024: one block per CU, dense, several LDS tiles' worth of
    groups), batches, reduction / power-method modes, weighting, rotation solver and zero points: after every step the
    nearest representatives, correspondences and the transform of every registration equal its own oracle bit for bit;
    then a graph run (chained where the size allows) from that state does too."""
    m = side * side
    g = engine.ICP(0, rot, weighted)
    g.init(m, nr, A, C_, batch=batch)
    set_modes(engine, g, squared, fused)
    oracles = []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=seed + 13 * b, rot_deg=1.0 + b, zero_fraction=zero_fraction)
        g.write(engine.Memory.F, F, batch_index=b)
        g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(m, nr, A, C_, rot=rot, weighted=weighted, power_fast=squared, threads=8, fused=fused)
        o.write_f(F)
        o.write_m(M)
        o.build_rbc()
        oracles.append(o)
    g.buildRBC()

    def compare(tag):
        for b, o in enumerate(oracles):
            assert np.array_equal(g.read(engine.Memory.RID, b), o.rid), (tag, b, "rid")
            gn = g.read(engine.Memory.NN_ID, b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (tag, b, "ids")
            assert_bits(gn["dist"], o.nn_id["dist"], "distances %s/%d" % (tag, b))
            assert_bits(g.read(engine.Memory.T, b), o.T, "T %s/%d" % (tag, b))

    for it in range(steps):
        g.step()
        for o in oracles:
            o.step()
        compare("step %d" % it)
    g.run_fixed(3)
    for o in oracles:
        for _ in range(3):
            o.step()
    compare("graph")
    g.close()


@pytest.mark.parametrize("tag,side,nr", [("fs32", 32, 16), ("fs128", 128, 256)])
def test_engine_against_committed_golden_vectors(engine, tag, side, nr):
    """The HIP path against the committed fixture itself (tests/golden/oracle_vectors.npz, no oracle in the loop):
    the bench's modes, four steps and a run to convergence, bit for bit."""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
    F, M = engine.synth_pair(side)
    g = engine.ICP(0)
    g.init(side * side, nr, A, C_)
    g.setPowerMode(engine.PowerMode.SQUARED)
    g.setReduceMode(engine.ReduceMode.FUSED)
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    g.buildRBC()
    for it in range(4):
        g.step()
        assert_bits(g.read(engine.Memory.T), gold[tag + "_T"][it], "T at step %d" % it)
        assert_bits(g.read(engine.Memory.S), gold[tag + "_S"][it], "S at step %d" % it)
        assert np.array_equal(g.read(engine.Memory.NN_ID)["id"][:64], gold[tag + "_ids"][it])
    g.buildRBC()                                     # the reference's contract: buildRBC (k <- 0, T kept) before every run
    assert g.run() == gold[tag + "_run_k"][0]
    assert_bits(g.read(engine.Memory.T), gold[tag + "_run_T"], "T after the run")
    g.close()


def test_defaults_are_the_benchmarked_modes(engine, oracle, monkeypatch):
    """A handle straight out of icp_create runs fused reductions + the squared power start (one launch per iteration at
    the reference's size) and equals that mode's oracle bit for bit; ICP_AMD_MODE=reference (read at icp_create)
    starts handles in the reference-order / literal modes."""
    monkeypatch.delenv("ICP_AMD_MODE", raising=False)
    monkeypatch.delenv("ICP_AMD_CHAIN", raising=False)
    F, M = engine.synth_pair(128)
    g = engine.ICP(0)
    g.init(16384, 256, A, C_)
    assert g.launches_per_iteration() == 1
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    g.buildRBC()
    o = oracle.OracleICP(16384, 256, A, C_, threads=8, power_fast=True, fused=True)
    o.write_f(F); o.write_m(M); o.build_rbc()
    assert g.run() == o.run()
    assert_bits(g.read(engine.Memory.T), o.T, "final T (default modes)")
    g.close()
    monkeypatch.setenv("ICP_AMD_MODE", "reference")
    r = engine.ICP(0)
    r.init(16384, 256, A, C_)
    assert r.launches_per_iteration() == 4
    r.write(engine.Memory.F, F)
    r.write(engine.Memory.M, M)
    r.buildRBC()
    orf = oracle.OracleICP(16384, 256, A, C_, threads=8)
    orf.write_f(F); orf.write_m(M); orf.build_rbc()
    assert r.run() == orf.run()
    assert_bits(r.read(engine.Memory.T), orf.T, "final T (ICP_AMD_MODE=reference)")
    r.close()


def test_config4_real_shape_batch64(engine, oracle):
    """BASELINE config 4 at its real per-GPU shape: 64 independent registrations of |F|=|M|=16384, |R|=256 in one launch
    set (the dense k_search<true,false,4,8> variant with exact stage-1 pruning), default modes (fused + squared).
    run() and run_fixed(40): k, converged, T and all 16384 correspondence ids of 8 of the 64 registrations (they stop at
    different k) bit for bit against the oracle AND against the committed fixture tests/golden/config4_vectors.npz."""
    from icp_amd import workloads as C4
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_vectors.npz"))
    g = engine.ICP(0)
    g.init(C4.M_POINTS, C4.NR, C4.A, C4.C_, batch=C4.PER_GPU)
    assert g.launches_per_iteration() == 2                    # dense: search + finalize
    pairs = {}
    for i in range(C4.PER_GPU):
        F, M = C4.pair(engine, i)
        g.write(engine.Memory.F, F, batch_index=i)
        g.write(engine.Memory.M, M, batch_index=i)
        if i in C4.CHECKED:
            pairs[i] = (F, M)
    g.buildRBC()
    g.run()
    oracles, ks = {}, []
    for i in C4.CHECKED:
        o = oracle.OracleICP(C4.M_POINTS, C4.NR, C4.A, C4.C_, power_fast=True, fused=True, threads=8)
        o.write_f(pairs[i][0]); o.write_m(pairs[i][1]); o.build_rbc()
        ko = o.run()
        st = g.state(i)
        ids = g.read(engine.Memory.NN_ID, i)["id"]
        T = g.read(engine.Memory.T, i)
        assert (st.k, bool(st.converged)) == (ko, o.converged), (i, st.k, ko)
        assert_bits(T, o.T, "T of registration %d" % i)
        assert np.array_equal(ids, o.nn_id["id"]), i
        assert (st.k, int(st.converged)) == tuple(int(v) for v in gold["r%d_run" % i]), i
        assert_bits(T, gold["r%d_run_T" % i], "T of registration %d vs fixture" % i)
        assert np.array_equal(ids[:256], gold["r%d_run_ids_head" % i]) and np.array_equal(C4.ids_digest(ids), gold["r%d_run_ids_digest" % i]), i
        oracles[i] = o
        ks.append(ko)
    assert len(set(ks)) >= 4                                   # the registrations really stop at different iterations
    g.reset_transform()
    g.run_fixed(40)
    for i in C4.CHECKED:
        o = oracles[i]
        o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
        for _ in range(40):
            o.step()
        ids = g.read(engine.Memory.NN_ID, i)["id"]
        T = g.read(engine.Memory.T, i)
        assert g.state(i).k == 40
        assert_bits(T, o.T, "T after 40 fixed iterations, registration %d" % i)
        assert np.array_equal(ids, o.nn_id["id"]), i
        assert_bits(T, gold["r%d_fixed40_T" % i], "fixed-40 T of registration %d vs fixture" % i)
        assert np.array_equal(ids[:256], gold["r%d_fixed40_ids_head" % i]) and np.array_equal(C4.ids_digest(ids), gold["r%d_fixed40_ids_digest" % i]), i
    g.close()


@pytest.mark.parametrize("fused", [False, True])
def test_metric_absolute_scale(engine, oracle, fused):
    """icp_set_metric_scale: dist = f_g (geo + a pho).  Same correspondences as f_g = 1 (the argmin depends on the ratio
    f_p / f_g = a only), distances scaled, and the WEIGHTED pipeline (weights, means, S, T) follows the scaled distances:
    bit for bit against the oracle with the same scale (normalised form f_g = 1 / (1 + a))."""
    fg = np.float32(1.0) / np.float32(1.0 + A)
    g, o, F, M = make(engine, oracle, 64, 64, power_fast=True, fused=fused)
    o.L.orc_icp_set_dist_scale(o.h, float(fg))
    g.setMetricScale(float(fg))
    assert g.getMetricScale() == pytest.approx(float(fg))
    g1, o1, _, _ = make(engine, oracle, 64, 64, power_fast=True, fused=fused)
    for x in (g, g1):
        x.buildRBC()
    o.build_rbc()
    g.step(); g1.step(); o.step()
    n, n1 = g.read(engine.Memory.NN_ID), g1.read(engine.Memory.NN_ID)
    assert np.array_equal(n["id"], n1["id"])
    assert_bits(n["dist"], (fg * n1["dist"]).astype(np.float32), "scaled distances")
    assert not np.array_equal(g.read(engine.Memory.T).view(np.uint32), g1.read(engine.Memory.T).view(np.uint32))   # the weights changed
    for _ in range(3):
        g.step(); o.step()
    check_step(engine, g, o, weighted=not fused)
    g.buildRBC(); o.build_rbc()                      # (k <- 0, T kept: the reference's contract before every run)
    assert g.run() == o.run()
    assert_bits(g.read(engine.Memory.T), o.T, "final T with a scaled metric")
    with pytest.raises(engine.ICPError):
        g.setMetricScale(0.0)
    g.close(); g1.close()


def test_fused_large_set_first_tree_level_kernel(engine, oracle):
    """|F| = |M| = 262144 (4096 blocks, 32 tree groups > ICP_L1_MIN_GROUPS): the first level of the moment tree runs as a
    kernel of its own (k_moment_level1) in front of k_finalize_fused — same tree, same bits as the oracle."""
    g, o, F, M = make(engine, oracle, 512, 1024, power_fast=True, fused=True)
    g.buildRBC()
    o.build_rbc()
    for _ in range(2):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=False)
    g.run_fixed(2)                                   # the graph form (two launches + the level-1 kernel per iteration)
    for _ in range(2):
        o.step()
    check_step(engine, g, o, weighted=False)
    g.close()


@pytest.mark.parametrize("side,nr,batch", [(128, 256, 1), (64, 64, 2), (256, 1024, 1)])
def test_run_fixed_fresh_equals_reset_plus_run(engine, oracle, side, nr, batch):
    """icp_run_fixed_fresh = reset_transform + run_fixed as one graph (chained form: the first search starts from the
    identity itself, no reset launch): same bits as the two calls, after a run that left another T, k and S behind."""
    m = side * side
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=batch)
    oracles = []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=0x99 + b, rot_deg=2.0 + b)
        g.write(engine.Memory.F, F, batch_index=b); g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=True, fused=True)
        o.write_f(F); o.write_m(M); o.build_rbc()
        oracles.append(o)
    g.buildRBC()
    g.run()                                          # leaves a converged T, k > 0 (and done = 1) behind
    g.run_fixed_fresh(6)
    for b, o in enumerate(oracles):
        for _ in range(6):
            o.step()
        assert g.state(b).k == 6 and not g.state(b).converged
        assert_bits(g.read(engine.Memory.T, b), o.T, "T after a fresh fixed run, registration %d" % b)
        assert_bits(g.read(engine.Memory.S, b), o.S, "S")
        assert np.array_equal(g.read(engine.Memory.NN_ID, b)["id"], o.nn_id["id"])
    g.run_fixed(2)                                   # continues from there
    for b, o in enumerate(oracles):
        for _ in range(2):
            o.step()
        assert_bits(g.read(engine.Memory.T, b), o.T, "T after continuing")
    g.close()


@pytest.mark.parametrize("side,nr,zero_fraction", [(256, 4096, 0.0), (256, 2048, 0.05), (128, 512, 0.1)])
def test_stage1_pruning_exact_fused_multi_tile(engine, oracle, side, nr, zero_fraction):
    """The dense search with several 256-representative LDS tiles (a block's tile set decided in one pre-pass: every query
    tests all tile boxes against its seed bound, the block ORs the answers, only those tiles are staged) in the default
    modes: RBC structure, nearest representatives, ids, distances, T bit for bit — loose seeds (first step: every tile)
    and tight ones (later steps: one or two tiles), exact ties (identical zero points)."""
    g, o, F, M = make(engine, oracle, side, nr, zero_fraction=zero_fraction, power_fast=True, fused=True)
    g.buildRBC()
    o.build_rbc()
    check_rbc(engine, g, o)
    for _ in range(3):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=False)
    g.run_fixed(2)
    for _ in range(2):
        o.step()
    check_step(engine, g, o, weighted=False)
    g.close()


@pytest.mark.parametrize("side,nr,batch,fused,zero_fraction,layout", [
    (256, 256, 1, True, 0.0, (1, 256, 1)),          # one 256-tile, lists of 256
    (256, 512, 1, True, 0.1, (1, 256, 1)),          # several 256-tiles, lists of 128, exact ties (identical zero points)
    (256, 256, 1, False, 0.05, (1, 256, 1)),        # reference-order reductions
    (128, 64, 3, True, 0.0, (1, 256, 1)),           # dense by batch
    (256, 1024, 1, True, 0.0, (1, 256, 0)),         # lists of 64: the lanes of a query scan (the other form)
])
def test_stage2_lanes_as_candidates(engine, oracle, side, nr, batch, fused, zero_fraction, layout):
    """Dense search, long lists (>= 128 candidates on average): stage 2 runs with lanes = candidates — a wave loads a list once
    for all of its queries that share it, per-lane best (distance, trip) per query, one 64-lane butterfly at the end.  Ids,
    distances (ties -> lowest position, clamped duplicates, empty / invalid queries), nearest representatives and T bit for
    bit against the oracle, over iterations in which the wave's queries go from many lists to one."""
    m = side * side
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=batch)
    set_modes(engine, g, power_fast=fused, fused=fused)
    assert g.search_layout() == layout
    oracles = []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=0x1C9D5EED + 11 * b, rot_deg=3.0 - 0.75 * b, zero_fraction=zero_fraction)
        g.write(engine.Memory.F, F, batch_index=b)
        g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=fused, fused=fused)
        o.write_f(F)
        o.write_m(M)
        o.build_rbc()
        oracles.append(o)
    g.buildRBC()
    for it in range(4):
        g.step()
        for b, o in enumerate(oracles):
            o.step()
            assert np.array_equal(g.read(engine.Memory.RID, b), o.rid), (it, b)
            gn = g.read(engine.Memory.NN_ID, b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (it, b, np.count_nonzero(gn["id"] != o.nn_id["id"]))
            assert_bits(gn["dist"], o.nn_id["dist"], "distances of registration %d" % b)
            assert_bits(g.read(engine.Memory.T, b), o.T, "T of registration %d" % b)
    g.close()


def test_stage2_lanes_as_candidates_1024_tile(engine, oracle):
    """The same with the 1024-representative tile (representative grid wider than 64: |R| = 8192 over 2^20 points), one step."""
    side, nr = 1024, 8192
    g, o, F, M = make(engine, oracle, side, nr, power_fast=True, fused=True)
    assert g.search_layout() == (1, 1024, 1)
    g.buildRBC()
    o.build_rbc()
    g.step()
    o.step()
    check_step(engine, g, o, weighted=False)
    g.close()


# ---- parity for exactly what bench.py times: the default modes at the benchmarked sizes, against committed fixtures
# ---- (tests/golden/bench_config_vectors.npz, made by tests/golden/make_golden.py from the oracle) and the live oracle

def _bench_gold():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_config_vectors.npz"))


def test_config_C_default_modes_fixture_and_oracle(engine, oracle):
    """BASELINE config 5 (|F|=|M|=2^20, |R|=4096) on a handle straight out of icp_create — fused moments, squared power
    start, MASKED 256-tiles, stage 2 with lanes = candidates, k_moment_level1, XCD bands: what bench.py's `other_configs.C`
    times.  The RBC structure, two free-running steps (T, S, means, sum of weights bit for bit; ALL 2^20 ids, distances and
    nearest representatives against the live oracle and, as digests, against the fixture) and the bench's own pass
    (10 fixed iterations from the identity, one graph)."""
    from icp_amd import workloads as W
    gold = _bench_gold()
    side, nr = W.CONFIGS["C"]
    m = side * side
    F, M = engine.synth_pair(side)
    g = engine.ICP(0)
    g.init(m, nr, W.A, W.C_)
    assert g.search_layout() == (1, 256, 1) and g.launches_per_iteration() == 3
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    g.buildRBC()
    Mem = engine.Memory
    assert np.array_equal(g.read(Mem.RBC_N), gold["C_N"]) and np.array_equal(g.read(Mem.RBC_O), gold["C_O"])
    assert np.array_equal(W.ids_digest(g.read(Mem.RBC_PERM)), gold["C_perm_digest"])
    assert np.array_equal(W.ids_digest(g.read(Mem.RBC_OWNER)), gold["C_owner_digest"])
    o = oracle.OracleICP(m, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    for it in range(2):
        g.step()
        o.step()
        check_step(engine, g, o, weighted=False)               # every id / distance / nearest representative, S, means, T
        nn = g.read(Mem.NN_ID)
        assert_bits(g.read(Mem.T), gold["C_T"][it], "T vs fixture, step %d" % it)
        assert_bits(g.read(Mem.S), gold["C_S"][it], "S vs fixture")
        assert_bits(g.read(Mem.MEANS), gold["C_means"][it], "means vs fixture")
        assert_bits(g.read(Mem.SUM_W), gold["C_sum_w"][it:it + 1], "sum of weights vs fixture")
        assert np.array_equal(W.ids_digest(nn["id"]), gold["C_ids_digest"][it])
        assert np.array_equal(W.ids_digest(g.read(Mem.RID)), gold["C_rid_digest"][it])
        assert np.array_equal(W.bits_digest(nn["dist"]), gold["C_dist_digest"][it])
    assert np.array_equal(g.read(Mem.NN_ID)["id"][:256], gold["C_ids_head"])
    g.run_fixed_fresh(10)                                       # the bench's step at C
    assert g.k == 10
    assert_bits(g.read(Mem.T), gold["C_fixed10_T"], "T after the bench's pass (10 fixed iterations)")
    assert np.array_equal(W.ids_digest(g.read(Mem.NN_ID)["id"]), gold["C_fixed10_ids_digest"])
    g.close()


def test_config_B_default_modes_run_fixture_and_oracle(engine, oracle):
    """BASELINE config 3 (|F|=|M|=65536, |R|=1024), default handle: ICP::run (k, converged, T, every correspondence) and the
    bench's own pass (40 fixed iterations from the identity), against the live oracle and the committed fixture."""
    from icp_amd import workloads as W
    gold = _bench_gold()
    side, nr = W.CONFIGS["B"]
    m = side * side
    F, M = engine.synth_pair(side)
    g = engine.ICP(0)
    g.init(m, nr, W.A, W.C_)
    assert g.search_layout()[:2] == (1, 256) and g.launches_per_iteration() == 3      # search, first tree level, finalize
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    g.buildRBC()
    o = oracle.OracleICP(m, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    kg, ko = g.run(), o.run()
    st = g.state()
    assert (kg, bool(st.converged)) == (ko, o.converged)
    assert (kg, int(st.converged)) == tuple(int(v) for v in gold["B_run"])
    check_step(engine, g, o, weighted=False)
    ids = g.read(engine.Memory.NN_ID)["id"]
    assert_bits(g.read(engine.Memory.T), gold["B_run_T"], "final T vs fixture")
    assert np.array_equal(ids[:256], gold["B_run_ids_head"]) and np.array_equal(W.ids_digest(ids), gold["B_run_ids_digest"])
    g.run_fixed_fresh(40)                                       # the bench's step at B
    o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
    for _ in range(40):
        o.step()
    assert g.k == 40
    check_step(engine, g, o, weighted=False)
    assert_bits(g.read(engine.Memory.T), gold["B_fixed40_T"], "T after the bench's pass (40 fixed iterations)")
    assert np.array_equal(W.ids_digest(g.read(engine.Memory.NN_ID)["id"]), gold["B_fixed40_ids_digest"])
    g.close()


def test_teacher_forced_default_modes_at_A(engine, oracle):
    """'Same T => same correspondences' at the benchmark size, over a whole trajectory: the REFERENCE-ORDER / literal oracle's
    T is written into a default-mode handle (fused + squared) before each of 10 iterations; ids and distances must equal that
    oracle's bit for bit every time — through the separate launches (step) and through the chained graph form (run_fixed (1))
    alternately."""
    side, nr = 128, 256
    m = side * side
    F, M = engine.synth_pair(side)
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    assert g.launches_per_iteration() == 1                     # default modes: the chained form
    g.write(engine.Memory.F, F)
    g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, threads=8)              # reference order, literal power method
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    for it in range(10):
        g.write(engine.Memory.T, o.T, block=True)
        if it % 2:
            g.run_fixed(1)
        else:
            g.step()
        o.step()
        gn, on = g.read(engine.Memory.NN_ID), o.nn_id
        assert np.array_equal(gn["id"], on["id"]), "iteration %d: %d ids differ" % (it, np.count_nonzero(gn["id"] != on["id"]))
        assert_bits(gn["dist"], on["dist"], "distances at iteration %d" % it)
        assert np.array_equal(g.read(engine.Memory.RID), o.rid)
    g.close()


# ---- the reference's one known-answer test on the device code (tests/testsICP.cpp:988-1052)

def test_reference_kat_through_the_hip_rotation_solvers(engine, oracle):
    """S[11], means[8] of the reference's ICP.icpPowerMethod test through icp_power_method (one wave of the rotation solvers
    the iteration's finalize runs): within the test's own 42000 eps (5.00679e-3) of its `svdTk` literal for the LITERAL and
    SQUARED power loops and for the EIGEN (SVD) branch; equal to the oracle bit for bit; LITERAL trip count as the oracle's,
    in the range the reference's comment gives (56 iterations on its device, :1027)."""
    import json
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.json")))["power_method"]
    S, means, svdTk = (np.array(kat[k], np.float32) for k in ("S", "means", "svdTk"))
    eps = kat["eps_vs_svd"]
    Tl, Rl, itl = engine.power_method(S, means, mode=engine.PowerMode.LITERAL)
    Tq, Rq, itq = engine.power_method(S, means, mode=engine.PowerMode.SQUARED)
    Te, Re, _ = engine.power_method(S, means, rot=engine.ICPStepConfigT.EIGEN)
    for name, T in (("literal", Tl), ("squared", Tq), ("eigen", Te)):
        assert np.all(np.abs(T - svdTk) < eps), (name, T, svdTk)
    ol, ol_it = oracle.power_method(S, means)
    oq, oq_it = oracle.power_method(S, means, fast=True)
    oR, oe = oracle.svd_rotation(S, means)
    assert_bits(Tl, ol, "literal power method vs oracle")
    assert_bits(Tq, oq, "squared-start power method vs oracle")
    assert_bits(Te, oe, "SVD branch vs oracle")
    assert_bits(Re, oR, "SVD rotation vs oracle")
    assert_bits(Rl, oracle.quat_to_rot(Tl[:4]), "rotation of qk")
    assert itl == ol_it and itq == oq_it
    assert 40 <= itl <= 70, itl
    # the reference CPU twin's own output on these inputs (cpuICPPowerMethod, recorded in SURVEY.md §8c): the test's GPU-vs-twin
    # bound of 420 eps (:1037), and the digits the oracle is pinned to
    twin = np.array(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.json")))["power_method_twin_output"]["Tk"], np.float32)
    assert np.all(np.abs(Tl - twin) < kat["eps_kernel_vs_twin"]) and np.allclose(Tl, twin, rtol=2e-7, atol=0)
    with pytest.raises(engine.ICPError):
        engine.power_method(S, means, rot=7)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_planar_scene_through_the_hip_rotation_solvers(engine, oracle, seed):
    """Coplanar points (the reference's wall scene at the solver, tests/test_oracle_golden.py::test_squared_power_method_on_a_planar_scene):
    the device's squared start takes the norm shift exactly as the oracle does — same bits, two passes — and the literal loop its 1000 trips."""
    from test_oracle_golden import _planar_case
    S, means, Rt = _planar_case(seed)
    Tq, Rq, itq = engine.power_method(S, means, mode=engine.PowerMode.SQUARED)
    Tl, Rl, itl = engine.power_method(S, means, mode=engine.PowerMode.LITERAL)
    oq, oq_it = oracle.power_method(S, means, fast=True)
    ol, ol_it = oracle.power_method(S, means)
    assert_bits(Tq, oq, "squared start, planar scene")
    assert_bits(Tl, ol, "literal loop, planar scene")
    assert (itq, itl) == (oq_it, ol_it) and itq == 2
    assert np.abs(Rq.reshape(3, 3) - Rt).max() < 1e-6


def test_first_search_seed_policy_does_not_change_a_bit(engine, oracle, monkeypatch):
    """A registration's first search (k = 0) is seeded from the queries' own grid cells; ICP_AMD_WARM_SEED=1 (diagnostics: what
    bench.py's `warm_seed_us_per_iteration` measures) keeps whatever the previous registration left in `rid`.  Any valid index is
    a legal seed: both handles, registering the same pair twice (the second time with the first run's converged nearest
    representatives lying around), give the oracle's bits — dense search, several tiles (256^2, 1024)."""
    side, nr = 256, 1024
    F, M = engine.synth_pair(side)
    o = oracle.OracleICP(side * side, nr, A, C_, threads=8, power_fast=True, fused=True)
    o.write_f(F); o.write_m(M); o.build_rbc()
    for _ in range(6):
        o.step()
    for warm in ("0", "1"):
        monkeypatch.setenv("ICP_AMD_WARM_SEED", warm)
        g = engine.ICP(0)
        g.init(side * side, nr, A, C_)
        g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
        for rep in range(2):
            g.buildRBC()
            g.run_fixed_fresh(6)
            check_step(engine, g, o, weighted=False)
        g.close()


@pytest.mark.parametrize("side,nr,fused", [(64, 64, True), (64, 64, False), (128, 256, True), (256, 1024, True)])
def test_nan_and_inf_points_do_not_break_the_search(engine, oracle, side, nr, fused):
    """Holes of a sensor are zeros in the reference (kernels/icp_kernels.cl:50-51); other pipelines leave NaN / inf.  NaN coordinates in
    the moving set, infinite ones in both: the RBC structure, the nearest representatives and every correspondence id still equal the
    oracle's (a NaN or infinite distance never wins a '<' on either side; the box pruning skips such coordinates), distances agree
    bit for bit — a query without any comparable candidate reports +inf and the first member of its list on both sides
     —; T turns NaN on both sides (garbage in, the same garbage out) —
    latency variant, dense variant with one and with several representative tiles."""
    m = side * side
    F, M = engine.synth_pair(side)
    rng = np.random.default_rng(5)
    for idx in rng.choice(m, 6, replace=False):
        M[idx, rng.integers(0, 7)] = np.nan
    for idx in rng.choice(m, 4, replace=False):
        F[idx, rng.integers(0, 7)] = np.inf
    M[9, 0] = -np.inf
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    set_modes(engine, g, power_fast=fused, fused=fused)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=fused, fused=fused)
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    assert np.array_equal(g.read(engine.Memory.RBC_OWNER), o.rbc_owner) and np.array_equal(g.read(engine.Memory.RBC_PERM), o.rbc_perm)
    assert np.array_equal(g.read(engine.Memory.RBC_N), o.rbc_N)
    g.step(); o.step()
    gn, on = g.read(engine.Memory.NN_ID), o.nn_id
    assert np.array_equal(g.read(engine.Memory.RID), o.rid)
    assert np.array_equal(gn["id"], on["id"])
    # (round 6: the oracle's scans start from +inf instead of from their first candidate's distance — "a NaN never wins a '<'" now holds for
    # the first candidate too —, so a query without any comparable candidate reports +inf on both sides: every distance bit for bit)
    assert 0 < np.count_nonzero(np.isinf(on["dist"])) <= 8 and not np.isnan(on["dist"]).any()
    assert_bits(gn["dist"], on["dist"], "distances")
    assert np.isnan(g.read(engine.Memory.T)).all() and np.isnan(o.T).all()
    g.close()


def test_random_parity_sweep(engine, oracle):
    """250 random cases of tools/diag/fuzz.py (fixed seed): landmark grids 6 .. 256 wide (also not multiples of 8), every valid
    representative count, alpha 0.5 / 200 / 10^4, both reduce modes, both power modes, both rotation solvers, REGULAR / WEIGHTED,
    10 % holes, batches of three — RBC structure, then three free-running steps or a checked run, every output bit for bit
    against the oracle.  (A 7-minute run of the same generator: 6926 cases, no difference — profiles/r03_fuzz.txt.)"""
    import fuzz as diag_fuzz
    n, fails = diag_fuzz.run(cases=250, seed=3, verbose=False)
    assert n == 250 and not fails, fails[:3]
