"""CPU tests: the oracle against the reference's own known-answer literals and tolerances
(tests/golden/reference_kat.json), against exact index formulas, against float64 re-computations with
the reference's test tolerances, and against the committed golden vectors.  No GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))
EPS = np.finfo(np.float32).eps


def rng(seed):
    return np.random.default_rng(seed)


def cloud8(r, n, lo=-1000.0, hi=1000.0):
    P = r.uniform(lo, hi, (n, 8)).astype(np.float32)
    P[:, 3] = 1
    P[:, 7] = 1
    P[:, 4:7] = r.uniform(0, 1, (n, 3)).astype(np.float32)
    return P


# ---- reference known-answer tests -------------------------------------------------------------

def test_power_method_kat_vs_svd_literal(oracle):
    k = KAT["power_method"]
    Tk, it = oracle.power_method(k["S"], k["means"])
    assert np.all(np.abs(Tk - np.array(k["svdTk"], np.float32)) < k["eps_vs_svd"])        # tests/testsICP.cpp:1049-1051
    assert 40 <= it <= 70                                                                # "56 iterations" on the author's GPU
    Tf, itf = oracle.power_method(k["S"], k["means"], fast=True)
    assert np.all(np.abs(Tf - Tk) < k["eps_kernel_vs_twin"])                             # kernel-vs-twin tolerance :1037
    assert itf <= 10


def test_power_method_matches_recorded_twin_output(oracle):
    """The reference CPU twin's output on the KAT (recorded in SURVEY.md §8c) — same digits."""
    k = KAT["power_method"]
    Tk, _ = oracle.power_method(k["S"], k["means"])
    want = np.array(KAT["power_method_twin_output"]["Tk"], np.float32)
    assert np.allclose(Tk, want, rtol=2e-7, atol=0)


def test_svd_branch_agrees_with_literal(oracle):
    k = KAT["power_method"]
    Rk, Tk = oracle.svd_rotation(k["S"], k["means"])
    assert np.all(np.abs(Tk - np.array(k["svdTk"], np.float32)) < k["eps_vs_svd"])
    assert abs(np.linalg.det(Rk.astype(np.float64)) - 1) < 1e-5
    assert np.allclose(Rk @ Rk.T, np.eye(3), atol=1e-5)


def test_rotation_matrix_literal(oracle):
    """36.21 deg about (1,1,1)/sqrt(3): quaternion -> matrix must reproduce the test's matrix literal, and the
    quaternion and matrix transforms must agree (tests/testsICP.cpp:917-922, tolerance 42000 eps)."""
    k = KAT["transform_matrix"]
    th = np.deg2rad(k["angle_deg"])
    q = np.concatenate([np.sin(th / 2) * np.array(k["axis"]), [np.cos(th / 2)]]).astype(np.float32)
    R = oracle.quat_to_rot(q)
    assert np.all(np.abs(R.reshape(-1) - np.array(k["R"], np.float32)) < 2e-6)           # literal has 6 digits
    assert np.allclose(oracle.rot_to_quat(R), q, atol=1e-6)
    r = rng(1)
    M = cloud8(r, 4096, 0, 255)
    s, t = 0.7, np.array([12.0, -3.0, 200.0])
    T8 = np.concatenate([q, t, [s]]).astype(np.float32)
    T16 = np.eye(4, dtype=np.float32)
    T16[:3, :3] = s * R
    T16[:3, 3] = t
    a = oracle.transform_q(M, T8)
    b = oracle.transform_m(M, T16)
    assert np.all(np.abs(a - b)[:, :3] < k["eps"])
    assert np.array_equal(a[:, 3:], M[:, 3:])


def test_transform_quaternion_variants(oracle):
    """q = (0.5144, 0.5743, 0.5632, 0.2973) (tests/testsICP.cpp:821-822): both kernels agree within 4200 eps and
    match a float64 evaluation of p' = s R(q) p + t."""
    q = np.array(KAT["transform_quaternion"]["q"], np.float32)
    r = rng(2)
    M = cloud8(r, 16384, 0, 255)
    T = np.concatenate([q, r.uniform(0, 255, 3), r.uniform(0, 1, 1)]).astype(np.float32)
    a = oracle.transform_q(M, T, 1)
    b = oracle.transform_q(M, T, 2)
    assert np.all(np.abs(a - b) < 4200 * EPS * 16)            # the two kernels differ by the |q| != 1 of the 4-digit literal
    qd = q.astype(np.float64)
    v, w = qd[:3], qd[3]
    P = M[:, :3].astype(np.float64)
    ref = T[7] * (P + 2 * np.cross(v, np.cross(v, P) + w * P)) + T[4:7]
    assert np.all(np.abs(a[:, :3] - ref) < KAT["transform_quaternion"]["eps"])
    ident = oracle.transform_q(M, np.array([0, 0, 0, 1, 0, 0, 0, 1], np.float32))
    assert np.array_equal(ident, M)                           # T0 leaves the set bit-identical


# ---- exact index formulas ------------------------------------------------------------------------

def test_get_lms_index_formula(oracle):
    """getLMs: landmark (i, j) = pixel (col 65 + 4 i, row 49 + 3 j) — kernels/icp_kernels.cl:63-76."""
    cloud = np.arange(640 * 480 * 8, dtype=np.float32).reshape(480, 640, 8)
    lm = oracle.get_lms(cloud).reshape(128, 128, 8)
    for (j, i) in [(0, 0), (0, 127), (127, 0), (127, 127), (5, 77)]:
        assert np.array_equal(lm[j, i], cloud[49 + 3 * j, 65 + 4 * i])
    # literal float4 index of the kernel: out[gY*256 + gX] = in[(48+yi)*1280 + 128 + xi + gX%2]
    flat = cloud.reshape(-1, 4)
    out = lm.reshape(-1, 4)
    for gY, gX in [(0, 0), (3, 17), (127, 255), (64, 128)]:
        xi = (((gX >> 1) << 1) << 2) + 2
        yi = gY * 3 + 1
        assert np.array_equal(out[gY * 256 + gX], flat[(48 + yi) * 1280 + (128 + xi) + gX % 2])


@pytest.mark.parametrize("side,nr,shape", [(128, 256, (16, 16)), (128, 32, (8, 4)), (256, 1024, (32, 32)), (32, 16, (4, 4)), (6, 4, (2, 2))])
def test_get_reps_index_formula(oracle, side, nr, shape):
    """getReps: xi = gX*step + step/2 - 1 (kernels/icp_kernels.cl:107-113), nrx/nry from src/ICP/algorithms.cpp:852-854."""
    assert oracle.reps_grid(side * side, nr) == (shape[0], shape[1], side)
    F = np.arange(side * side * 8, dtype=np.float32).reshape(-1, 8)
    R, src = oracle.get_reps(F, nr)
    nrx, nry = shape
    sx, sy = side // nrx, side // nry
    for gY in range(nry):
        for gX in range(nrx):
            want = (gY * sy + sy // 2 - 1) * side + gX * sx + sx // 2 - 1
            assert src[gY * nrx + gX] == want
    assert np.array_equal(R, F[src])


def test_reps_grid_rejects(oracle):
    assert oracle.reps_grid(16384, 100) is None       # not a power of two
    assert oracle.reps_grid(1000, 4) is None          # not a square grid
    assert oracle.reps_grid(16, 64) is None           # more representatives than points


# ---- reductions against float64 with the reference's tolerances ---------------------------------------

def test_weights_and_sum(oracle):
    r = rng(3)
    nn = np.zeros(16384, oracle.DIST_ID)
    nn["dist"] = r.uniform(0, 1, 16384).astype(np.float32)
    W, sw = oracle.weights(nn)
    ref = 100.0 / (100.0 + nn["dist"].astype(np.float64))
    assert np.all(np.abs(W - ref) < 42 * EPS)                                  # tests/testsICP.cpp:262
    assert abs(sw - ref.sum()) < 4200 * EPS                                    # :263
    assert np.array_equal(W, (np.float32(100) / (np.float32(100) + nn["dist"])).astype(np.float32))


@pytest.mark.parametrize("n", [16384, 128, 2, 900, 65536])
def test_means(oracle, n):
    r = rng(4)
    F = cloud8(r, n, 0, 10000)
    M = cloud8(r, n, 0, 255)
    W = r.uniform(0, 1, n).astype(np.float32)
    sw = float(W.astype(np.float64).sum())
    mw = oracle.mean_weighted(F, M, W, sw)
    ref_f = (W[:, None].astype(np.float64) / sw * F[:, :3]).sum(0)
    ref_m = (W[:, None].astype(np.float64) / sw * M[:, :3]).sum(0)
    tol = 420000 * EPS                                                          # tests/testsICP.cpp:446
    assert np.all(np.abs(mw[:3] - ref_f) < tol) and np.all(np.abs(mw[4:7] - ref_m) < tol)
    assert mw[3] == 0 and mw[7] == 0
    mr = oracle.mean(F, M)
    assert np.all(np.abs(mr[:3] - F[:, :3].astype(np.float64).mean(0)) < tol)
    assert np.all(np.abs(mr[4:7] - M[:, :3].astype(np.float64).mean(0)) < tol)


def test_devs_and_sij(oracle):
    r = rng(5)
    m, c = 16384, 1e-6
    F = cloud8(r, m)
    M = cloud8(r, m)
    mean8 = oracle.mean(F, M)
    DF, DM = oracle.devs(F, M, mean8)
    assert np.array_equal(DF[:, :3], F[:, :3] - mean8[:3])                      # exact, tests/testsICP.cpp:556 (42 eps there)
    assert np.array_equal(DM[:, 3], M[:, 3] - mean8[7])
    W = r.uniform(0, 1, m).astype(np.float32)
    for Wt in (None, W):
        S = oracle.sij(DM, DF, Wt, c)
        w = np.ones(m) if Wt is None else Wt.astype(np.float64)
        Mp, Fp = c * DM[:, :3].astype(np.float64), c * DF[:, :3].astype(np.float64)
        ref = np.concatenate([np.einsum("i,ia,ib->ab", w, Mp, Fp).reshape(-1),
                              [(w * (Fp ** 2).sum(1)).sum(), (w * (Mp ** 2).sum(1)).sum()]])   # kernel order: [9] = f.f, [10] = m.m
        assert np.all(np.abs(S - ref) < 4200 * EPS)                             # tests/testsICP.cpp:637, 736


def test_reduce_sum_f(oracle):
    r = rng(6)
    a = r.uniform(0, 1, (11, 4096)).astype(np.float32)
    out = oracle.reduce_sum_f(a)
    assert np.all(np.abs(out - a.astype(np.float64).sum(1)) < 42000 * EPS)      # tests/testsReduce.cpp:252
    b = r.uniform(0, 1, (3, 1024 * 1024)).astype(np.float32)
    assert np.all(np.abs(oracle.reduce_sum_f(b) - b.astype(np.float64).sum(1)) < 1.0)


# ---- Random Ball Cover (parity unpinned: structural checks only) ---------------------------------------

def test_rbc_structure_and_search(oracle, engine):
    F, M = engine.synth_pair(64)
    R, src = oracle.get_reps(F, 64)
    rbc = oracle.rbc_construct(F, R, 2e2)
    N, O, perm, owner = rbc["N"], rbc["O"], rbc["perm"], rbc["owner"]
    assert N.sum() == F.shape[0] and np.array_equal(O, np.concatenate([[0], np.cumsum(N)[:-1]]))
    assert np.array_equal(np.sort(perm), np.arange(F.shape[0]))
    assert np.array_equal(owner[perm], np.repeat(np.arange(64), N))
    for r_ in range(64):                                       # stable: original order inside every list
        seg = perm[O[r_]:O[r_] + N[r_]]
        assert np.all(np.diff(seg.astype(np.int64)) > 0)
    assert np.all(owner[src] == np.arange(64))                 # a representative owns itself
    d = np.array([[oracle.metric8(F[i], R[r_], 2e2) for r_ in range(64)] for i in range(0, 4096, 97)])
    assert np.array_equal(owner[::97], d.argmin(1))
    nn, NN, rid = oracle.rbc_search(M, R, rbc, src, 2e2)
    assert np.array_equal(NN, F[nn["id"]])
    exact = oracle.nn_brute(M, F, 2e2)
    assert np.all(nn["dist"] >= exact["dist"])                 # one-shot RBC is approximate, never better than exact
    assert np.mean(nn["id"] == exact["id"]) > 0.85


def test_metric_definition(oracle):
    x = np.array([1, 2, 3, 1, .1, .2, .3, 1], np.float32)
    y = np.array([2, 4, 6, 9, .2, .4, .6, 7], np.float32)
    want = np.float32(np.float32(np.float32(1 + 4) + 9) + np.float32(200) * np.float32(np.float32(np.float32(.1) ** 2 + np.float32(.2) ** 2) + np.float32(.3) ** 2))
    assert abs(oracle.metric8(x, y, 200.0) - want) <= 4 * EPS * want      # lanes 3 and 7 ignored


# ---- pipeline -------------------------------------------------------------------------------------------

def test_golden_vectors(oracle, engine):
    g = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))
    F, M = engine.synth_pair(32)
    assert np.array_equal(F[:64], g["F_head"]) and np.array_equal(M[:64], g["M_head"])       # generator is deterministic
    assert np.allclose(F.astype(np.float64).sum(0), g["F_sum"]) and np.allclose(M.astype(np.float64).sum(0), g["M_sum"])
    o = oracle.OracleICP(1024, 16, 2e2, 1e-6)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    for k in ("rbc_N", "rbc_O", "rbc_perm", "rbc_owner"):
        assert np.array_equal(getattr(o, k), g[k])
    for it in range(5):
        o.step()
        assert np.array_equal(o.T, g["T"][it]) and np.array_equal(o.Tk, g["Tk"][it])
        assert np.array_equal(o.S, g["S"][it]) and np.array_equal(o.means, g["means"][it])
        assert o.sum_w == g["sum_w"][it]
        assert np.array_equal(o.nn_id["id"][:64], g["nn_id_head"][it])
        assert np.array_equal(o.nn_id["dist"][:64], g["nn_dist_head"][it])
    o.run()
    assert o.k == g["run_k"][0] and np.array_equal(o.T, g["run_T"])


@pytest.mark.parametrize("tag,side,nr", [("fs32", 32, 16), ("fs128", 128, 256)])
def test_golden_vectors_fused_squared(oracle, engine, tag, side, nr):
    """The bench's modes (single-pass double moments + squared-start power method) pinned the same way: four steps and
    a run to convergence reproduce the committed vectors bit for bit."""
    g = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))
    F, M = engine.synth_pair(side)
    o = oracle.OracleICP(side * side, nr, 2e2, 1e-6, power_fast=True, fused=True, threads=8)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    for it in range(4):
        o.step()
        assert np.array_equal(o.T, g[tag + "_T"][it]) and np.array_equal(o.S, g[tag + "_S"][it])
        assert np.array_equal(o.means, g[tag + "_means"][it])
        assert o.power_iters == g[tag + "_pm_iters"][it]
        assert np.array_equal(o.nn_id["id"][:64], g[tag + "_ids"][it])
    assert o.run() == g[tag + "_run_k"][0] and np.array_equal(o.T, g[tag + "_run_T"])


def test_config1_svd_path_recovers_motion(oracle, engine):
    """BASELINE config 1 (plumbing, no GPU): kg-like pair, |F|=|M|=16384, |R|=256, SVD rotation path."""
    F, M = engine.synth_pair(128)
    o = oracle.OracleICP(16384, 256, 2e2, 1e-6, rot=oracle.ROT_SVD, threads=8)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    k = o.run()
    assert 5 < k <= 40
    ax = np.array([0.3, 0.9, 0.1]) / np.linalg.norm([0.3, 0.9, 0.1])
    q_true = np.concatenate([-ax * np.sin(np.deg2rad(1.5)), [np.cos(np.deg2rad(1.5))]])     # T maps moving -> fixed: inverse motion
    assert np.abs(o.T[:4] - q_true).max() < 3e-3
    assert abs(o.T[7] - 1) < 2e-3
    p = oracle.OracleICP(16384, 256, 2e2, 1e-6, rot=oracle.ROT_POWER, threads=8)
    p.write_f(F)
    p.write_m(M)
    p.build_rbc()
    p.run()
    assert np.abs(p.T[:4] - o.T[:4]).max() < 1e-4 and np.abs(p.T[4:7] - o.T[4:7]).max() < 0.05


def test_power_start_variants_agree_in_free_run(oracle, engine):
    F, M = engine.synth_pair(64)
    res = []
    for fast in (False, True):
        o = oracle.OracleICP(4096, 64, 2e2, 1e-6, power_fast=fast, threads=8)
        o.write_f(F)
        o.write_m(M)
        o.build_rbc()
        o.run()
        res.append((o.k, o.T))
    assert abs(res[0][0] - res[1][0]) <= 1
    assert np.abs(res[0][1][:4] - res[1][1][:4]).max() < 1e-5
    assert np.abs(res[0][1][4:7] - res[1][1][4:7]).max() < 1e-5 * 1500      # relative to the scene scale (mm)


def test_write_t_continues_from_given_transform(oracle, engine):
    F, M = engine.synth_pair(32)
    a = oracle.OracleICP(1024, 16, 2e2, 1e-6)
    a.write_f(F); a.write_m(M); a.build_rbc()
    a.step(); a.step()
    T2 = a.T
    b = oracle.OracleICP(1024, 16, 2e2, 1e-6)
    b.write_f(F); b.write_m(M); b.build_rbc()
    b.write_t(T2)
    a.step(); b.step()
    assert np.array_equal(a.nn_id["id"], b.nn_id["id"])       # same correspondences from the same T
    assert np.allclose(a.T, b.T, rtol=1e-4, atol=1e-4)        # R is re-derived from q in b


def test_init_rejects(oracle):
    for bad in [(0, 4), (16, 0), (15, 4), (16, 3)]:
        with pytest.raises(ValueError):
            oracle.OracleICP(bad[0], bad[1])
    with pytest.raises(ValueError):
        oracle.OracleICP(16, 4, a=0.0)


def test_fused_mode_agrees_with_reference_order(oracle, engine):
    """DESIGN.md §3 item 8; docs/HISTORY.md §3.11: the single-pass double-moment formulation gives the reference-order means / S / T to
    within fp32 rounding of the coordinates, the same iteration count and the same final correspondences."""
    F, M = engine.synth_pair(64)
    res = []
    for fused in (False, True):
        o = oracle.OracleICP(4096, 64, 2e2, 1e-6, threads=8, fused=fused)
        o.write_f(F); o.write_m(M); o.build_rbc()
        o.step()
        first = (o.S, o.means, o.sum_w, o.nn_id["id"])
        k = o.run()
        res.append((first, k, o.T, o.nn_id["id"]))
    (S0, m0, sw0, id0), k0, T0, idf0 = res[0]
    (S1, m1, sw1, id1), k1, T1, idf1 = res[1]
    assert np.array_equal(id0, id1)                                       # same T0 => same correspondences
    assert abs(sw0 - sw1) < 4200 * EPS                                    # tests/testsICP.cpp:263
    assert np.all(np.abs(S0 - S1) <= 4200 * EPS * np.abs(S0).max())      # :736
    assert np.all(np.abs(m0 - m1) < 420000 * EPS)                         # :446
    assert abs(k0 - k1) <= 1
    scale = float(np.abs(F[:, :3]).max())                                 # 1e-5 relative to the magnitudes involved
    assert np.abs(T0[:4] - T1[:4]).max() < 1e-5 and np.abs(T0[4:7] - T1[4:7]).max() < 1e-5 * scale
    assert abs(T0[7] - T1[7]) < 1e-5
    assert np.mean(idf0 == idf1) > 0.999


def test_fused_moments_against_float64(oracle):
    r = rng(9)
    n, c = 4096, 1e-6
    NN = cloud8(r, n, 0, 2000); Q = cloud8(r, n, 0, 2000); W = r.uniform(0.2, 1, n).astype(np.float32)
    sw, means, S = oracle.moments_fused(NN, Q, W, 64, c)
    w = W.astype(np.float64)
    f, q = NN[:, :3].astype(np.float64), Q[:, :3].astype(np.float64)
    mf, mq = (w[:, None] * f).sum(0) / w.sum(), (w[:, None] * q).sum(0) / w.sum()
    Sref = c * c * np.einsum("i,ia,ib->ab", w, q - mq, f - mf).reshape(-1)
    assert abs(sw - w.sum()) < 1e-9 * w.sum()
    assert np.all(np.abs(means[:3] - mf) <= EPS * np.abs(mf)) and np.all(np.abs(means[4:7] - mq) <= EPS * np.abs(mq))
    assert np.all(np.abs(S[:9] - Sref) <= 4 * EPS * np.abs(Sref).max())
    assert abs(S[9] - c * c * (w * ((f - mf) ** 2).sum(1)).sum()) <= 4 * EPS * S[9]
    assert abs(S[10] - c * c * (w * ((q - mq) ** 2).sum(1)).sum()) <= 4 * EPS * S[10]


@pytest.mark.parametrize("i", [0, 36])
def test_config4_fixture_pins_the_oracle(oracle, engine, i):
    """tests/golden/config4_vectors.npz (BASELINE config 4's real shape; script tests/golden/make_golden.py): the oracle
    reproduces it (two of the eight checked registrations here, to keep the CPU suite short; the GPU test checks all)."""
    from icp_amd import workloads as C4
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_vectors.npz"))
    assert i in gold["checked"]
    F, M = C4.pair(engine, i)
    o = oracle.OracleICP(C4.M_POINTS, C4.NR, C4.A, C4.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    k = o.run()
    assert (k, int(o.converged)) == tuple(int(v) for v in gold["r%d_run" % i])
    assert np.array_equal(o.T.view(np.uint32), gold["r%d_run_T" % i].view(np.uint32))
    ids = o.nn_id["id"]
    assert np.array_equal(ids[:256], gold["r%d_run_ids_head" % i])
    assert np.array_equal(C4.ids_digest(ids), gold["r%d_run_ids_digest" % i])


def _round5():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round5_vectors.npz"))


@pytest.mark.parametrize("name", ["blobs10", "scattered10_rgb0"])
def test_holes_fixture_pins_the_oracle(oracle, engine, name):
    """tests/golden/round5_vectors.npz (make_golden.py: round5): the benchmark pair with a Kinect frame's invalid points — colour kept
    (reference src/kinect_frame_grabber.cpp:246-262) and, the degenerate case, zeroed — : the oracle reproduces the fixture (two of the
    six cases here; the GPU test checks the engine against all of them)."""
    from icp_amd import workloads as W
    gold = _round5()
    F, M = W.holes_pair(engine, name)
    o = oracle.OracleICP(W.M_POINTS, W.NR, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    assert int(o.rbc_N.max()) == int(gold[name + "_N_max"][0])
    k = o.run()
    assert (k, int(o.converged)) == tuple(int(v) for v in gold[name + "_run"])
    assert np.array_equal(o.T.view(np.uint32), gold[name + "_run_T"].view(np.uint32))
    assert np.array_equal(W.ids_digest(o.nn_id["id"]), gold[name + "_run_ids_digest"])


def test_wall_scene_needs_the_photometric_term(oracle, engine):
    """The reference's second example pair (data/kg_pc8d_wall, data/README.md:11-16): "non-salient surface geometry. It highlights the
    benefit of utilizing the photometric information. To see what happens in the absence of color, change the a parameter ... to a really
    small strictly positive number."  Stand-in: a textured plane moved IN its plane (2 degrees about its normal).  With a = 2e2 the
    registration converges and finds the rotation; with a = 1e-6 it slides: after 300 iterations it has neither converged nor found
    it, and the reference's scale estimate has collapsed.  Both runs are in the fixture."""
    from icp_amd import workloads as W
    gold = _round5()
    F, M, Tt = W.wall_pair(engine)
    assert np.array_equal(Tt, gold["wall_T_true"])
    res = {}
    for tag, a in (("wall_a2e2", W.A), ("wall_asmall", W.WALL_A_SMALL)):
        o = oracle.OracleICP(W.M_POINTS, W.NR, a, W.C_, power_fast=True, fused=True, threads=8, max_iterations=W.WALL_MAX_ITERATIONS)
        o.write_f(F); o.write_m(M); o.build_rbc()
        k = o.run()
        assert (k, int(o.converged)) == tuple(int(v) for v in gold[tag + "_run"])
        assert np.array_equal(o.T.view(np.uint32), gold[tag + "_run_T"].view(np.uint32))
        assert np.array_equal(W.ids_digest(o.nn_id["id"]), gold[tag + "_run_ids_digest"])
        res[tag] = (k, o.converged, W.rotation_error_deg(o.T, Tt), float(o.T[7]))
    k, conv, err, s = res["wall_a2e2"]
    assert conv and k < 200 and err < 0.1 and s > 0.98            # 2 degrees of in-plane rotation found to 0.1 degree
    k, conv, err, s = res["wall_asmall"]
    assert not conv and k == W.WALL_MAX_ITERATIONS and err > 3 * res["wall_a2e2"][2] and s < 0.97


def _planar_case(seed, tilt=0.3):
    """S[11], means[8] of an exactly planar pair: points on a tilted plane, the moving set rotated by 7 degrees about the plane's normal
    (S of rank 2: Horn's N has the eigenvalue pairs +-(s1 + s2), +-(s1 - s2))."""
    rng = np.random.default_rng(seed)
    uv = rng.uniform(-1, 1, (400, 2)) * np.array([300.0, 200.0])
    e1 = np.array([1.0, 0.0, tilt]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(np.array([0.2, 1.0, 0.1]), e1); e2 /= np.linalg.norm(e2)
    n = np.cross(e1, e2)
    f = uv[:, :1] * e1 + uv[:, 1:] * e2
    th = np.radians(7.0)
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    q = f @ R.T                                            # moving = R fixed: the solver must find R^T
    c = 1e-3
    S9 = (c * q).T @ (c * f)                               # S_ab = sum m_a f_b
    S = np.concatenate([S9.ravel(), [((c * f) ** 2).sum(), ((c * q) ** 2).sum()]]).astype(np.float32)
    means = np.zeros(8, np.float32)
    return S, means, R.T


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_squared_power_method_on_a_planar_scene(oracle, seed):
    """The reference's kg_pc8d_wall case (data/README.md:11-16) at the solver: for coplanar points Horn's matrix has eigenvalues in
    +- pairs, which no power method separates (the literal loop spends all its 1000 trips: kernels/icp_kernels.cl:1012-1022); the
    squared start shifts the matrix by its largest absolute row sum once when its first pass does not converge and finds the rotation
    in two passes.  Ordinary scenes never take that branch (every other fixture is unchanged)."""
    S, means, Rt = _planar_case(seed)
    Tq, itq = oracle.power_method(S, means, fast=True)
    assert itq == 2                                        # the first pass does not converge (+- pairs), the shifted one does
    assert np.abs(oracle.quat_to_rot(Tq[:4]) - Rt).max() < 1e-6


def test_wall_scene_power_method_trip_counts(oracle, engine):
    """S and the means of the wall pair's first iterations (icp_amd/workloads.py: wall_pair — a plane with a millimetre of roughness
    and noise): the literal loop runs into its 1000-trip limit every time (its vector is still a mixture of the +lambda and -lambda
    eigenvectors), the squared start needs its two passes; both are what the engine's modes run per ICP iteration on that scene."""
    from icp_amd import workloads as W
    F, M, Tt = W.wall_pair(engine)
    o = oracle.OracleICP(W.M_POINTS, W.NR, W.A, W.C_, threads=8, power_fast=True, fused=True)
    o.write_f(F); o.write_m(M); o.build_rbc()
    for it in range(3):
        o.step()
        assert o.power_iters == 2
        Tl, itl = oracle.power_method(o.S, o.means)
        assert itl >= 1000
        Tq, itq = oracle.power_method(o.S, o.means, fast=True)
        assert itq == 2 and np.array_equal(Tq.view(np.uint32), o.Tk.view(np.uint32))


def test_host_sanitizer_build_is_clean():
    """`make asan`: the oracle + the synthetic generator under AddressSanitizer / UBSan over every oracle
    entry point at small, ragged and degenerate sizes; any report aborts the program."""
    import subprocess
    root = os.path.dirname(HERE)
    r = subprocess.run(["make", "-C", root, "-s", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "asan_host: clean" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_oracle_bits_do_not_depend_on_the_thread_count(oracle, engine):
    """bench.py's cpu_baseline sweeps the OpenMP team size: the parallel loops are over independent queries and over the independent
    64-pair blocks of the moment reduction (fixed trees), so every team size gives the same bits — both modes, whole runs."""
    F, M = engine.synth_pair(128)
    for kw in (dict(power_fast=True, fused=True), dict()):
        res = []
        for t in (1, 3, 8):
            o = oracle.OracleICP(16384, 256, 2e2, 1e-6, threads=t, **kw)
            o.write_f(F); o.write_m(M); o.build_rbc()
            k = o.run()
            res.append((k, o.T.tobytes(), o.S.tobytes(), o.means.tobytes(), o.nn_id["id"].tobytes(), o.nn_id["dist"].tobytes()))
        assert res[0] == res[1] == res[2]


def test_cpu_baseline_reports_a_thread_sweep(engine):
    """bench.py's cpu_baseline: a bounded thread sweep, the best count as `value`, the host's core count beside it."""
    import bench
    F, M = engine.synth_pair(64)
    r = bench.cpu_baseline(F, M, 4096, 64, True, budget_s=1.0)
    assert r["kind"] == "port" and r["host_cores"] == os.cpu_count() and r["unit"] == "iterations/s"
    assert [x["threads"] for x in r["sweep"]] == sorted({t for t in (1, 8, 16, 64, os.cpu_count()) if t <= os.cpu_count()})
    assert r["value"] == max(x["iterations_per_s"] for x in r["sweep"]) and r["cores"] in [x["threads"] for x in r["sweep"]]
