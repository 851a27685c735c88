"""Round 6.

Item 2 — invalid points at config C's size (|F|=|M|=2^20, |R|=4096): the MASKED + lanes-as-candidates search with the origin list read per wave
behind colour boxes, the Morton colour sort and the compaction by the last-arriving block of `k_reps_and_boxes` had timings and no oracle
comparison.  Here: the RBC structure and two free-running steps against the committed fixture (digests of ALL 2^20 ids, distances and nearest
representatives, T / S / means / sum of weights bit for bit: tests/golden/round6_vectors.npz, made by make_golden.py from the oracle) and
against the LIVE oracle on a 4096-query subset; the reductions from the GPU's own correspondences.  B `blobs30_rgb0` through run ().

Item 3 — members of a long list that repeat an earlier member are dropped from the SEARCH's view of the list (k_list_boxes): every
output must stay the exhaustive scan's, N / O / perm / XP the construction's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import A, C_, assert_bits, check_rbc, check_step, set_modes      # noqa: E402

pytestmark = pytest.mark.gpu


def _gold():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round6_vectors.npz"))


@pytest.mark.parametrize("name", ["scattered10", "blobs30", "blobs30_rgb0"])
def test_config_C_with_invalid_points(engine, oracle, name):
    from icp_amd import workloads as W
    gold = _gold()
    tag = "C_" + name
    side, nr = W.CONFIGS["C"]
    m = side * side
    F, M = W.holes_pair(engine, name, side)
    assert [int(np.count_nonzero((X[:, 0] == 0) & (X[:, 1] == 0) & (X[:, 2] == 0))) for X in (F, M)] == gold[tag + "_invalid"].tolist()
    g = engine.ICP(0)
    g.init(m, nr, W.A, W.C_)
    assert g.search_layout() == (1, 256, 1)                     # dense, 256-tiles (MASKED), stage 2 with lanes = candidates
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    g.buildRBC()
    Mem = engine.Memory
    N, O, perm, owner = g.read(Mem.RBC_N), g.read(Mem.RBC_O), g.read(Mem.RBC_PERM), g.read(Mem.RBC_OWNER)
    # structure invariants, then the fixture
    assert int(N.sum()) == m and np.array_equal(O, np.concatenate([[0], np.cumsum(N[:-1], dtype=np.uint64)]).astype(np.uint32))
    assert np.array_equal(np.sort(perm), np.arange(m, dtype=np.uint32))
    assert np.array_equal(owner[perm], np.repeat(np.arange(nr, dtype=np.uint32), N))
    starts = np.zeros(m, bool); starts[O[N > 0]] = True
    assert np.all((np.diff(perm.astype(np.int64)) > 0) | starts[1:])              # stable inside every list
    assert np.array_equal(N, gold[tag + "_N"]) and np.array_equal(O, gold[tag + "_O"])
    assert np.array_equal(W.ids_digest(perm), gold[tag + "_perm_digest"]) and np.array_equal(W.ids_digest(owner), gold[tag + "_owner_digest"])
    assert_bits(g.read(Mem.RBC_XP), F[perm], "XP")
    # live oracle on a subset: owners of fixed points, then the search of moving points
    R, src = oracle.get_reps(F, nr)
    assert_bits(g.read(Mem.REPS), R, "representatives")
    sub = gold["sub"].astype(np.int64)
    rbc = dict(owner=owner, N=N, O=O, perm=perm, XP=np.ascontiguousarray(F[perm]))
    assert np.array_equal(oracle.rbc_search(F[sub], R, rbc, src, W.A)[2], owner[sub])
    T = np.array([0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    for it in range(2):
        Q = oracle.transform_q(M[sub], T)                        # the queries the step is about to search (T of the previous step)
        g.step()
        nn, rid = g.read(Mem.NN_ID), g.read(Mem.RID)
        nn_o, _, rid_o = oracle.rbc_search(Q, R, rbc, src, W.A)
        assert np.array_equal(rid[sub], rid_o), "nearest representative (live oracle, subset), step %d" % it
        assert np.array_equal(nn["id"][sub], nn_o["id"]), "ids (live oracle, subset), step %d" % it
        assert_bits(nn["dist"][sub], nn_o["dist"], "distances (live oracle, subset)")
        assert np.array_equal(W.ids_digest(nn["id"]), gold[tag + "_ids_digest"][it]), "all 2^20 ids vs fixture, step %d" % it
        assert np.array_equal(W.ids_digest(rid), gold[tag + "_rid_digest"][it])
        assert np.array_equal(W.bits_digest(nn["dist"]), gold[tag + "_dist_digest"][it])
        T = g.read(Mem.T)
        assert_bits(T, gold[tag + "_T"][it], "T vs fixture, step %d" % it)
        assert_bits(g.read(Mem.S), gold[tag + "_S"][it], "S vs fixture")
        assert_bits(g.read(Mem.MEANS), gold[tag + "_means"][it], "means vs fixture")
        assert_bits(g.read(Mem.SUM_W), gold[tag + "_sum_w"][it:it + 1], "sum of weights vs fixture")
        if it == 0:
            # reductions from the GPU's own correspondences: the oracle's fused moments on them, the rotation solver on those
            Wt, sw = oracle.weights(nn)
            assert_bits(g.read(Mem.W), Wt, "weights")
            tM = oracle.transform_q(M, np.array([0, 0, 0, 1, 0, 0, 0, 1], np.float32))
            sw2, means, S = oracle.moments_fused(np.ascontiguousarray(F[nn["id"]]), tM, Wt, side, W.C_)
            assert_bits(g.read(Mem.SUM_W), np.array([sw2]), "sum of weights (the fused moments' double sum)")
            assert_bits(g.read(Mem.S), S, "S from the GPU's correspondences")
            assert_bits(g.read(Mem.MEANS), means, "means from the GPU's correspondences")
            Tk, _ = oracle.power_method(S, means, fast=True)
            assert_bits(g.read(Mem.TK), Tk, "Tk")
    g.close()


def test_config_B_degenerate_list_run(engine, oracle):
    """B (65536, 1024) with 30 % invalid points whose colour is zeroed too — ONE list of 19 673 identical points: ICP::run against the live
    oracle (k, T, every id and distance) and the fixture."""
    from icp_amd import workloads as W
    gold = _gold()
    side, nr = W.CONFIGS["B"]
    m = side * side
    F, M = W.holes_pair(engine, "blobs30_rgb0", side)
    g = engine.ICP(0)
    g.init(m, nr, W.A, W.C_)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, W.A, W.C_, power_fast=True, fused=True, threads=16)
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    check_rbc(engine, g, o)
    assert int(g.read(engine.Memory.RBC_N).max()) == int(gold["B_blobs30_rgb0_N_max"][0])
    kg, ko = g.run(), o.run()
    assert (kg, bool(g.state().converged)) == (ko, o.converged) == tuple(int(v) for v in gold["B_blobs30_rgb0_run"][:1]) + (bool(gold["B_blobs30_rgb0_run"][1]),)
    check_step(engine, g, o, weighted=False)
    assert_bits(g.read(engine.Memory.T), gold["B_blobs30_rgb0_run_T"], "final T vs fixture")
    assert np.array_equal(W.ids_digest(g.read(engine.Memory.NN_ID)["id"]), gold["B_blobs30_rgb0_run_ids_digest"])
    g.close()


def _with_repeats(F, rng, pattern, side):
    """A fixed set with bit-identical points planted in it (a copy): `runs` = contiguous index ranges filled with one point each, `scatter` =
    a third of the set drawn from five points, `alternate` = two points alternating over half of the set, `zeros` = the origin with +0 / -0
    coordinates and a colour of +0 / -0 (equal as numbers, different bits), `inf` / `nan` = repeats of a point with an infinite / a NaN coordinate (at distance +inf / NaN from every query:
    never a winner; the infinite ones equal one another and are legitimately dropped, NaN equals nothing and stays) beside repeats of an
    ordinary point — a quarter of the fixed set each, representatives among them."""
    F = F.copy()
    m = F.shape[0]
    if pattern == "runs":
        for _ in range(6):
            a = int(rng.integers(0, m - 1)); n = int(rng.integers(200, max(201, m // 5)))
            F[a:a + n] = F[int(rng.integers(0, m))]
    elif pattern == "scatter":
        src = F[rng.choice(m, 5, replace=False)].copy()
        idx = rng.choice(m, m // 3, replace=False)
        F[idx] = src[rng.integers(0, 5, idx.size)]
    elif pattern == "alternate":
        a, b = F[int(rng.integers(0, m))].copy(), F[int(rng.integers(0, m))].copy()
        idx = np.sort(rng.choice(m, m // 2, replace=False))
        F[idx[0::2]] = a; F[idx[1::2]] = b
    elif pattern == "zeros":
        idx = rng.choice(m, m // 3, replace=False)
        z = np.where(rng.integers(0, 2, (idx.size, 6)) == 1, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
        F[idx, 0:3] = z[:, 0:3]; F[idx, 4:7] = z[:, 3:6]
    elif pattern == "nan":
        p = F[int(rng.integers(0, m))].copy(); p[1] = np.nan
        F[rng.choice(m, m // 4, replace=False)] = p
        q = F[int(rng.integers(0, m))].copy()
        F[rng.choice(m, m // 4, replace=False)] = q
    elif pattern == "inf":
        p = F[int(rng.integers(0, m))].copy(); p[1] = np.inf
        F[rng.choice(m, m // 4, replace=False)] = p
        q = F[int(rng.integers(0, m))].copy()
        F[rng.choice(m, m // 4, replace=False)] = q
    return F


@pytest.mark.parametrize("side,nr,batch", [(128, 256, 1), (128, 256, 3), (256, 1024, 1), (256, 256, 1), (192, 2048, 1), (64, 4, 1), (96, 16, 2)])
@pytest.mark.parametrize("pattern", ["runs", "scatter", "alternate", "zeros", "inf", "nan"])
def test_repeated_points_in_long_lists(engine, oracle, side, nr, batch, pattern):
    """Bit-identical points planted in the fixed set (and some in the moving set): long lists whose tails the search's view drops.  Latency
    variant, dense variants (256-tiles, 1024-tile, batched), lanes = candidates; N / O / perm / owner / XP = the construction's; three steps
    with every per-query output against the oracle's exhaustive scans, then a rebuild on swapped frames."""
    m = side * side
    rng = np.random.default_rng(1000 * side + nr + len(pattern))
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=batch)
    pairs, orcs = [], []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=77 + 13 * b)
        F = _with_repeats(F, rng, pattern, side)
        if b % 2 == 0:
            M = _with_repeats(M, rng, "zeros" if pattern == "zeros" else "scatter", side)
        pairs.append((F, M))
    for rnd in range(2):
        orcs = []
        for b, (F, M) in enumerate(pairs):
            if rnd == 1:
                F, M = M, F
            g.write(engine.Memory.F, F, batch_index=b); g.write(engine.Memory.M, M, batch_index=b)
            g.write(engine.Memory.T, [0, 0, 0, 1, 0, 0, 0, 1], batch_index=b)
            o = oracle.OracleICP(m, nr, A, C_, threads=16, power_fast=True, fused=True)
            o.write_f(F); o.write_m(M); o.build_rbc()
            orcs.append((o, F))
        g.buildRBC()
        Mem = engine.Memory
        longest = 0
        for b, (o, F) in enumerate(orcs):
            assert np.array_equal(g.read(Mem.RBC_N, batch_index=b), o.rbc_N) and np.array_equal(g.read(Mem.RBC_O, batch_index=b), o.rbc_O)
            perm = g.read(Mem.RBC_PERM, batch_index=b)
            assert np.array_equal(perm, o.rbc_perm) and np.array_equal(g.read(Mem.RBC_OWNER, batch_index=b), o.rbc_owner)
            assert_bits(g.read(Mem.RBC_XP, batch_index=b), F[perm], "XP")
            longest = max(longest, int(o.rbc_N.max()))
        if rnd == 0 and pattern != "nan":
            assert longest > (1024 if g.search_layout()[2] else 256), longest          # (the case has a list with a tail)
        for it in range(3):
            g.step()
            for b, (o, F) in enumerate(orcs):
                o.step()
                gn = g.read(Mem.NN_ID, batch_index=b)
                assert np.array_equal(g.read(Mem.RID, batch_index=b), o.rid), (rnd, it, b)
                assert np.array_equal(gn["id"], o.nn_id["id"]), (rnd, it, b, int(np.count_nonzero(gn["id"] != o.nn_id["id"])))
                assert_bits(g.read(Mem.NN, batch_index=b)[:, :3], F[gn["id"]][:, :3], "matched points (the winner's own record)")
                assert_bits(gn["dist"], o.nn_id["dist"], "distances")
                if pattern in ("inf", "nan"):
                    # (non-finite coordinates — also in the FIXED set, representatives included: such a distance never wins a '<', a query
                    # without any comparable candidate reports +inf and its list's first member on both sides; T is then NaN on both sides)
                    assert np.array_equal(np.isnan(g.read(Mem.T, batch_index=b)), np.isnan(o.T))
                    if np.isnan(o.T).any():
                        continue
                assert_bits(g.read(Mem.T, batch_index=b), o.T, "T")
    g.close()
