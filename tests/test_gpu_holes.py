"""Invalid points of real captures: a Kinect frame's pixels without depth are points at the origin with
their colour kept (reference src/kinect_frame_grabber.cpp:246-262), getLMs picks them on purpose (kernels/icp_kernels.cl:49-50), and
the one-shot search then has (a) representatives at the origin, which the stage-1 pruning keeps out of its boxes and scans as a list
of their own, and (b) — with the colours zeroed too — ONE list that holds every invalid point, which stage 2 scans behind chunk
boxes.  Both are exact: everything here is compared with the oracle's serial scans bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import A, C_, assert_bits, check_rbc, check_step, set_modes      # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(engine, name, side, seed=None):
    from icp_amd import workloads as W
    return W.holes_pair(engine, name, side, seed=W.BASE_SEED if seed is None else seed)


@pytest.mark.parametrize("name", ["scattered10", "blobs10", "blobs30", "scattered10_rgb0", "blobs10_rgb0", "blobs30_rgb0"])
def test_holes_at_config_A_steps_and_run(engine, oracle, name):
    """Config A (16384, 256), the benchmarked modes: RBC structure, two free-running steps (every per-query and per-iteration output),
    then ICP::run — k, T and all 16384 correspondence ids."""
    side, nr = 128, 256
    m = side * side
    F, M = _pair(engine, name, side)
    holes = int(np.count_nonzero((F[:, 0] == 0) & (F[:, 1] == 0) & (F[:, 2] == 0)))
    assert holes > 0.08 * m
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=True, fused=True)
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    check_rbc(engine, g, o)
    if name.endswith("_rgb0"):
        assert o.rbc_N.max() >= holes - 1           # one list holds every invalid point (an invalid representative owns them all)
    for it in range(2):
        g.step(); o.step()
        check_step(engine, g, o)
    g.reset_transform(); g.buildRBC()
    o.write_t([0, 0, 0, 1, 0, 0, 0, 1]); o.build_rbc()
    k, ko = g.run(), o.run()
    assert k == ko
    assert_bits(g.read(engine.Memory.T), o.T, "T")
    assert np.array_equal(g.read(engine.Memory.NN_ID)["id"], o.nn_id["id"])
    assert_bits(g.read(engine.Memory.NN_ID)["dist"], o.nn_id["dist"], "distances")
    g.close()


@pytest.mark.parametrize("name", ["blobs10", "blobs30_rgb0"])
def test_holes_reference_order_mode(engine, oracle, name):
    side, nr = 128, 256
    m = side * side
    F, M = _pair(engine, name, side)
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    set_modes(engine, g, power_fast=False, fused=False)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, threads=8)
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    for it in range(2):
        g.step(); o.step()
        check_step(engine, g, o)
    g.close()


@pytest.mark.parametrize("fused", [True, False])
def test_holes_dense_batched(engine, oracle, fused):
    """The dense search (several blocks per CU, exact stage-1 pruning): six registrations of config A's shape in one handle, one per
    hole case — the representatives at the origin are scanned as a list of their own by the queries near the origin."""
    from icp_amd import workloads as W
    side, nr = 128, 256
    m = side * side
    names = list(W.HOLES)
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=len(names))
    set_modes(engine, g, power_fast=fused, fused=fused)
    assert g.search_layout()[0] == 1
    pairs = [_pair(engine, n, side, seed=W.BASE_SEED + 3 * b) for b, n in enumerate(names)]
    for b, (F, M) in enumerate(pairs):
        g.write(engine.Memory.F, F, batch_index=b); g.write(engine.Memory.M, M, batch_index=b)
    g.buildRBC()
    orcs = []
    for F, M in pairs:
        o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=fused, fused=fused)
        o.write_f(F); o.write_m(M); o.build_rbc()
        orcs.append(o)
    for it in range(3):
        g.step()
        for b, o in enumerate(orcs):
            o.step()
            assert np.array_equal(g.read(engine.Memory.RID, batch_index=b), o.rid), (it, b)
            gn = g.read(engine.Memory.NN_ID, batch_index=b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (it, b, names[b])
            assert_bits(gn["dist"], o.nn_id["dist"], "distances")
            assert_bits(g.read(engine.Memory.T, batch_index=b), o.T, "T of %s" % names[b])
    g.close()


@pytest.mark.parametrize("side,nr,name", [(256, 1024, "scattered10"), (256, 1024, "blobs30_rgb0"), (256, 256, "blobs10"), (256, 256, "scattered10_rgb0"),
                                          (192, 2048, "blobs30"), (256, 4096, "scattered10_rgb0")])
def test_holes_dense_layouts(engine, oracle, side, nr, name):
    """Config B (several 256-tiles, tile masks), a set with long lists (lanes = candidates in stage 2) and one on the 1024-tile variant."""
    m = side * side
    F, M = _pair(engine, name, side)
    g = engine.ICP(0)
    g.init(m, nr, A, C_)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    o = oracle.OracleICP(m, nr, A, C_, threads=16, power_fast=True, fused=True)
    o.write_f(F); o.write_m(M)
    g.buildRBC(); o.build_rbc()
    check_rbc(engine, g, o)
    for it in range(3):
        g.step(); o.step()
        check_step(engine, g, o)
    g.close()


def test_holes_large_sets_batched_and_rebuilt(engine, oracle):
    """More than 1024 representatives: their blocks of k_reps_and_boxes leave a ballot each and the one that arrives last lists the
    representatives at the origin (an arrival counter per registration of the handle, back at zero for the next construction) —
    three registrations with different hole patterns in one handle, constructed twice with the frames swapped in between."""
    side, nr = 192, 2048
    m = side * side
    names = ["blobs30", "clean", "scattered10_rgb0"]
    pairs = [engine.synth_pair(side, seed=91 + 5 * b) if n == "clean" else _pair(engine, n, side, seed=91 + 5 * b) for b, n in enumerate(names)]
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=len(names))
    orcs_last = []
    for rnd in range(2):
        order = list(range(len(names))) if rnd == 0 else [2, 0, 1]
        for b, j in enumerate(order):
            g.write(engine.Memory.F, pairs[j][0], batch_index=b); g.write(engine.Memory.M, pairs[j][1], batch_index=b)
            g.write(engine.Memory.T, [0, 0, 0, 1, 0, 0, 0, 1], batch_index=b)
        g.buildRBC()
        for b, j in enumerate(order):
            o = oracle.OracleICP(m, nr, A, C_, threads=16, power_fast=True, fused=True)
            o.write_f(pairs[j][0]); o.write_m(pairs[j][1]); o.build_rbc()
            assert np.array_equal(g.read(engine.Memory.RBC_OWNER, batch_index=b), o.rbc_owner), (rnd, b)
            assert np.array_equal(g.read(engine.Memory.RBC_N, batch_index=b), o.rbc_N), (rnd, b)
            assert np.array_equal(g.read(engine.Memory.RBC_PERM, batch_index=b), o.rbc_perm), (rnd, b)
            o.step()
            orcs_last = orcs_last + [o] if b else [o]
        g.step()
        for b, o in enumerate(orcs_last):
            assert np.array_equal(g.read(engine.Memory.NN_ID, batch_index=b)["id"], o.nn_id["id"]), (rnd, b)
            assert_bits(g.read(engine.Memory.T, batch_index=b), o.T, "T")
    g.close()


@pytest.mark.parametrize("side,nr,batch,ff,fm", [(128, 256, 4, 1.0, 0.9), (128, 256, 4, 0.9, 1.0), (256, 1024, 1, 1.0, 1.0), (256, 1024, 1, 0.95, 0.5),
                                                 (192, 2048, 1, 1.0, 0.3), (128, 256, 1, 1.0, 0.9), (128, 256, 1, 0.95, 1.0)])
def test_holes_extreme_fractions_dense(engine, oracle, side, nr, batch, ff, fm):
    """Frames that are (nearly) nothing but invalid points, dense layouts (and the latency-bound one: batch 1 at 16384): every representative at the origin (all pruning boxes empty, the
    origin list as long as the set: sorted, staged or read per wave by its length), no valid representative for the seeds to fall back on,
    blocks whose 64 queries are all handed over as invalid ones."""
    m = side * side
    rng = np.random.default_rng(side * nr + batch)

    def punch(cloud, frac, seed):
        c = cloud.reshape(-1, 8).copy()
        if frac >= 1.0:
            c[:, 0:3] = 0.0
        else:
            c[rng.random(m) < frac, 0:3] = 0.0
        return c

    pairs = []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=4242 + 17 * b)
        pairs.append((punch(F, ff, b), punch(M, fm, b)))
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=batch)
    for b, (F, M) in enumerate(pairs):
        g.write(engine.Memory.F, F, batch_index=b); g.write(engine.Memory.M, M, batch_index=b)
    g.buildRBC()
    orcs = []
    for F, M in pairs:
        o = oracle.OracleICP(m, nr, A, C_, threads=16, power_fast=True, fused=True)
        o.write_f(F); o.write_m(M); o.build_rbc()
        orcs.append(o)
    for b, o in enumerate(orcs):
        assert np.array_equal(g.read(engine.Memory.RBC_OWNER, batch_index=b), o.rbc_owner), b
        assert np.array_equal(g.read(engine.Memory.RBC_PERM, batch_index=b), o.rbc_perm), b
    for it in range(2):
        g.step()
        for b, o in enumerate(orcs):
            o.step()
            assert np.array_equal(g.read(engine.Memory.RID, batch_index=b), o.rid), (it, b)
            gn = g.read(engine.Memory.NN_ID, batch_index=b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (it, b)
            assert_bits(gn["dist"], o.nn_id["dist"], "distances")
            gT = g.read(engine.Memory.T, batch_index=b)
            if np.isnan(o.T).any():          # (a fixed set of nothing but invalid points: S = 0, no rotation to find — NaN in the same places; payloads are not compared)
                assert np.array_equal(np.isnan(gT), np.isnan(o.T)) and np.array_equal(gT[~np.isnan(gT)], o.T[~np.isnan(o.T)]), (it, b)
                break
            assert_bits(gT, o.T, "T")
        else:
            continue
        break
    g.close()


@pytest.mark.parametrize("warm", [False, True])
def test_holes_tracked_sequence(engine, oracle, warm):
    """Frame-to-frame tracking on 640 x 480 frames with contiguous invalid regions (10 - 30 %, another pattern in every frame; one frame with
    the colours zeroed as well): every hop's k and T equal the oracle's."""
    pats = [(1, 0.1, True), (1, 0.3, True), (0, 0.1, True), (1, 0.2, False)]
    clouds = [engine.punch_holes(engine.synth_cloud_vga(moved=f), 640, 480, p, fr, keep, seed=77 + f) for f, (p, fr, keep) in enumerate(pats)]
    order = [0, 1, 2, 3, 2, 1, 0]
    lms = [oracle.get_lms(c) for c in clouds]
    assert all(np.count_nonzero(l[:, 2] == 0) > 1000 for l in lms)
    g = engine.ICP(0)
    g.init(16384, 256, A, C_)
    res = g.track_pipelined([clouds[i] for i in order], warm_start=warm, depth=2)
    o = oracle.OracleICP(16384, 256, A, C_, threads=8, power_fast=True, fused=True)
    for i in range(1, len(order)):
        o.write_f(lms[order[i - 1]]); o.write_m(lms[order[i]])
        o.write_t(o.T if (warm and i > 1) else [0, 0, 0, 1, 0, 0, 0, 1])
        o.build_rbc()
        ko = o.run()
        k, T = res[i]
        assert k == ko, (i, k, ko)
        assert np.array_equal(T.view(np.uint32), o.T.view(np.uint32)), i
    assert np.array_equal(g.read(engine.Memory.NN_ID)["id"], o.nn_id["id"])
    g.close()


@pytest.mark.parametrize("batch", [1, 5])
def test_long_lists_of_distinct_points(engine, oracle, batch):
    """Long lists that are NOT piles of identical points: the representatives of one quarter of the fixed frame — all but one — are
    moved far away (each to a place of its own), so the quarter's points fall to the one that stays and to the representatives
    around the quarter: lists of up to ~1100 distinct candidates with the scene's geometry and colours, where the chunk boxes have to
    discriminate (latency variant, and the dense one through a batch).  Everything against the serial scan."""
    side, nr = 128, 256
    m = side * side
    g = engine.ICP(0)
    g.init(m, nr, A, C_, batch=batch)
    orcs = []
    for b in range(batch):
        F, M = engine.synth_pair(side, seed=0x1C9D5EED + b)
        R, rep_src = oracle.get_reps(F, nr)
        yy, xx = np.divmod(rep_src, side)
        move = rep_src[(yy < 64) & (xx < 64)][1:]
        F = F.copy()
        F[move, 2] += 50000 + np.arange(move.size, dtype=np.float32) * 100
        g.write(engine.Memory.F, F, batch_index=b); g.write(engine.Memory.M, M, batch_index=b)
        o = oracle.OracleICP(m, nr, A, C_, threads=8, power_fast=True, fused=True)
        o.write_f(F); o.write_m(M); o.build_rbc()
        assert o.rbc_N.max() > 512
        orcs.append(o)
    g.buildRBC()
    for it in range(3):
        g.step()
        for b, o in enumerate(orcs):
            o.step()
            gn = g.read(engine.Memory.NN_ID, batch_index=b)
            assert np.array_equal(gn["id"], o.nn_id["id"]), (it, b)
            assert_bits(gn["dist"], o.nn_id["dist"], "distances")
            assert_bits(g.read(engine.Memory.T, batch_index=b), o.T, "T")
    g.close()


@pytest.mark.parametrize("name", ["scattered10", "blobs10", "blobs30", "scattered10_rgb0", "blobs10_rgb0", "blobs30_rgb0"])
def test_holes_engine_against_the_committed_fixture(engine, name):
    """tests/golden/round5_vectors.npz (oracle outputs, make_golden.py): ICP::run and the bench's 40-iteration fixed pass on the six hole
    cases at config A — k, convergence, T bit for bit, a digest of all 16384 correspondence ids — without the oracle in the loop."""
    from icp_amd import workloads as W
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round5_vectors.npz"))
    F, M = W.holes_pair(engine, name)
    g = engine.ICP(0)
    g.init(W.M_POINTS, W.NR, W.A, W.C_)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    g.buildRBC()
    assert int(g.read(engine.Memory.RBC_N).max()) == int(gold[name + "_N_max"][0])
    k = g.run()
    assert (k, int(g.state().converged)) == tuple(int(v) for v in gold[name + "_run"])
    assert_bits(g.read(engine.Memory.T), gold[name + "_run_T"], "T")
    assert np.array_equal(W.ids_digest(g.read(engine.Memory.NN_ID)["id"]), gold[name + "_run_ids_digest"])
    g.run_fixed_fresh(40)
    assert_bits(g.read(engine.Memory.T), gold[name + "_fixed40_T"], "T after 40 fixed iterations")
    assert np.array_equal(W.ids_digest(g.read(engine.Memory.NN_ID)["id"]), gold[name + "_fixed40_ids_digest"])
    g.close()


@pytest.mark.parametrize("tag", ["wall_a2e2", "wall_asmall"])
def test_wall_scene_engine_equals_oracle_and_fixture(engine, oracle, tag):
    """The reference's second example pair (data/kg_pc8d_wall: geometry that does not constrain the motion, data/README.md:11-16), stand-in:
    the engine's run with max_iterations = 300 equals the oracle's bit for bit — k, convergence, T, every correspondence id — with the
    photometric term (a = 2e2: converges, finds the in-plane rotation) and without it (a = 1e-6: 300 iterations, neither)."""
    from icp_amd import workloads as W
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round5_vectors.npz"))
    a = W.A if tag == "wall_a2e2" else W.WALL_A_SMALL
    F, M, Tt = W.wall_pair(engine)
    g = engine.ICP(0)
    g.init(W.M_POINTS, W.NR, a, W.C_, max_iterations=W.WALL_MAX_ITERATIONS)
    g.write(engine.Memory.F, F); g.write(engine.Memory.M, M)
    g.buildRBC()
    k = g.run()
    o = oracle.OracleICP(W.M_POINTS, W.NR, a, W.C_, threads=8, power_fast=True, fused=True, max_iterations=W.WALL_MAX_ITERATIONS)
    o.write_f(F); o.write_m(M); o.build_rbc()
    assert k == o.run() and bool(g.state().converged) == bool(o.converged)
    assert (k, int(o.converged)) == tuple(int(v) for v in gold[tag + "_run"])
    assert_bits(g.read(engine.Memory.T), o.T, "T")
    assert_bits(g.read(engine.Memory.T), gold[tag + "_run_T"], "T (fixture)")
    n = g.read(engine.Memory.NN_ID)
    assert np.array_equal(n["id"], o.nn_id["id"])
    assert_bits(n["dist"], o.nn_id["dist"], "distances")
    err = W.rotation_error_deg(g.read(engine.Memory.T), Tt)
    assert (err < 0.1) if tag == "wall_a2e2" else (err > 0.1)
    g.close()
