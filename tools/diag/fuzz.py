"""Diagnostic (not a test): randomized parity sweep of the HIP path against the CPU oracle — shapes, representative counts, alpha,
reduce / power modes, rotation solver, weighting, holes, synthetic-scene seeds, batch position; RBC + a few free-running steps or a
checked run per case, everything compared bit for bit.  usage: python tools/diag/fuzz.py [SECONDS] [SEED]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd as E
from oracle import oracle as O

Mem = E.Memory
SIDES = [6, 16, 24, 30, 32, 40, 48, 64, 64, 96, 128, 128, 192, 256]


def valid_nr(side):
    out = []
    for lg in range(1, 14):
        nr = 1 << lg
        x, y = 1 << (lg - lg // 2), 1 << (lg // 2)
        if nr <= side * side and side % x == 0 and side % y == 0 and nr <= 8192:
            out.append(nr)
    return out


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def same(a, b):
    """bit for bit; NaN against NaN counts as equal whatever the payload (garbage in: the same garbage kind out)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(bits(a).reshape(-1)[~na.reshape(-1)], bits(b).reshape(-1)[~nb.reshape(-1)])


def compare(g, o, b, weighted, what):
    bad = []
    rd = lambda mem: g.read(mem, b) if b is not None else g.read(mem)
    if not np.array_equal(rd(Mem.RID), o.rid): bad.append("rid")
    gn, on = rd(Mem.NN_ID), o.nn_id
    if not np.array_equal(gn["id"], on["id"]): bad.append("ids(%d)" % np.count_nonzero(gn["id"] != on["id"]))
    if not same(gn["dist"], on["dist"]): bad.append("dist")
    if weighted and not same(rd(Mem.W), o.W): bad.append("W")
    for mem, ref, nm in [(Mem.MEANS, o.means, "means"), (Mem.S, o.S, "S"), (Mem.TK, o.Tk, "Tk"), (Mem.T, o.T, "T")]:
        if not same(rd(mem), ref): bad.append(nm)
    return bad


def plant_repeats(X, rng):
    """Bit-identical points planted in a point set (round 6: the search's view of a long list drops the members that repeat an earlier one):
    contiguous runs of one point, a share of the set drawn from a few points, the origin with +0 / -0 coordinates and colours."""
    X = X.copy(); m = X.shape[0]
    kind = int(rng.integers(0, 4))
    if kind == 0:
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, m)); n = int(rng.integers(2, max(3, m // 3)))
            X[a:a + n] = X[int(rng.integers(0, m))]
    elif kind == 1:
        src = X[rng.choice(m, int(rng.integers(1, 6)), replace=True)].copy()
        idx = rng.choice(m, max(1, int(m * float(rng.choice([0.05, 0.3, 0.6])))), replace=False)
        X[idx] = src[rng.integers(0, src.shape[0], idx.size)]
    elif kind == 3:                                # non-finite coordinates (round 6: such a distance never wins, first candidates included)
        for bad in (np.nan, np.inf):
            p = X[int(rng.integers(0, m))].copy(); p[int(rng.integers(0, 7))] = bad
            X[rng.choice(m, max(1, m // int(rng.choice([4, 50, 500]))), replace=False)] = p
    else:
        idx = rng.choice(m, max(1, m // 3), replace=False)
        z = np.where(rng.integers(0, 2, (idx.size, 6)) == 1, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
        X[idx, 0:3] = z[:, 0:3]
        if rng.integers(0, 2): X[idx, 4:7] = z[:, 3:6]
    return X


def run(seconds=None, cases=None, seed=1, verbose=True):
    """Random cases until `seconds` have passed or `cases` have run; returns (cases, list of failure descriptions)."""
    rng = np.random.default_rng(seed)
    t0 = time.time(); ncase = 0; fails = []
    while (seconds is None or time.time() - t0 < seconds) and (cases is None or ncase < cases):
        side = int(rng.choice(SIDES)); m = side * side
        nr = int(rng.choice(valid_nr(side)))
        if m // nr > 4096: continue                       # (lists far beyond anything the configs use: minutes on the oracle)
        alpha = float(rng.choice([0.5, 2e2, 2e2, 1e4]))
        fused = bool(rng.integers(0, 2)); fast = bool(rng.integers(0, 2)) if fused else bool(rng.integers(0, 4) == 0)
        rot = int(rng.integers(0, 5) != 0); weighted = int(rng.integers(0, 4) != 0)
        zero = float(rng.choice([0.0, 0.0, 0.1])); seed = int(rng.integers(1, 1 << 30))
        batch = int(rng.choice([1, 1, 3])) if m <= 16384 else 1
        bsel = int(rng.integers(0, batch)) if batch > 1 else None
        runmode = bool(rng.integers(0, 3) == 0)
        holes = (int(rng.integers(0, 2)), float(rng.choice([0.1, 0.3])), bool(rng.integers(0, 2))) if rng.integers(0, 3) == 0 else None
        rng2 = np.random.default_rng(seed)                 # (draws of its own: the cases of earlier rounds' seeds stay what they were)
        repeats = int(rng2.integers(0, 3)) == 0
        desc = "side %d nr %d a %g fused %d fast %d rot %d w %d zero %.1f holes %s repeats %d seed %d batch %d/%s %s" % (side, nr, alpha, fused, fast, rot, weighted, zero, holes, repeats, seed, batch, bsel, "run" if runmode else "steps")
        try:
            g = E.ICP(0, rot, weighted)
            g.init(m, nr, alpha, 1e-6, batch=batch) if batch > 1 else g.init(m, nr, alpha, 1e-6)
            g.setPowerMode(E.PowerMode.SQUARED if fast else E.PowerMode.LITERAL)
            g.setReduceMode(E.ReduceMode.FUSED if fused else E.ReduceMode.REFERENCE_ORDER)
            pairs = [E.synth_pair(side, seed=seed + 7 * b, zero_fraction=zero) for b in range(batch)]
            if holes:                                               # a Kinect frame's invalid points (round 5): scattered / contiguous, colour kept / zeroed
                hp, hf, hk = holes
                pairs = [(E.punch_holes(F, side, side, hp, hf, hk, seed=seed + 11 * b), E.punch_holes(M, side, side, hp, hf, hk, seed=seed + 13 * b)) for b, (F, M) in enumerate(pairs)]
            if repeats:
                pairs = [(plant_repeats(F, rng2), plant_repeats(M, rng2) if rng2.integers(0, 2) else M) for F, M in pairs]
            for b, (F, M) in enumerate(pairs):
                if batch > 1: g.write(Mem.F, F, batch_index=b); g.write(Mem.M, M, batch_index=b)
                else: g.write(Mem.F, F); g.write(Mem.M, M)
            F, M = pairs[bsel or 0]
            o = O.OracleICP(m, nr, alpha, 1e-6, rot=rot, weighted=weighted, power_fast=fast, threads=16, fused=fused)
            o.write_f(F); o.write_m(M)
            g.buildRBC(); o.build_rbc()
            rd = (lambda mem: g.read(mem, bsel)) if bsel is not None else g.read
            bad = []
            if not np.array_equal(rd(Mem.RBC_OWNER), o.rbc_owner): bad.append("owner")
            if not np.array_equal(rd(Mem.RBC_PERM), o.rbc_perm): bad.append("perm")
            if not np.array_equal(rd(Mem.RBC_N), o.rbc_N): bad.append("N")
            if runmode and batch == 1:
                k = g.run(); ko = o.run()
                if k != ko: bad.append("k %d/%d" % (k, ko))
                bad += compare(g, o, bsel, weighted, "run")
            else:
                for it in range(3):
                    g.step(); o.step()
                    b_ = compare(g, o, bsel, weighted, "step %d" % it)
                    if b_: bad += ["step%d:" % it] + b_; break
            g.close()
        except Exception as ex:                             # noqa: BLE001 (diagnostic: report and go on)
            bad = ["EXC " + repr(ex)[:200]]
        ncase += 1
        if bad:
            fails.append(desc + " -> " + " ".join(bad))
            print("FAIL", fails[-1], flush=True)
        elif verbose and ncase % 10 == 0:
            print("ok   %4d cases, %.0f s (last: %s)" % (ncase, time.time() - t0, desc), flush=True)
    return ncase, fails


if __name__ == "__main__":
    t0 = time.time()
    n, fails = run(seconds=float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("cases %d failures %d in %.0f s" % (n, len(fails), time.time() - t0))
    sys.exit(1 if fails else 0)
