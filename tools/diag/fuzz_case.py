"""Diagnostic: rebuilds ONE case of tools/diag/fuzz.py from the numbers in its description and shows where the RBC structure / the first step differ.
usage: python tools/diag/fuzz_case.py SIDE NR ALPHA FUSED FAST ROT W ZERO HOLES(none|p,f,k) SEED"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz
E_, O = fuzz.E, fuzz.O
Mem = E_.Memory
side, nr, alpha, fused, fast, rot, w, zero, holes, seed = sys.argv[1:11]
side, nr, fused, fast, rot, w, seed = int(side), int(nr), int(fused), int(fast), int(rot), int(w), int(seed)
alpha, zero = float(alpha), float(zero)
m = side * side
F, M = E_.synth_pair(side, seed=seed, zero_fraction=zero)
if holes != "none":
    hp, hf, hk = holes.split(","); hp, hf, hk = int(hp), float(hf), hk == "True"
    F, M = E_.punch_holes(F, side, side, hp, hf, hk, seed=seed), E_.punch_holes(M, side, side, hp, hf, hk, seed=seed)
rng2 = np.random.default_rng(seed)
if int(rng2.integers(0, 3)) == 0:
    F = fuzz.plant_repeats(F, rng2); M = fuzz.plant_repeats(M, rng2) if rng2.integers(0, 2) else M
g = E_.ICP(0, rot, w); g.init(m, nr, alpha, 1e-6)
g.setPowerMode(E_.PowerMode.SQUARED if fast else E_.PowerMode.LITERAL); g.setReduceMode(E_.ReduceMode.FUSED if fused else E_.ReduceMode.REFERENCE_ORDER)
g.write(Mem.F, F); g.write(Mem.M, M)
o = O.OracleICP(m, nr, alpha, 1e-6, rot=rot, weighted=w, power_fast=bool(fast), threads=16, fused=bool(fused))
o.write_f(F); o.write_m(M); g.buildRBC(); o.build_rbc()
print("layout", g.search_layout(), "non-finite fixed points", int((~np.isfinite(F[:, :7])).any(1).sum()), "non-finite reps", int((~np.isfinite(o.reps[:, :7])).any(1).sum()))
ow, oo = g.read(Mem.RBC_OWNER), o.rbc_owner
bad = np.nonzero(ow != oo)[0]
print("owners differ:", bad.size)
for i in bad[:8]:
    print("  point", i, F[i], "engine owner", ow[i], "oracle owner", oo[i], "rep(engine)", o.reps[ow[i]][:7], "rep(oracle)", o.reps[oo[i]][:7],
          "d(engine rep)", O.metric8(F[i], o.reps[ow[i]], alpha), "d(oracle rep)", O.metric8(F[i], o.reps[oo[i]], alpha))
if not bad.size:
    g.step(); o.step()
    gi, oi = g.read(Mem.NN_ID), o.nn_id
    b2 = np.nonzero(gi["id"] != oi["id"])[0]
    print("ids differ:", b2.size, "rid differ:", int((g.read(Mem.RID) != o.rid).sum()))
    for i in b2[:8]:
        print("  query", i, M[i], "engine", gi[i], "rid", g.read(Mem.RID)[i], "oracle", oi[i], "rid", o.rid[i])
