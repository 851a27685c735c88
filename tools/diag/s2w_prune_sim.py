"""Diagnostic (not a test, CPU only, ~15 min): how much exact stage-2 pruning over list chunks could skip in the lanes = candidates form at config C
(profiles/r05_s2w_pruning_sim.txt)."""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
from oracle import oracle as O
side, nr = 1024, 4096
m = side*side
F, M = icp_amd.synth_pair(side)
o = O.OracleICP(m, nr, 2e2, 1e-6, threads=8, power_fast=True, fused=True)
o.write_f(F); o.write_m(M); o.build_rbc()
N = o.rbc_N.astype(np.int64); Ooff = o.rbc_O.astype(np.int64); perm = o.rbc_perm
XP = F[perm]                      # permuted database
a = 200.0
# chunk boxes (geo only + colour) per global chunk of list: chunk c of list r covers positions O[r]+16c..
nch = (N + 15)//16
choff = np.concatenate([[0], np.cumsum(nch)])[:-1]
tot = int(nch.sum())
lo = np.full((tot, 6), np.inf, np.float32); hi = np.full((tot, 6), -np.inf, np.float32)
pos_list = np.repeat(np.arange(nr), N)            # list of each position
pos_in = np.arange(m) - Ooff[pos_list]
cid = choff[pos_list] + pos_in//16
P6 = XP[:, [0,1,2,4,5,6]]
np.minimum.at(lo, cid, P6); np.maximum.at(hi, cid, P6)
prev_win = None
for it in range(1, 11):
    T = o.T.copy()
    o.step()
    rid = o.rid.astype(np.int64)
    nn = o.nn_id
    tM = O.transform_q(M, T)
    q6 = np.concatenate([tM[:, :3], M[:, 4:7]], 1).astype(np.float32)
    # sample queries: waves = rows of 8 within 8x8 tiles; sample 4000 waves
    rng = np.random.default_rng(it)
    wy = rng.integers(0, side, 3000); wx = rng.integers(0, side//8, 3000)*8
    Utot = 0; Ctot = 0; trips_p = 0; trips_f = 0; surv_q = 0; cnt_q = 0
    Utot2 = 0; trips_p2 = 0
    for y, x in zip(wy, wx):
        qi = y*side + x + np.arange(8)
        for r in np.unique(rid[qi]):
            qs = qi[rid[qi] == r]
            c0 = choff[r]; nc = nch[r]
            L, H = lo[c0:c0+nc], hi[c0:c0+nc]
            need = np.zeros(nc, bool); need2 = np.zeros(nc, bool)
            for qq in qs:
                e = np.maximum(np.maximum(L - q6[qq], q6[qq] - H), 0)
                bound = (e[:, :3]**2).sum(1) + a*(e[:, 3:]**2).sum(1)
                # bound (i): distance to the representative
                R = o.reps[r]
                dr = ((q6[qq,:3]-R[:3])**2).sum() + a*((q6[qq,3:]-R[4:7])**2).sum()
                p1 = bound <= dr
                need |= p1
                # bound (ii): previous iteration's winner re-evaluated under the current transform (if any), else dr
                lim = dr
                if prev_win is not None:
                    w = F[prev_win[qq]]
                    lim = min(lim, ((q6[qq,:3]-w[:3])**2).sum() + a*((q6[qq,3:]-w[4:7])**2).sum())
                p2 = bound <= lim
                need2 |= p2
                surv_q += p2.sum(); cnt_q += nc
            Utot += need.sum(); Utot2 += need2.sum(); Ctot += nc
            trips_f += -(-nc//4); trips_p += -(-need.sum()//4); trips_p2 += -(-need2.sum()//4)
    print("iteration %2d: chunks needed per wave-list: bound dr %.2f, bound prev-winner %.2f of %.2f; trips %.2f / %.2f of %.2f; per-query survivors %.2f" % (
        it, Utot/3000, Utot2/3000, Ctot/3000, trips_p/3000, trips_p2/3000, trips_f/3000, surv_q/cnt_q))
    prev_win = nn["id"].astype(np.int64).copy()
