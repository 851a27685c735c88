"""Diagnostic (not a test, CPU only): the list of the representatives at the origin (a frame's invalid points) — how many of its chunks of 8
entries pass the colour-box test of a query with a tight bound, in index order (what k_reps_and_boxes writes) and sorted by a Morton key of
the colour.  Measured: 10 of 49 / 157 chunks at (2^20, 4096) with 10 % scattered / 30 % contiguous invalid points, 4 when sorted — built afterwards
(k_reps_and_boxes: origin_list_close; profiles/r05_holes_colour_order_ab.txt)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
from icp_amd import workloads as W
def morton(q):
    k = np.zeros(len(q), np.uint32)
    for bit in range(5):
        for c in range(3):
            k |= ((q[:, c] >> bit) & 1).astype(np.uint32) << (3 * bit + c)
    return k
for name, side, nr in (("scattered10", 1024, 4096), ("blobs30", 1024, 4096), ("blobs30", 256, 1024), ("scattered10", 256, 1024)):
    F, M = W.holes_pair(icp_amd, name, side)
    F = F.reshape(-1, 8); M = M.reshape(-1, 8)
    nrx = 1 << ((nr.bit_length() - 1) - (nr.bit_length() - 1) // 2); nry = nr // nrx
    sx, sy = side // nrx, side // nry
    gy, gx = np.divmod(np.arange(nr), nrx)
    src = (gy * sy + (sy >> 1) - 1) * side + gx * sx + (sx >> 1) - 1
    R = F[src]
    org = np.where((R[:, 0] == 0) & (R[:, 1] == 0) & (R[:, 2] == 0))[0]
    col = R[org][:, 4:7].astype(np.float64)
    hq = M[(M[:, 0] == 0) & (M[:, 1] == 0) & (M[:, 2] == 0)][:, 4:7].astype(np.float64)
    hq = hq[np.random.default_rng(1).choice(len(hq), min(2000, len(hq)), replace=False)]
    def chunks_passing(c):
        n = len(c); nch = (n + 7) // 8
        lo = np.array([c[8*i:8*i+8].min(0) for i in range(nch)]); hi = np.array([c[8*i:8*i+8].max(0) for i in range(nch)])
        tot = 0
        for q in hq:
            best = ((c - q) ** 2).sum(1).min()
            e = np.maximum(np.maximum(lo - q, q - hi), 0)
            tot += np.count_nonzero((e ** 2).sum(1) <= best)
        return tot / len(hq), nch
    a, nch = chunks_passing(col)
    lo, hi = col.min(0), col.max(0)
    q = np.clip(((col - lo) / np.maximum(hi - lo, 1e-30) * 31.999).astype(np.int64), 0, 31)
    order = np.argsort(morton(q), kind="stable")
    b, _ = chunks_passing(col[order])
    print("%-12s side %4d nr %4d: origin reps %4d, chunks %3d; chunks passing per hole query (tight bound): index order %.1f, colour-sorted %.1f" % (name, side, nr, len(org), nch, a, b))
