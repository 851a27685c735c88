#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle and the synthetic generator.

The reference itself cannot be built or run in this image (CLUtils / RandomBallCover / Eigen /
OpenCL device absent; its CPU twins need <RBC/data_types.hpp>), so these vectors are the
oracle's own outputs on seeded inputs: they pin the oracle (and the engine) against regressions.
The reference-held literals live in reference_kat.json.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
import icp_amd  # noqa: E402  (host-only generator; no GPU needed)


def main():
    out = {}
    side, nr = 32, 16
    F, M = icp_amd.synth_pair(side)
    out["F_head"] = F[:64]
    out["M_head"] = M[:64]
    out["F_sum"] = F.astype(np.float64).sum(0)
    out["M_sum"] = M.astype(np.float64).sum(0)
    o = O.OracleICP(side * side, nr, 2e2, 1e-6)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    out["rbc_N"], out["rbc_O"] = o.rbc_N, o.rbc_O
    out["rbc_perm"], out["rbc_owner"] = o.rbc_perm, o.rbc_owner
    Ts, Tks, Ss, ms, sws, ids, dists = [], [], [], [], [], [], []
    for _ in range(5):
        o.step()
        Ts.append(o.T); Tks.append(o.Tk); Ss.append(o.S); ms.append(o.means); sws.append(o.sum_w)
        nn = o.nn_id
        ids.append(nn["id"][:64].copy()); dists.append(nn["dist"][:64].copy())
    out.update(T=np.array(Ts), Tk=np.array(Tks), S=np.array(Ss), means=np.array(ms), sum_w=np.array(sws),
               nn_id_head=np.array(ids), nn_dist_head=np.array(dists))
    k = o.run()
    out["run_k"] = np.array([o.k])
    out["run_T"] = o.T
    # config 1 plumbing: SVD rotation path on the CPU oracle, kg-like pair at 16384 / 256 is covered by
    # tests; here a small one
    s = O.OracleICP(side * side, nr, 2e2, 1e-6, rot=O.ROT_SVD)
    s.write_f(F); s.write_m(M); s.build_rbc()
    out["svd_k"] = np.array([s.run()])
    out["svd_T"] = s.T
    # the bench's modes: single-pass double moments (fused) + squared-start power method, at 32 x 32 / 16 and at the
    # benchmark size 128 x 128 / 256
    for tag, sd, r in (("fs32", 32, 16), ("fs128", 128, 256)):
        Ff, Mf = icp_amd.synth_pair(sd)
        f = O.OracleICP(sd * sd, r, 2e2, 1e-6, power_fast=True, fused=True, threads=8)
        f.write_f(Ff); f.write_m(Mf); f.build_rbc()
        Ts, Ss, ms, its, ids = [], [], [], [], []
        for _ in range(4):
            f.step()
            Ts.append(f.T); Ss.append(f.S); ms.append(f.means); its.append(f.power_iters)
            ids.append(f.nn_id["id"][:64].copy())
        out.update({tag + "_T": np.array(Ts), tag + "_S": np.array(Ss), tag + "_means": np.array(ms),
                    tag + "_pm_iters": np.array(its), tag + "_ids": np.array(ids)})
        out[tag + "_run_k"] = np.array([f.run()])
        out[tag + "_run_T"] = f.T
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"), {k: np.asarray(v).shape for k, v in out.items()})
    config4()
    bench_configs()
    round5()
    round6()


def round5():
    """Round 5's workloads (icp_amd/workloads.py), default modes, oracle (power_fast=True, fused=True):
    holes — the benchmark pair with a Kinect frame's invalid points in both frames (six cases): N.max, run () (k, converged, T, digest of all
            16384 ids), and the 40-iteration fixed pass from the identity (T, ids digest);
    wall  — the textured plane moved in its own plane (the reference's kg_pc8d_wall stand-in): run () with max_iterations = 300 at a = 2e2
            and at a = 1e-6 (k, converged, T, ids digest), the ground truth beside them."""
    from icp_amd import workloads as W
    out = {}
    for name in W.HOLES:
        F, M = W.holes_pair(icp_amd, name)
        o = O.OracleICP(W.M_POINTS, W.NR, W.A, W.C_, power_fast=True, fused=True, threads=8)
        o.write_f(F); o.write_m(M); o.build_rbc()
        out[name + "_N_max"] = np.array([int(o.rbc_N.max())])
        k = o.run()
        out[name + "_run"] = np.array([k, int(o.converged)])
        out[name + "_run_T"], out[name + "_run_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
        o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
        for _ in range(40):
            o.step()
        out[name + "_fixed40_T"], out[name + "_fixed40_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
    F, M, Tt = W.wall_pair(icp_amd)
    out["wall_T_true"] = Tt
    for tag, a in (("wall_a2e2", W.A), ("wall_asmall", W.WALL_A_SMALL)):
        o = O.OracleICP(W.M_POINTS, W.NR, a, W.C_, power_fast=True, fused=True, threads=8, max_iterations=W.WALL_MAX_ITERATIONS)
        o.write_f(F); o.write_m(M); o.build_rbc()
        k = o.run()
        out[tag + "_run"] = np.array([k, int(o.converged)])
        out[tag + "_run_T"], out[tag + "_run_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
    np.savez_compressed(os.path.join(HERE, "round5_vectors.npz"), **out)
    print("wrote round5_vectors.npz:", {k: (v.tolist() if v.size <= 2 else v.shape) for k, v in out.items() if k.endswith("_run") or k.endswith("N_max")})


ROUND6_C_CASES = ("scattered10", "blobs30", "blobs30_rgb0")


def round6():
    """Round 6: invalid points at config C's size (|F|=|M|=2^20, |R|=4096) — the MASKED + lanes-as-candidates
    search with the origin list read per wave behind colour boxes, the Morton colour sort of `k_reps_and_boxes`, the compaction by the
    last-arriving block — in the default modes, from the oracle (power_fast=True, fused=True):
      C_<case>_{N,O} + digests of perm / owner; two free-running steps (T, S, means, sum_w; digests of ALL 2^20 ids, distances and
      nearest representatives); `sub` = the 4096 query indices the GPU test re-checks against the live oracle.
    B `blobs30_rgb0` through run () (k, converged, T, ids digest)."""
    from icp_amd import workloads as W
    out = {}
    side, nr = W.CONFIGS["C"]
    m = side * side
    out["sub"] = np.sort(np.random.default_rng(60606).choice(m, 4096, replace=False)).astype(np.uint32)
    for name in ROUND6_C_CASES:
        F, M = W.holes_pair(icp_amd, name, side)
        o = O.OracleICP(m, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
        o.write_f(F); o.write_m(M); o.build_rbc()
        tag = "C_" + name
        out[tag + "_N"], out[tag + "_O"] = o.rbc_N, o.rbc_O
        out[tag + "_perm_digest"], out[tag + "_owner_digest"] = W.ids_digest(o.rbc_perm), W.ids_digest(o.rbc_owner)
        out[tag + "_invalid"] = np.array([int(np.count_nonzero((X[:, 0] == 0) & (X[:, 1] == 0) & (X[:, 2] == 0))) for X in (F, M)])
        Ts, Ss, ms, sws, idd, ridd, dd = [], [], [], [], [], [], []
        for _ in range(2):
            o.step()
            nn = o.nn_id
            Ts.append(o.T); Ss.append(o.S); ms.append(o.means); sws.append(o.sum_w)
            idd.append(W.ids_digest(nn["id"])); ridd.append(W.ids_digest(o.rid)); dd.append(W.bits_digest(nn["dist"]))
        out.update({tag + "_T": np.array(Ts), tag + "_S": np.array(Ss), tag + "_means": np.array(ms), tag + "_sum_w": np.array(sws),
                    tag + "_ids_digest": np.array(idd), tag + "_rid_digest": np.array(ridd), tag + "_dist_digest": np.array(dd)})
        print("round6", tag, "N_max", int(o.rbc_N.max()), "invalid", out[tag + "_invalid"].tolist(), flush=True)
    side, nr = W.CONFIGS["B"]
    F, M = W.holes_pair(icp_amd, "blobs30_rgb0", side)
    o = O.OracleICP(side * side, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    k = o.run()
    out["B_blobs30_rgb0_run"] = np.array([k, int(o.converged)])
    out["B_blobs30_rgb0_run_T"], out["B_blobs30_rgb0_run_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
    out["B_blobs30_rgb0_N_max"] = np.array([int(o.rbc_N.max())])
    np.savez_compressed(os.path.join(HERE, "round6_vectors.npz"), **out)
    print("wrote round6_vectors.npz; B blobs30_rgb0 run k = %d converged = %d N_max = %d" % (k, int(o.converged), int(o.rbc_N.max())))


def config4():
    """BASELINE config 4 at its real per-GPU shape (64 pairs of 16384 / 256): for the checked registrations of
    icp_amd/workloads.py, the default modes' (fused + squared) run to convergence and the 40-iteration fixed run from the
    identity: k, converged, T, the first 256 correspondence ids and a digest of all of them."""
    from icp_amd import workloads as C4
    out = {"checked": np.array(C4.CHECKED)}
    for i in C4.CHECKED:
        F, M = C4.pair(icp_amd, i)
        o = O.OracleICP(C4.M_POINTS, C4.NR, C4.A, C4.C_, power_fast=True, fused=True, threads=8)
        o.write_f(F); o.write_m(M); o.build_rbc()
        k = o.run()
        ids = o.nn_id["id"]
        out["r%d_run" % i] = np.array([k, int(o.converged)])
        out["r%d_run_T" % i] = o.T
        out["r%d_run_ids_head" % i] = ids[:256].copy()
        out["r%d_run_ids_digest" % i] = C4.ids_digest(ids)
        o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
        for _ in range(40):
            o.step()
        ids = o.nn_id["id"]
        out["r%d_fixed40_T" % i] = o.T
        out["r%d_fixed40_ids_head" % i] = ids[:256].copy()
        out["r%d_fixed40_ids_digest" % i] = C4.ids_digest(ids)
    np.savez_compressed(os.path.join(HERE, "config4_vectors.npz"), **out)
    print("wrote config4_vectors.npz; k of the checked registrations:", [int(out["r%d_run" % i][0]) for i in C4.CHECKED])


def bench_configs():
    """BASELINE configs B (256^2, 1024) and C (1024^2, 4096) in exactly the modes and at exactly the sizes bench.py times
    (the handle's defaults: fused double moments + squared power start), from the oracle (power_fast=True, fused=True).
    C: the RBC structure, then two free-running steps (T, S, means, sum_w, digests of all 2^20 ids / distances / nearest
       representatives), then the bench's own pass: 10 fixed iterations from the identity (T, ids digest).
    B: run () to convergence (k, converged, T, ids digest) and the bench's pass of 40 fixed iterations from the identity.
    Digests (icp_amd/workloads.py): ids = (crc32, sum), everything else = (crc32 of the raw bytes, byte count)."""
    from icp_amd import workloads as W
    out = {}
    side, nr = W.CONFIGS["C"]
    F, M = icp_amd.synth_pair(side)
    o = O.OracleICP(side * side, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    out["C_N"], out["C_O"] = o.rbc_N, o.rbc_O
    out["C_perm_digest"], out["C_owner_digest"] = W.ids_digest(o.rbc_perm), W.ids_digest(o.rbc_owner)
    Ts, Ss, ms, sws, idd, ridd, dd = [], [], [], [], [], [], []
    for _ in range(2):
        o.step()
        nn = o.nn_id
        Ts.append(o.T); Ss.append(o.S); ms.append(o.means); sws.append(o.sum_w)
        idd.append(W.ids_digest(nn["id"])); ridd.append(W.ids_digest(o.rid)); dd.append(W.bits_digest(nn["dist"]))
    out.update(C_T=np.array(Ts), C_S=np.array(Ss), C_means=np.array(ms), C_sum_w=np.array(sws), C_ids_digest=np.array(idd),
               C_rid_digest=np.array(ridd), C_dist_digest=np.array(dd), C_ids_head=o.nn_id["id"][:256].copy())
    o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
    for _ in range(10):
        o.step()
    out["C_fixed10_T"], out["C_fixed10_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
    side, nr = W.CONFIGS["B"]
    F, M = icp_amd.synth_pair(side)
    o = O.OracleICP(side * side, nr, W.A, W.C_, power_fast=True, fused=True, threads=8)
    o.write_f(F); o.write_m(M); o.build_rbc()
    k = o.run()
    out["B_run"] = np.array([k, int(o.converged)])
    out["B_run_T"], out["B_run_ids_digest"], out["B_run_ids_head"] = o.T, W.ids_digest(o.nn_id["id"]), o.nn_id["id"][:256].copy()
    o.write_t([0, 0, 0, 1, 0, 0, 0, 1])
    for _ in range(40):
        o.step()
    out["B_fixed40_T"], out["B_fixed40_ids_digest"] = o.T, W.ids_digest(o.nn_id["id"])
    np.savez_compressed(os.path.join(HERE, "bench_config_vectors.npz"), **out)
    print("wrote bench_config_vectors.npz; B run k = %d converged = %d" % (k, int(o.converged)))


if __name__ == "__main__":
    if len(sys.argv) > 1:                            # e.g. `make_golden.py round6`: one fixture file only
        for name in sys.argv[1:]:
            {"round5": round5, "round6": round6, "config4": config4, "bench_configs": bench_configs}[name]()
    else:
        main()
