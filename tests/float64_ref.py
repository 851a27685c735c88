"""float64 solution of the registration an ICP run converges to — TEST / MEASUREMENT INFRASTRUCTURE, not part of the engine and not an
oracle of its bits: it answers "which of the two fp32 formulations of the iteration is closer to the exact arithmetic?".

Given the correspondences an fp32 run found in each of its iterations, the SAME iteration is restated in numpy float64 —
transform (s R p + t), weights 100 / (100 + d) on the metric geo + a pho (kernels/icp_kernels.cl:139-180; the metric text of
src/ICP/algorithms.cpp:4393-4398), weighted centroids (kernels/icp_kernels.cl:455-495), deviations and the scaled cross-covariance
S_ab = sum w (c dm_a)(c df_b), S9 / S10 (:703-743), Horn's 4 x 4 matrix (:993-999) and its dominant eigenvector by a symmetric
eigendecomposition (what the power method of :1012-1041 iterates towards), s_k = sqrt (S9 / S10), t_k = m_f - s_k R_k m_m (:989, 1050),
and the composition R <- R_k R, t <- s_k R_k t + t_k, s <- s_k s (src/ICP/algorithms.cpp:4683-4695) — and carried through all
iterations.  ICP contracts towards the fixed point its final correspondences define, so the result does not depend on the fp32 run's
rounding beyond which correspondences it found (the two modes of the engine agree on 99.99 % of them).
"""
import numpy as np


def quat_to_rot(q):
    x, y, z, w = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def rot_to_quat(R):
    """Unit quaternion (x, y, z, w) of a rotation matrix, w >= 0."""
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], R[1, 0] + R[0, 1], R[2, 0] + R[0, 2], R[2, 1] - R[1, 2]],
                  [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], R[2, 1] + R[1, 2], R[0, 2] - R[2, 0]],
                  [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], R[1, 0] - R[0, 1]],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]], np.float64) / 3.0
    w, v = np.linalg.eigh(K)
    q = v[:, np.argmax(w)]
    return q if q[3] >= 0 else -q


def horn_matrix(S):
    """kernels/icp_kernels.cl:993-999 (S_ab: a = moving, b = fixed)."""
    (Sxx, Sxy, Sxz), (Syx, Syy, Syz), (Szx, Szy, Szz) = S
    return np.array([[Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz, Syz - Szy],
                     [Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy, Szx - Sxz],
                     [Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz, Sxy - Syx],
                     [Syz - Szy, Szx - Sxz, Sxy - Syx, Sxx + Syy + Szz]], np.float64)


class Float64ICP:
    """R, t, s in float64; step (ids) = one iteration with the given correspondences (ids[i] = index into F of query i)."""

    def __init__(self, F, M, a, c, weighted=True, dist_scale=1.0):
        self.F, self.M = np.asarray(F, np.float64), np.asarray(M, np.float64)
        self.a, self.c, self.weighted, self.fg = float(a), float(c), bool(weighted), float(dist_scale)
        self.R, self.t, self.s = np.eye(3), np.zeros(3), 1.0

    def step(self, ids):
        ids = np.asarray(ids, np.int64)
        tM = self.s * (self.M[:, :3] @ self.R.T) + self.t
        NN = self.F[ids, :3]
        if self.weighted:
            geo = ((tM - NN) ** 2).sum(1)
            pho = ((self.M[:, 4:7] - self.F[ids, 4:7]) ** 2).sum(1)
            w = 100.0 / (100.0 + self.fg * (geo + self.a * pho))
        else:
            w = np.ones(len(ids))
        sw = w.sum()
        mf, mm = (w[:, None] * NN).sum(0) / sw, (w[:, None] * tM).sum(0) / sw
        df, dm = self.c * (NN - mf), self.c * (tM - mm)
        S = np.einsum("i,ia,ib->ab", w, dm, df)
        sk = np.sqrt((w * (df ** 2).sum(1)).sum() / (w * (dm ** 2).sum(1)).sum())
        ev, vec = np.linalg.eigh(horn_matrix(S))
        Rk = quat_to_rot(vec[:, np.argmax(ev)])
        tk = mf - sk * (Rk @ mm)
        self.R, self.t, self.s = Rk @ self.R, sk * (Rk @ self.t) + tk, sk * self.s
        return Rk, tk, sk

    @property
    def T(self):
        """[q | t, s] like the engine's T."""
        return np.concatenate([rot_to_quat(self.R), self.t, [self.s]])


def errors_against(T32, T64, scene_scale):
    """How far an fp32 result [q | t, s] is from the float64 one: |dq| (sign-aligned), |dt| in mm, |dt| / |t|, |dt| / scene, |ds| / s."""
    T32, T64 = np.asarray(T32, np.float64), np.asarray(T64, np.float64)
    q32 = T32[:4] if np.dot(T32[:4], T64[:4]) >= 0 else -T32[:4]
    dt = np.linalg.norm(T32[4:7] - T64[4:7])
    return {"dq": float(np.linalg.norm(q32 - T64[:4])), "dt_mm": float(dt), "dt_over_t": float(dt / np.linalg.norm(T64[4:7])),
            "dt_over_scene": float(dt / scene_scale), "ds_over_s": float(abs(T32[7] - T64[7]) / T64[7])}
