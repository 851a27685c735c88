"""ctypes binding of the CPU oracle (oracle/libicp_oracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never from the product package icp_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libicp_oracle.so")

DIST_ID = np.dtype([("dist", np.float32), ("id", np.uint32)])

ROT_SVD, ROT_POWER = 0, 1
W_REGULAR, W_WEIGHTED = 0, 1


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("icp_oracle.c", "icp_oracle.h", "Makefile"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    up = C.POINTER(C.c_uint32)
    vp = C.c_void_p
    u32 = C.c_uint32

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_get_lms", None, vp, vp)
    sig("orc_reps_grid", C.c_int, u32, u32, up, up, up)
    sig("orc_get_reps", C.c_int, vp, u32, u32, vp, vp)
    sig("orc_transform_q", None, vp, vp, vp, u32)
    sig("orc_transform_q2", None, vp, vp, vp, u32)
    sig("orc_transform_m", None, vp, vp, vp, u32)
    sig("orc_metric8", C.c_float, vp, vp, C.c_float)
    sig("orc_rbc_construct", None, vp, u32, vp, u32, C.c_float, vp, vp, vp, vp, vp)
    sig("orc_rbc_search", None, vp, u32, vp, u32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp)
    sig("orc_nn_brute", None, vp, u32, vp, u32, C.c_float, vp)
    sig("orc_weights", None, vp, u32, vp, C.POINTER(C.c_double))
    sig("orc_mean", None, vp, vp, u32, vp)
    sig("orc_mean_weighted", None, vp, vp, vp, C.c_double, u32, vp)
    sig("orc_devs", None, vp, vp, vp, u32, vp, vp)
    sig("orc_sij", None, vp, vp, vp, u32, C.c_float, vp)
    sig("orc_power_method", C.c_int, vp, vp, vp)
    sig("orc_power_method_fast", C.c_int, vp, vp, vp)
    sig("orc_svd_rotation", None, vp, vp, vp, vp)
    sig("orc_quat_to_rot", None, vp, vp)
    sig("orc_rot_to_quat", None, vp, vp)
    sig("orc_reduce_sum_f", None, vp, u32, u32, vp)
    sig("orc_exscan_u32", None, vp, u32, vp)
    sig("orc_icp_create", vp, C.c_int, C.c_int)
    sig("orc_icp_destroy", None, vp)
    sig("orc_icp_init", C.c_int, vp, u32, u32, C.c_float, C.c_float, u32, C.c_double, C.c_double)
    sig("orc_icp_set_power_fast", None, vp, C.c_int)
    sig("orc_icp_set_fused", None, vp, C.c_int)
    sig("orc_moments_fused", None, vp, vp, vp, u32, u32, C.c_float, C.POINTER(C.c_double), vp, vp)
    sig("orc_icp_set_threads", None, vp, C.c_int)
    sig("orc_icp_set_dist_scale", None, vp, C.c_float)
    sig("orc_icp_set_alpha", None, vp, C.c_float)
    sig("orc_icp_write_f", None, vp, vp)
    sig("orc_icp_write_m", None, vp, vp)
    sig("orc_icp_write_t", None, vp, vp)
    sig("orc_icp_build_rbc", None, vp)
    sig("orc_icp_step", None, vp)
    sig("orc_icp_run", u32, vp)
    sig("orc_icp_converged", C.c_int, vp)
    for n in ("T", "Tk", "R", "Rk", "S", "means", "W", "reps"):
        sig("orc_icp_" + n, fp, vp)
    sig("orc_icp_sum_w", C.c_double, vp)
    sig("orc_icp_nn_id", vp, vp)
    for n in ("rid", "rbc_N", "rbc_O", "rbc_perm", "rbc_owner"):
        sig("orc_icp_" + n, up, vp)
    sig("orc_icp_k", u32, vp)
    sig("orc_icp_last_power_iters", C.c_int, vp)
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- thin functional wrappers --------------------------------------------------------------

def get_lms(cloud):
    cloud = _f32(cloud).reshape(-1)
    assert cloud.size == 640 * 480 * 8
    out = np.empty((16384, 8), np.float32)
    lib().orc_get_lms(_p(cloud), _p(out))
    return out


def reps_grid(m, nr):
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().orc_reps_grid(m, nr, C.byref(a), C.byref(b), C.byref(c))
    return None if rc else (a.value, b.value, c.value)


def get_reps(F, nr):
    F = _f32(F)
    m = F.shape[0]
    R = np.empty((nr, 8), np.float32)
    src = np.empty(nr, np.uint32)
    rc = lib().orc_get_reps(_p(F), m, nr, _p(R), _p(src))
    if rc:
        raise ValueError("unsupported (m, nr)")
    return R, src


def transform_q(M, T, variant=1):
    M = _f32(M)
    T = _f32(T)
    out = np.empty_like(M)
    fn = lib().orc_transform_q if variant == 1 else lib().orc_transform_q2
    fn(_p(M), _p(out), _p(T), M.shape[0])
    return out


def transform_m(M, T16):
    M = _f32(M)
    T16 = _f32(T16).reshape(-1)
    out = np.empty_like(M)
    lib().orc_transform_m(_p(M), _p(out), _p(T16), M.shape[0])
    return out


def metric8(x, y, a):
    x = _f32(x)
    y = _f32(y)
    return float(lib().orc_metric8(_p(x), _p(y), a))


def rbc_construct(F, R, a):
    F = _f32(F)
    R = _f32(R)
    m, nr = F.shape[0], R.shape[0]
    owner = np.empty(m, np.uint32)
    N = np.empty(nr, np.uint32)
    O = np.empty(nr, np.uint32)
    perm = np.empty(m, np.uint32)
    XP = np.empty((m, 8), np.float32)
    lib().orc_rbc_construct(_p(F), m, _p(R), nr, a, _p(owner), _p(N), _p(O), _p(perm), _p(XP))
    return dict(owner=owner, N=N, O=O, perm=perm, XP=XP)


def rbc_search(Q, R, rbc, rep_src, a):
    Q = _f32(Q)
    R = _f32(R)
    nq, nr = Q.shape[0], R.shape[0]
    nn_id = np.empty(nq, DIST_ID)
    NN = np.empty((nq, 8), np.float32)
    rid = np.empty(nq, np.uint32)
    rs = np.ascontiguousarray(rep_src, np.uint32)
    lib().orc_rbc_search(_p(Q), nq, _p(R), nr, _p(rbc["XP"]), _p(rbc["perm"]), _p(rbc["O"]),
                         _p(rbc["N"]), _p(rs), a, _p(nn_id), _p(NN), _p(rid))
    return nn_id, NN, rid


def nn_brute(Q, F, a):
    Q = _f32(Q)
    F = _f32(F)
    out = np.empty(Q.shape[0], DIST_ID)
    lib().orc_nn_brute(_p(Q), Q.shape[0], _p(F), F.shape[0], a, _p(out))
    return out


def weights(nn_id):
    nn_id = np.ascontiguousarray(nn_id, DIST_ID)
    n = nn_id.shape[0]
    Wt = np.zeros(n, np.float32)
    sw = C.c_double()
    lib().orc_weights(_p(nn_id), n, _p(Wt), C.byref(sw))
    return Wt, sw.value


def mean(F, M):
    F = _f32(F)
    M = _f32(M)
    out = np.empty(8, np.float32)
    lib().orc_mean(_p(F), _p(M), F.shape[0], _p(out))
    return out


def mean_weighted(F, M, Wt, sum_w):
    F = _f32(F)
    M = _f32(M)
    Wt = _f32(Wt)
    out = np.empty(8, np.float32)
    lib().orc_mean_weighted(_p(F), _p(M), _p(Wt), sum_w, F.shape[0], _p(out))
    return out


def devs(F, M, mean8):
    F = _f32(F)
    M = _f32(M)
    mean8 = _f32(mean8)
    n = F.shape[0]
    DF = np.empty((n, 4), np.float32)
    DM = np.empty((n, 4), np.float32)
    lib().orc_devs(_p(F), _p(M), _p(mean8), n, _p(DF), _p(DM))
    return DF, DM


def sij(DM, DF, Wt, c):
    DM = _f32(DM)
    DF = _f32(DF)
    out = np.empty(11, np.float32)
    wp = None if Wt is None else _p(_f32(Wt))
    if Wt is not None:
        Wt = _f32(Wt)
        wp = _p(Wt)
    lib().orc_sij(_p(DM), _p(DF), wp, DM.shape[0], c, _p(out))
    return out


def moments_fused(NN, tM, Wt, side, c):
    NN = _f32(NN)
    tM = _f32(tM)
    m = NN.shape[0]
    means = np.empty(8, np.float32)
    S = np.empty(11, np.float32)
    sw = C.c_double()
    wp = None
    if Wt is not None:
        Wt = _f32(Wt)
        wp = _p(Wt)
    lib().orc_moments_fused(_p(NN), _p(tM), wp, m, side, c, C.byref(sw), _p(means), _p(S))
    return sw.value, means, S


def power_method(S, means, fast=False):
    S = _f32(S)
    means = _f32(means)
    Tk = np.empty(8, np.float32)
    fn = lib().orc_power_method_fast if fast else lib().orc_power_method
    it = fn(_p(S), _p(means), _p(Tk))
    return Tk, it


def svd_rotation(S, means):
    S = _f32(S)
    means = _f32(means)
    Rk = np.empty(9, np.float32)
    Tk = np.empty(8, np.float32)
    lib().orc_svd_rotation(_p(S), _p(means), _p(Rk), _p(Tk))
    return Rk.reshape(3, 3), Tk


def quat_to_rot(q):
    q = _f32(q)
    R = np.empty(9, np.float32)
    lib().orc_quat_to_rot(_p(q), _p(R))
    return R.reshape(3, 3)


def rot_to_quat(R):
    R = _f32(R).reshape(-1)
    q = np.empty(4, np.float32)
    lib().orc_rot_to_quat(_p(R), _p(q))
    return q


def reduce_sum_f(a):
    a = _f32(a)
    rows, cols = a.shape
    out = np.empty(rows, np.float32)
    lib().orc_reduce_sum_f(_p(a), cols, rows, _p(out))
    return out


class OracleICP:
    """Mirror of cl_algo::ICP::ICP<CR,CW> on the CPU oracle."""

    def __init__(self, m, nr, a=2e2, c=1e-6, rot=ROT_POWER, weighted=W_WEIGHTED,
                 max_iterations=40, angle_threshold=0.001, translation_threshold=0.01,
                 power_fast=False, threads=1, fused=False, dist_scale=1.0):
        self.L = lib()
        self.h = self.L.orc_icp_create(rot, weighted)
        self.m, self.nr = m, nr
        rc = self.L.orc_icp_init(self.h, m, nr, a, c, max_iterations, angle_threshold, translation_threshold)
        if rc:
            self.L.orc_icp_destroy(self.h)
            self.h = None
            raise ValueError("orc_icp_init rejected the arguments")
        self.L.orc_icp_set_power_fast(self.h, int(power_fast))
        self.L.orc_icp_set_threads(self.h, threads)
        self.L.orc_icp_set_fused(self.h, int(fused))
        self.L.orc_icp_set_dist_scale(self.h, dist_scale)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_icp_destroy(self.h)
            self.h = None

    def write_f(self, F):
        F = _f32(F)
        assert F.shape == (self.m, 8)
        self.L.orc_icp_write_f(self.h, _p(F))

    def write_m(self, M):
        M = _f32(M)
        assert M.shape == (self.m, 8)
        self.L.orc_icp_write_m(self.h, _p(M))

    def write_t(self, T):
        T = _f32(T)
        self.L.orc_icp_write_t(self.h, _p(T))

    def build_rbc(self):
        self.L.orc_icp_build_rbc(self.h)

    def step(self):
        self.L.orc_icp_step(self.h)

    def run(self):
        return self.L.orc_icp_run(self.h)

    def _arr(self, name, n, dtype=np.float32):
        ptr = getattr(self.L, "orc_icp_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    T = property(lambda s: s._arr("T", 8))
    Tk = property(lambda s: s._arr("Tk", 8))
    R = property(lambda s: s._arr("R", 9).reshape(3, 3))
    Rk = property(lambda s: s._arr("Rk", 9).reshape(3, 3))
    S = property(lambda s: s._arr("S", 11))
    means = property(lambda s: s._arr("means", 8))
    W = property(lambda s: s._arr("W", s.m))
    reps = property(lambda s: s._arr("reps", s.nr * 8).reshape(s.nr, 8))
    rid = property(lambda s: s._arr("rid", s.m, np.uint32))
    rbc_N = property(lambda s: s._arr("rbc_N", s.nr, np.uint32))
    rbc_O = property(lambda s: s._arr("rbc_O", s.nr, np.uint32))
    rbc_perm = property(lambda s: s._arr("rbc_perm", s.m, np.uint32))
    rbc_owner = property(lambda s: s._arr("rbc_owner", s.m, np.uint32))
    sum_w = property(lambda s: s.L.orc_icp_sum_w(s.h))
    k = property(lambda s: s.L.orc_icp_k(s.h))
    converged = property(lambda s: bool(s.L.orc_icp_converged(s.h)))
    power_iters = property(lambda s: s.L.orc_icp_last_power_iters(s.h))

    @property
    def nn_id(self):
        ptr = self.L.orc_icp_nn_id(self.h)
        buf = (C.c_char * (self.m * 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=DIST_ID).copy()
