"""icp_amd — MI355X-native photogeometric ICP iteration engine (host-side Python mirror).

Thin ctypes binding of the C-ABI in include/icp_amd.h (icp_amd/libicp_amd.so, hand-written HIP
for gfx950).  The classes keep the names and argument meaning of the reference's
cl_algo::ICP::ICPStep<CR,CW> / ICP<CR,CW> (include/ICP/algorithms.hpp:2234-2496 of nlamprian/ICP).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is visible,
calls raise.  Nothing in this package imports oracle/.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["ICP", "ICPStep", "ICPError", "Memory", "ICPStepConfigT", "ICPStepConfigW",
           "PowerMode", "ReduceMode", "TransformKind", "ICPBatch", "batch_partition", "power_method", "KernelObject", "kernel_lms", "kernel_reps", "kernel_weights", "kernel_mean", "kernel_devs", "kernel_s", "ReduceScan", "lib", "lib_path", "reduce", "scan", "ReduceConfig", "synth_pair", "synth_cloud_vga", "punch_holes", "HOLES_SCATTERED", "HOLES_CONTIGUOUS", "synth_pair_scene", "SCENE_CURVED", "SCENE_WALL", "device_count", "DIST_ID"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ICP_AMD_LIB", os.path.join(_HERE, "libicp_amd.so"))   # override: A/B builds of the same ABI

DIST_ID = np.dtype([("dist", np.float32), ("id", np.uint32)])


class ICPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("icp_amd error %d: %s" % (code, msg))
        self.code = code


class ICPStepConfigT:            # include/ICP/algorithms.hpp:1544
    EIGEN = 0
    POWER_METHOD = 1


class ICPStepConfigW:            # include/ICP/algorithms.hpp:1560
    REGULAR = 0
    WEIGHTED = 1


class PowerMode:
    LITERAL = 0
    SQUARED = 1


class ReduceMode:
    REFERENCE_ORDER = 0
    FUSED = 1


class TransformKind:             # ICPTransformConfig (include/ICP/algorithms.hpp:1189) + the second quaternion kernel
    QUATERNION = 0
    QUATERNION_2 = 1
    MATRIX = 2


class Memory:                    # icp_mem in include/icp_amd.h
    F, M, T, TK, MEANS, S, NN_ID, W, SUM_W, REPS, RBC_N, RBC_O, RBC_PERM, RBC_OWNER, RBC_XP, RID, R, RK, NN, QT = range(20)
    # reference spellings (ICPStep::Memory, include/ICP/algorithms.hpp:2241-2267)
    D_IN_F, D_IN_M, D_IO_T, H_IO_T = F, M, T, T


class _State(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("q", C.c_float * 4), ("t", C.c_float * 3), ("s", C.c_float),
                ("Rk", C.c_float * 9), ("qk", C.c_float * 4), ("tk", C.c_float * 3), ("sk", C.c_float),
                ("k", C.c_uint32), ("converged", C.c_uint32), ("power_iterations", C.c_uint32),
                ("reserved", C.c_uint32)]


_lib = None


def lib_path():
    return _SO


def lib():
    """Loads icp_amd/libicp_amd.so; raises if it has not been built (`make` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ICPError(-1, "%s not found: build it with `make` (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % _SO)
    L = C.CDLL(_SO)
    vp, u32, i32, f32, f64 = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_double

    def sig(name, res, *args):
        f = getattr(L, name, None)
        if f is None:
            if "ICP_AMD_LIB" in os.environ:          # an A/B build of an older ABI (tools/diag/ab.sh): entry points it lacks stay unbound
                return
            raise AttributeError("%s does not export %s" % (_SO, name))
        f.restype = res
        f.argtypes = list(args)

    sig("icp_create", i32, C.POINTER(vp), i32, i32, i32)
    sig("icp_destroy", i32, vp)
    sig("icp_init", i32, vp, u32, u32, f32, f32, u32, f64, f64)
    sig("icp_init_batched", i32, vp, u32, u32, u32, f32, f32, u32, f64, f64)
    sig("icp_write", i32, vp, i32, vp, i32)
    sig("icp_write_b", i32, vp, u32, i32, vp, i32)
    sig("icp_read", i32, vp, i32, vp, C.c_size_t)
    sig("icp_read_b", i32, vp, u32, i32, vp, C.c_size_t)
    sig("icp_mem_size", C.c_size_t, vp, i32)
    sig("icp_device_ptr", i32, vp, i32, C.POINTER(vp))
    sig("icp_adopt_device_buffer", i32, vp, i32, vp)
    sig("icp_build_rbc", i32, vp)
    sig("icp_step", i32, vp, i32)
    sig("icp_run", i32, vp, C.POINTER(u32))
    sig("icp_run_stats", i32, vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32))
    sig("icp_set_run_depth", i32, vp, u32, i32)
    sig("icp_run_timeline", i32, vp, C.POINTER(f64))
    sig("icp_set_output_mode", i32, vp, i32)
    sig("icp_launch_stats", i32, vp, C.POINTER(f64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32)
    sig("icp_run_fixed", i32, vp, u32)
    sig("icp_run_fixed_fresh", i32, vp, u32)
    sig("icp_sync", i32, vp)
    sig("icp_get_alpha", i32, vp, C.POINTER(f32))
    sig("icp_set_alpha", i32, vp, f32)
    sig("icp_get_scaling", i32, vp, C.POINTER(f32))
    sig("icp_set_scaling", i32, vp, f32)
    sig("icp_set_metric_scale", i32, vp, f32)
    sig("icp_get_metric_scale", i32, vp, C.POINTER(f32))
    sig("icp_get_max_iterations", i32, vp, C.POINTER(u32))
    sig("icp_set_max_iterations", i32, vp, u32)
    sig("icp_get_angle_threshold", i32, vp, C.POINTER(f64))
    sig("icp_set_angle_threshold", i32, vp, f64)
    sig("icp_get_translation_threshold", i32, vp, C.POINTER(f64))
    sig("icp_set_translation_threshold", i32, vp, f64)
    sig("icp_set_power_mode", i32, vp, i32)
    sig("icp_set_reduce_mode", i32, vp, i32)
    sig("icp_state", i32, vp, C.POINTER(_State))
    sig("icp_state_b", i32, vp, u32, C.POINTER(_State))
    sig("icp_write_cloud", i32, vp, i32, vp, i32)
    sig("icp_transform_cloud", i32, vp, vp, vp, u32)
    sig("icp_transform_cloud_ex", i32, vp, i32, vp, vp, vp, u32)
    sig("icp_track_next", i32, vp, vp, i32, C.POINTER(u32), C.POINTER(i32))
    sig("icp_track_reset", i32, vp)
    sig("icp_track_form", i32, vp, C.POINTER(i32))
    sig("icp_track_submit", i32, vp, vp, i32)
    sig("icp_track_collect", i32, vp, C.POINTER(u32), vp, C.POINTER(i32))
    sig("icp_track_staging", i32, vp, u32, C.POINTER(vp))
    sig("icp_batch_create", i32, C.POINTER(vp), C.POINTER(i32), i32, i32, i32)
    sig("icp_batch_destroy", i32, vp)
    sig("icp_batch_init", i32, vp, u32, u32, u32, f32, f32, u32, f64, f64)
    sig("icp_batch_set_modes", i32, vp, i32, i32)
    sig("icp_batch_write", i32, vp, u32, i32, vp)
    sig("icp_batch_build_rbc", i32, vp)
    sig("icp_batch_run", i32, vp)
    sig("icp_batch_run_fixed", i32, vp, u32, i32)
    sig("icp_batch_state", i32, vp, u32, C.POINTER(_State))
    sig("icp_batch_read", i32, vp, u32, i32, vp, C.c_size_t)
    sig("icp_batch_size", i32, vp, C.POINTER(u32), C.POINTER(u32))
    sig("icp_batch_time_run_fixed", i32, vp, u32, u32, C.POINTER(f64))
    sig("icp_batch_time_run_fixed_slots", i32, vp, u32, u32, u32, C.POINTER(f64), vp)
    sig("icp_batch_partition", i32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32))
    sig("icp_batch_last_error", C.c_char_p, vp)
    sig("icp_time_run_fixed", i32, vp, u32, u32, i32, C.POINTER(f32))
    sig("icp_time_run_fixed_tail", i32, vp, u32, u32, i32, C.POINTER(f32), C.POINTER(u32))
    sig("icp_reset_transform", i32, vp)
    sig("icp_time_kernels", i32, vp, u32, C.POINTER(f32))
    sig("icp_profile_run", i32, vp, u32, vp, C.POINTER(f32))
    sig("icp_time_masked", i32, vp, u32, u32, u32, C.POINTER(f32))
    sig("icp_launches_per_iteration", i32, vp, C.POINTER(u32))
    sig("icp_run_form", i32, vp, C.POINTER(i32))
    sig("icp_search_layout", i32, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32))
    sig("icp_power_method", i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(u32))
    sig("icp_kernel_lms", i32, i32, vp, vp)
    sig("icp_kernel_reps", i32, i32, vp, u32, u32, vp)
    sig("icp_kernel_weights", i32, i32, vp, u32, vp, C.POINTER(f64))
    sig("icp_kernel_mean", i32, i32, i32, vp, vp, vp, f64, u32, vp)
    sig("icp_kernel_devs", i32, i32, vp, vp, vp, u32, vp, vp)
    sig("icp_kernel_s", i32, i32, i32, vp, vp, vp, u32, f32, vp)
    sig("icp_kernel_last_error", C.c_char_p)
    sig("icp_ko_create", i32, C.POINTER(vp), i32, i32, u32, u32, f32)
    sig("icp_ko_destroy", i32, vp)
    sig("icp_ko_adopt", i32, vp, i32, vp)
    sig("icp_ko_device_ptr", i32, vp, i32, C.POINTER(vp))
    sig("icp_ko_slot_bytes", C.c_size_t, vp, i32)
    sig("icp_ko_write", i32, vp, i32, vp)
    sig("icp_ko_read", i32, vp, i32, vp)
    sig("icp_ko_run", i32, vp)
    sig("icp_ko_set_scaling", i32, vp, f32)
    sig("icp_reduce", i32, i32, i32, vp, u32, u32, vp)
    sig("icp_scan", i32, i32, i32, vp, u32, u32, vp)
    sig("icp_reduce_scan_last_error", C.c_char_p)
    sig("icp_rs_create", i32, C.POINTER(vp), i32, i32, u32, u32)
    sig("icp_rs_write", i32, vp, vp)
    sig("icp_rs_run", i32, vp)
    sig("icp_rs_read", i32, vp, vp)
    sig("icp_rs_device_ptr", i32, vp, i32, C.POINTER(vp))
    sig("icp_rs_time", i32, vp, u32, C.POINTER(f32))
    sig("icp_rs_destroy", i32, vp)
    sig("icp_last_error", C.c_char_p, vp)
    sig("icp_version", C.c_char_p)
    sig("icp_device_count", i32, C.POINTER(i32))
    sig("icp_device_pci_bus_id", i32, i32, C.c_char_p, C.c_size_t)
    sig("icp_numa_cpulist", i32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t)
    sig("icp_batch_slot_cpus", i32, vp, u32, C.c_char_p, C.c_size_t)
    sig("icp_synth_pair", i32, C.c_uint64, u32, f32, vp, vp, f32, f32, f32, vp, vp)
    sig("icp_synth_cloud_vga", i32, C.c_uint64, i32, vp)
    sig("icp_track_register_source", i32, vp, vp, C.c_size_t)
    sig("icp_track_unregister_source", i32, vp, vp)
    sig("icp_synth_pair_scene", i32, C.c_uint64, u32, i32, f32, vp, vp, f32, f32, vp, vp, vp)
    sig("icp_synth_punch_holes", i32, C.c_uint64, u32, u32, i32, f32, i32, vp)
    _lib = L
    return L


class ReduceConfig:              # include/ICP/algorithms.hpp:52-57
    MIN, MAX, SUM = 0, 1, 2


def reduce(a, config=ReduceConfig.SUM, device=0):
    """Row-wise reduce of a rows x cols array — cl_algo::ICP::Reduce<C,T> (MIN: float, MAX: uint32, SUM: float)."""
    dt = np.uint32 if config == ReduceConfig.MAX else np.float32
    a = np.ascontiguousarray(a, dt)
    if a.ndim != 2:
        raise ValueError("expected a rows x cols array")
    out = np.empty(a.shape[0], dt)
    rc = lib().icp_reduce(device, config, _p(a), a.shape[1], a.shape[0], _p(out))
    if rc:
        raise ICPError(rc, lib().icp_reduce_scan_last_error().decode())
    return out


def scan(a, inclusive=True, device=0):
    """Row-wise integer scan of a rows x cols array — cl_algo::ICP::Scan<INCLUSIVE|EXCLUSIVE,int>."""
    a = np.ascontiguousarray(a, np.int32)
    if a.ndim != 2:
        raise ValueError("expected a rows x cols array")
    out = np.empty_like(a)
    rc = lib().icp_scan(device, int(inclusive), _p(a), a.shape[1], a.shape[0], _p(out))
    if rc:
        raise ICPError(rc, lib().icp_reduce_scan_last_error().decode())
    return out


class ReduceScan:
    """Resident Reduce / Scan object (icp_rs_*): kind = ReduceConfig.MIN / MAX / SUM, or "inclusive" / "exclusive" scan.
    Device buffers live with the object; run() enqueues kernels only; time(reps) = mean microseconds per run."""

    def __init__(self, kind, cols, rows, device=0):
        self._L = lib()
        self._kind = {"inclusive": 3, "exclusive": 4}.get(kind, kind)
        self._dt = np.int32 if self._kind >= 3 else (np.uint32 if self._kind == ReduceConfig.MAX else np.float32)
        self.cols, self.rows = cols, rows
        self._r = C.c_void_p()
        rc = self._L.icp_rs_create(C.byref(self._r), device, self._kind, cols, rows)
        if rc:
            self._r = None
            raise ICPError(rc, self._L.icp_reduce_scan_last_error().decode())

    def _chk(self, rc):
        if rc:
            raise ICPError(rc, self._L.icp_reduce_scan_last_error().decode())

    def write(self, a):
        a = np.ascontiguousarray(a, self._dt)
        if a.shape != (self.rows, self.cols):
            raise ValueError("expected a %d x %d array" % (self.rows, self.cols))
        self._chk(self._L.icp_rs_write(self._r, _p(a)))

    def run(self):
        self._chk(self._L.icp_rs_run(self._r))

    def read(self):
        out = np.empty((self.rows, self.cols) if self._kind >= 3 else self.rows, self._dt)
        self._chk(self._L.icp_rs_read(self._r, _p(out)))
        return out

    def time(self, reps=100):
        us = C.c_float()
        self._chk(self._L.icp_rs_time(self._r, reps, C.byref(us)))
        return us.value

    def close(self):
        if getattr(self, "_r", None):
            self._L.icp_rs_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def power_method(S, means, rot=ICPStepConfigT.POWER_METHOD, mode=PowerMode.LITERAL, device=0):
    """ICPPowerMethod (include/ICP/algorithms.hpp:1451-1537): S[11], means[8] -> (Tk[8], Rk[3,3], loop trips) by one wave of the
    engine's rotation solver on the device (rot = EIGEN: the SVD branch)."""
    S = np.ascontiguousarray(S, np.float32).reshape(-1)
    means = np.ascontiguousarray(means, np.float32).reshape(-1)
    if S.size != 11 or means.size != 8:
        raise ValueError("expected S[11] and means[8]")
    Tk, Rk, it = np.empty(8, np.float32), np.empty(9, np.float32), C.c_uint32()
    rc = lib().icp_power_method(device, rot, mode, _p(S), _p(means), _p(Tk), _p(Rk), C.byref(it))
    if rc:
        raise ICPError(rc, lib().icp_last_error(None).decode())
    return Tk, Rk.reshape(3, 3), it.value


def _kchk(rc):
    if rc:
        raise ICPError(rc, lib().icp_kernel_last_error().decode())


def kernel_lms(cloud, device=0):
    """ICPLMs: 640 x 480 float8 cloud -> the 128 x 128 landmarks (getLMs)."""
    cloud = np.ascontiguousarray(cloud, np.float32).reshape(-1, 8)
    if cloud.shape[0] != 640 * 480:
        raise ValueError("expected a 640x480 float8 cloud")
    out = np.empty((16384, 8), np.float32)
    _kchk(lib().icp_kernel_lms(device, _p(cloud), _p(out)))
    return out


def kernel_reps(F, nr, device=0):
    """ICPReps: nr representatives of a square landmark set (getReps with the grid side sqrt(m))."""
    F = np.ascontiguousarray(F, np.float32).reshape(-1, 8)
    out = np.empty((nr, 8), np.float32)
    _kchk(lib().icp_kernel_reps(device, _p(F), F.shape[0], nr, _p(out)))
    return out


def kernel_weights(nn_id, device=0):
    """ICPWeights: {dist, id}[n] -> (W[n], sum of weights)."""
    nn_id = np.ascontiguousarray(nn_id, DIST_ID)
    W, sw = np.empty(nn_id.shape[0], np.float32), C.c_double()
    _kchk(lib().icp_kernel_weights(device, _p(nn_id), nn_id.shape[0], _p(W), C.byref(sw)))
    return W, sw.value


def kernel_mean(F, M, W=None, sum_w=1.0, device=0):
    """ICPMean<REGULAR> (W is None) / ICPMean<WEIGHTED>: [mean_F, 0 | mean_M, 0]."""
    F = np.ascontiguousarray(F, np.float32).reshape(-1, 8)
    M = np.ascontiguousarray(M, np.float32).reshape(-1, 8)
    Wp = None if W is None else np.ascontiguousarray(W, np.float32)
    out = np.empty(8, np.float32)
    _kchk(lib().icp_kernel_mean(device, int(W is not None), _p(F), _p(M), None if Wp is None else _p(Wp), float(sum_w), F.shape[0], _p(out)))
    return out


def kernel_devs(F, M, mean8, device=0):
    """ICPDevs: (DF[n, 4], DM[n, 4])."""
    F = np.ascontiguousarray(F, np.float32).reshape(-1, 8)
    M = np.ascontiguousarray(M, np.float32).reshape(-1, 8)
    mean8 = np.ascontiguousarray(mean8, np.float32)
    DF, DM = np.empty((F.shape[0], 4), np.float32), np.empty((F.shape[0], 4), np.float32)
    _kchk(lib().icp_kernel_devs(device, _p(F), _p(M), _p(mean8), F.shape[0], _p(DF), _p(DM)))
    return DF, DM


def kernel_s(DM, DF, W=None, c=1e-6, device=0):
    """ICPS<REGULAR> (W is None) / ICPS<WEIGHTED>: S[11]."""
    DM = np.ascontiguousarray(DM, np.float32).reshape(-1, 4)
    DF = np.ascontiguousarray(DF, np.float32).reshape(-1, 4)
    Wp = None if W is None else np.ascontiguousarray(W, np.float32)
    out = np.empty(11, np.float32)
    _kchk(lib().icp_kernel_s(device, int(W is not None), _p(DM), _p(DF), None if Wp is None else _p(Wp), DM.shape[0], c, _p(out)))
    return out


class KernelObject:
    """Resident per-kernel object (icp_ko_*): the reference's ICPLMs / ICPReps / ICPWeights / ICPMean<> / ICPDevs / ICPS<> with device
    buffers that live with the object.  kind: "lms", "reps", "weights", "mean", "mean_weighted", "devs", "s", "s_weighted"; slots (Memory
    objects) as listed in include/icp_amd.h.  get(slot) = the device pointer; adopt(slot, ptr) before the slot's first use wires another
    object's buffer in (no copy); run() enqueues kernels only."""
    KINDS = {"lms": 0, "reps": 1, "weights": 2, "mean": 3, "mean_weighted": 4, "devs": 5, "s": 6, "s_weighted": 7}

    def __init__(self, kind, n=0, aux=0, c=1e-6, device=0):
        self._L = lib()
        self._k = C.c_void_p()
        _kchk(self._L.icp_ko_create(C.byref(self._k), device, self.KINDS[kind], n, aux, c))

    def close(self):
        if getattr(self, "_k", None):
            self._L.icp_ko_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self, slot):
        p = C.c_void_p()
        _kchk(self._L.icp_ko_device_ptr(self._k, slot, C.byref(p)))
        return p.value

    def adopt(self, slot, device_ptr):
        _kchk(self._L.icp_ko_adopt(self._k, slot, C.c_void_p(device_ptr)))

    def write(self, slot, a):
        a = np.ascontiguousarray(a)
        if a.nbytes != self._L.icp_ko_slot_bytes(self._k, slot):
            raise ValueError("slot %d holds %d bytes, got %d" % (slot, self._L.icp_ko_slot_bytes(self._k, slot), a.nbytes))
        _kchk(self._L.icp_ko_write(self._k, slot, _p(a)))

    def read(self, slot, dtype=np.float32):
        out = np.empty(self._L.icp_ko_slot_bytes(self._k, slot) // np.dtype(dtype).itemsize, dtype)
        _kchk(self._L.icp_ko_read(self._k, slot, _p(out)))
        return out

    def run(self):
        _kchk(self._L.icp_ko_run(self._k))


def device_count():
    n = C.c_int(0)
    lib().icp_device_count(C.byref(n))
    return n.value


def device_pci_bus_id(device):
    """PCI bus id of a device ("0000:c1:00.0")."""
    buf = C.create_string_buffer(64)
    rc = lib().icp_device_pci_bus_id(device, buf, len(buf))
    if rc:
        raise ICPError(rc, "icp_device_pci_bus_id (%d)" % device)
    return buf.value.decode()


def numa_cpulist(pci_bus_id, sysfs_root=None):
    """cpulist of the NUMA node a PCI device hangs on, from a sysfs tree ("" when the tree has no answer): where icp_batch_create pins
    the host thread of a device slot by default (host code only: works without a GPU)."""
    buf = C.create_string_buffer(4096)
    rc = lib().icp_numa_cpulist(sysfs_root.encode() if sysfs_root else None, pci_bus_id.encode(), buf, len(buf))
    if rc:
        raise ICPError(rc, "icp_numa_cpulist")
    return buf.value.decode()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def synth_pair(side, seed=0x1C9D5EED, rot_deg=3.0, axis=(0.3, 0.9, 0.1), t=(25.0, -10.0, 15.0),
               noise_mm=1.0, noise_rgb=0.01, zero_fraction=0.0):
    """Synthetic fixed/moving landmark pair (side*side points each), SURVEY.md §8d. Host only."""
    F = np.empty((side * side, 8), np.float32)
    M = np.empty((side * side, 8), np.float32)
    ax = np.asarray(axis, np.float32)
    tt = np.asarray(t, np.float32)
    rc = lib().icp_synth_pair(seed, side, rot_deg, _p(ax), _p(tt), noise_mm, noise_rgb, zero_fraction, _p(F), _p(M))
    if rc:
        raise ICPError(rc, "icp_synth_pair")
    return F, M


def synth_cloud_vga(seed=0x1C9D5EED, moved=False):
    """Synthetic 640x480 float8 cloud; `moved` = frame number of a sequence (False / 0: the scene, True / 1: one step)."""
    cloud = np.empty((480 * 640, 8), np.float32)
    rc = lib().icp_synth_cloud_vga(seed, int(moved), _p(cloud))
    if rc:
        raise ICPError(rc, "icp_synth_cloud_vga")
    return cloud


HOLES_SCATTERED, HOLES_CONTIGUOUS = 0, 1
SCENE_CURVED, SCENE_WALL = 0, 1


def synth_pair_scene(side, scene=SCENE_CURVED, seed=0x1C9D5EED, rot_deg=3.0, axis=(0.3, 0.9, 0.1), t=(25.0, -10.0, 15.0),
                     noise_mm=1.0, noise_rgb=0.01):
    """(F, M, T_true): a synthetic pair of the chosen scene and the ground truth [q | t, 1] that maps M onto F.  SCENE_WALL: the
    reference's kg_pc8d_wall stand-in (data/README.md:11-16) — a textured plane moved in its own plane. Host only."""
    F = np.empty((side * side, 8), np.float32)
    M = np.empty((side * side, 8), np.float32)
    T = np.empty(8, np.float32)
    ax = np.asarray(axis, np.float32)
    tt = np.asarray(t, np.float32)
    rc = lib().icp_synth_pair_scene(seed, side, scene, rot_deg, _p(ax), _p(tt), noise_mm, noise_rgb, _p(F), _p(M), _p(T))
    if rc:
        raise ICPError(rc, "icp_synth_pair_scene")
    return F, M, T


def punch_holes(cloud, width, height, pattern=HOLES_SCATTERED, fraction=0.1, keep_rgb=True, seed=0x1C9D5EED):
    """A copy of `cloud` (width x height float8 points) with a Kinect frame's invalid pixels: xyz = 0, the colour kept
    (reference src/kinect_frame_grabber.cpp:246-262) unless keep_rgb is False (all holes identical). Host only."""
    out = np.ascontiguousarray(cloud, np.float32).copy()
    assert out.size == width * height * 8
    rc = lib().icp_synth_punch_holes(seed, width, height, pattern, fraction, int(bool(keep_rgb)), _p(out))
    if rc:
        raise ICPError(rc, "icp_synth_punch_holes")
    return out


_MEM_DTYPE = {
    Memory.F: (np.float32, 8), Memory.M: (np.float32, 8), Memory.RBC_XP: (np.float32, 8),
    Memory.T: (np.float32, None), Memory.TK: (np.float32, None), Memory.MEANS: (np.float32, None),
    Memory.S: (np.float32, None), Memory.NN_ID: (DIST_ID, None), Memory.W: (np.float32, None),
    Memory.SUM_W: (np.float64, None), Memory.REPS: (np.float32, 8), Memory.RBC_N: (np.uint32, None),
    Memory.RBC_O: (np.uint32, None), Memory.RBC_PERM: (np.uint32, None), Memory.RBC_OWNER: (np.uint32, None),
    Memory.RID: (np.uint32, None), Memory.R: (np.float32, 3), Memory.RK: (np.float32, 3),
    Memory.NN: (np.float32, 4), Memory.QT: (np.float32, 4),
}


class ICPStep:
    """One ICP iteration engine — mirror of cl_algo::ICP::ICPStep<CR,CW>.

    Reference                                   here
    ICPStep(env, infoRBC, infoICP)              ICPStep(device=0, CR=POWER_METHOD, CW=WEIGHTED)
    init(m, nr, a=1e2, c=1e-6, staging)         init(m, nr, a=1e2, c=1e-6, batch=1)
    write(mem, ptr, block)                      write(mem, array, block=False, batch_index=0)
    read(mem, block)                            read(mem, batch_index=0) -> numpy array
    buildRBC() / run(config=False)              buildRBC() / run(config=False)
    getAlpha/setAlpha/getScaling/setScaling     same names
    members Rk qk tk sk R q t s                 properties of the same names (blocking read)
    """

    def __init__(self, device=0, CR=ICPStepConfigT.POWER_METHOD, CW=ICPStepConfigW.WEIGHTED):
        self._L = lib()
        self._h = C.c_void_p()
        rc = self._L.icp_create(C.byref(self._h), device, CR, CW)
        if rc:
            msg = self._L.icp_last_error(None).decode()
            self._h = None
            raise ICPError(rc, msg)
        self.m = self.nr = 0
        self.batch = 1
        self._max_it, self._ang, self._tra = 40, 0.001, 0.01

    def close(self):
        if getattr(self, "_h", None):
            self._L.icp_destroy(self._h)           # (unregisters what track_register page-locked)
            self._h = None
            self.__dict__.pop("_registered", None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise ICPError(rc, self._L.icp_last_error(self._h).decode())

    # -- reference API ---------------------------------------------------------------------
    def init(self, m, nr, a=1e2, c=1e-6, batch=1):
        self._chk(self._L.icp_init_batched(self._h, batch, m, nr, a, c, self._max_it, self._ang, self._tra))
        self.m, self.nr, self.batch = m, nr, batch
        self.__dict__.pop("_registered", None)     # (icp_init unregisters the frame buffers of an earlier configuration)

    def write(self, mem=Memory.D_IN_F, ptr=None, block=False, batch_index=0):
        arr = None
        if ptr is not None:
            arr = np.ascontiguousarray(ptr, dtype=np.float32)
            want = 8 if mem == Memory.T else self.m * 8
            if arr.size != want:
                raise ValueError("write(%d): expected %d floats, got %d" % (mem, want, arr.size))
        self._chk(self._L.icp_write_b(self._h, batch_index, mem, _p(arr) if arr is not None else None, int(block)))

    def read(self, mem=Memory.H_IO_T, batch_index=0):
        nbytes = self._L.icp_mem_size(self._h, mem)
        if nbytes == 0:
            raise ValueError("unknown memory object %r" % (mem,))
        dt, cols = _MEM_DTYPE[mem]
        out = np.empty(nbytes // np.dtype(dt).itemsize, dt)
        self._chk(self._L.icp_read_b(self._h, batch_index, mem, _p(out), nbytes))
        return out.reshape(-1, cols) if cols else out

    def buildRBC(self):
        self._chk(self._L.icp_build_rbc(self._h))

    def run(self, config=False):
        """ICPStep::run — one iteration (enqueue only)."""
        self._chk(self._L.icp_step(self._h, int(config)))

    def getAlpha(self):
        v = C.c_float()
        self._chk(self._L.icp_get_alpha(self._h, C.byref(v)))
        return v.value

    def setAlpha(self, a):
        self._chk(self._L.icp_set_alpha(self._h, a))

    def getScaling(self):
        v = C.c_float()
        self._chk(self._L.icp_get_scaling(self._h, C.byref(v)))
        return v.value

    def setScaling(self, c):
        self._chk(self._L.icp_set_scaling(self._h, c))

    def setMetricScale(self, f_g):
        """Absolute scale of the metric: reported dist = f_g (geo + a pho); matters for the weights only."""
        self._chk(self._L.icp_set_metric_scale(self._h, f_g))

    def getMetricScale(self):
        v = C.c_float()
        self._chk(self._L.icp_get_metric_scale(self._h, C.byref(v)))
        return v.value

    # -- extensions ------------------------------------------------------------------------
    def setPowerMode(self, mode):
        self._chk(self._L.icp_set_power_mode(self._h, mode))

    def setReduceMode(self, mode):
        self._chk(self._L.icp_set_reduce_mode(self._h, mode))

    def sync(self):
        self._chk(self._L.icp_sync(self._h))

    def run_fixed(self, iterations):
        """ICP::run(timer) — exactly `iterations` steps, no convergence test (enqueue only)."""
        self._chk(self._L.icp_run_fixed(self._h, iterations))

    def run_fixed_fresh(self, iterations):
        """reset_transform() + run_fixed(iterations) as one graph (enqueue only)."""
        self._chk(self._L.icp_run_fixed_fresh(self._h, iterations))

    def write_cloud(self, which, cloud):
        cloud = np.ascontiguousarray(cloud, np.float32)
        if cloud.size != 640 * 480 * 8:
            raise ValueError("expected a 640x480 float8 cloud")
        self._chk(self._L.icp_write_cloud(self._h, which, _p(cloud), 1))

    def track_next(self, cloud, warm_start=False):
        """Frame-to-frame tracking: feeds the next 640x480 float8 frame; returns k (iterations) once there is a previous
        frame to register against, else None.  One upload per frame; the previous landmarks stay on the device."""
        cloud = np.ascontiguousarray(cloud, np.float32)
        if cloud.size != 640 * 480 * 8:
            raise ValueError("expected a 640x480 float8 cloud")
        k, reg = C.c_uint32(), C.c_int()
        self._chk(self._L.icp_track_next(self._h, _p(cloud), int(warm_start), C.byref(k), C.byref(reg)))
        return k.value if reg.value else None

    def track_form(self):
        """1: tracked frames alternate between two streams behind device-side gates; 0: one stream, host-ordered (icp_track_form)."""
        g = C.c_int()
        self._chk(self._L.icp_track_form(self._h, C.byref(g)))
        return g.value

    def track_reset(self):
        self._chk(self._L.icp_track_reset(self._h))

    def track_submit(self, cloud, warm_start=False):
        """Enqueues a frame (upload of its band + getLMs on the copy stream, buildRBC + as many iterations of a host-driven checked run as the
        last two registrations suggest; later track_* calls top it up): returns at once.
        `cloud`: a 640x480 float8 array, or the index (0 / 1) of one of the engine's pinned frame buffers (track_staging)."""
        if isinstance(cloud, int):
            ptr = C.c_void_p(self.track_staging(cloud).ctypes.data)
        else:
            cloud = np.ascontiguousarray(cloud, np.float32)
            if cloud.size != 640 * 480 * 8:
                raise ValueError("expected a 640x480 float8 cloud")
            ptr = _p(cloud)
        self._chk(self._L.icp_track_submit(self._h, ptr, int(warm_start)))

    def track_collect(self):
        """Result of the oldest frame in flight (blocking): (k, T[8]) or None for the first frame of a sequence."""
        k, reg, T = C.c_uint32(), C.c_int(), np.empty(8, np.float32)
        self._chk(self._L.icp_track_collect(self._h, C.byref(k), _p(T), C.byref(reg)))
        return (k.value, T) if reg.value else None

    def track_staging(self, slot):
        """Pinned host buffer `slot` (0 / 1) for a whole frame, as a (307200, 8) float32 array: fill it, then track_submit(slot)."""
        p = C.c_void_p()
        self._chk(self._L.icp_track_staging(self._h, slot, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(640 * 480, 8))

    def track_register(self, frames):
        """Page-locks the caller's own frame array(s) (C-contiguous float32, one or more 640x480x8 frames) as DMA sources."""
        a = np.asarray(frames)
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.nbytes >= 640 * 480 * 32
        self._chk(self._L.icp_track_register_source(self._h, _p(a), a.nbytes))
        # (the array stays alive while its pages are locked: the engine unregisters what is left at init / close)
        self.__dict__.setdefault("_registered", {})[a.ctypes.data] = a

    def track_unregister(self, frames):
        a = np.asarray(frames)
        self._chk(self._L.icp_track_unregister_source(self._h, _p(a)))
        self.__dict__.get("_registered", {}).pop(a.ctypes.data, None)

    def track_pipelined(self, frames, warm_start=False, depth=2, pinned=False):
        """Feeds a sequence with `depth` frames in flight; returns [None | (k, T)] per frame.  pinned: every frame is first copied
        into one of the engine's two pinned frame buffers (what a capture loop that writes there directly would skip)."""
        out, inflight = [], 0
        for i, f in enumerate(frames):
            if inflight >= depth:
                out.append(self.track_collect())
                inflight -= 1
            if pinned:
                self.track_staging(i & 1)[...] = np.asarray(f, np.float32).reshape(-1, 8)
                self.track_submit(i & 1, warm_start)
            else:
                self.track_submit(f, warm_start)
            inflight += 1
        while inflight:
            out.append(self.track_collect())
            inflight -= 1
        return out

    def transform_cloud(self, cloud, T=None, kind=TransformKind.QUATERNION):
        """ICPTransform: the handle's current T (default), or an explicit one — [q | t, s] for the quaternion kinds,
        a row-major 4x4 for TransformKind.MATRIX."""
        cloud = np.ascontiguousarray(cloud, np.float32).reshape(-1, 8)
        out = np.empty_like(cloud)
        if T is None:
            if kind != TransformKind.QUATERNION:
                raise ValueError("an explicit T is required for this kind")
            self._chk(self._L.icp_transform_cloud(self._h, _p(cloud), _p(out), cloud.shape[0]))
        else:
            T = np.ascontiguousarray(T, np.float32).reshape(-1)
            if T.size != (16 if kind == TransformKind.MATRIX else 8):
                raise ValueError("T has %d floats" % T.size)
            self._chk(self._L.icp_transform_cloud_ex(self._h, kind, _p(T), _p(cloud), _p(out), cloud.shape[0]))
        return out

    def time_run_fixed(self, iterations, reps, from_identity=False):
        ms = C.c_float()
        self._chk(self._L.icp_time_run_fixed(self._h, iterations, reps, int(from_identity), C.byref(ms)))
        return ms.value

    def time_run_fixed_tail(self, iterations, reps, from_identity=False):
        """`reps` passes, HIP events around all but the first: (ms over the timed passes, number of timed passes)."""
        ms, n = C.c_float(), C.c_uint32()
        self._chk(self._L.icp_time_run_fixed_tail(self._h, iterations, reps, int(from_identity), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_transform(self):
        """T <- identity, k <- 0 (enqueue only)."""
        self._chk(self._L.icp_reset_transform(self._h))

    def launches_per_iteration(self):
        """Kernel launches per iteration of run() / run_fixed(): 4 (reference order), 2 (fused) or 1 (fused, chained)."""
        n = C.c_uint32()
        self._chk(self._L.icp_launches_per_iteration(self._h, C.byref(n)))
        return n.value

    def run_form(self):
        """0 separate launches per stage, 1 chained (one launch per iteration)."""
        f = C.c_int()
        self._chk(self._L.icp_run_form(self._h, C.byref(f)))
        return f.value

    def search_layout(self):
        """(dense, representatives per LDS tile, stage-2 form) of the search kernel in use (diagnostic, see icp_search_layout)."""
        d, t, s2 = C.c_int(), C.c_int(), C.c_int()
        self._chk(self._L.icp_search_layout(self._h, C.byref(d), C.byref(t), C.byref(s2)))
        return d.value, t.value, s2.value

    def time_masked(self, mask, iterations=40, reps=20):
        """us per iteration of a graph holding only the kernels in `mask` (diagnostic)."""
        ms = C.c_float()
        self._chk(self._L.icp_time_masked(self._h, mask, iterations, reps, C.byref(ms)))
        return ms.value * 1e3 / (iterations * reps)

    def profile_run(self, iterations=40, print_table=False):
        """ICP::run(timer) (include/ICP/algorithms.hpp:2482-2494): `iterations` steps with per-step, per-stage times.
        Returns (table[iterations][4] in ms: search, means, sij, finalize; total ms); print_table: the reference's
        ProfilingInfo-style summary (mean / min / max / total per stage)."""
        t = np.zeros((iterations, 4), np.float32)
        tot = C.c_float()
        self._chk(self._L.icp_profile_run(self._h, iterations, _p(t), C.byref(tot)))
        if print_table:
            print(" ICP::run (timer): %d steps, %.3f ms in all" % (iterations, tot.value))
            print(" %-10s %10s %10s %10s %10s" % ("stage", "mean [us]", "min [us]", "max [us]", "total [ms]"))
            for k, name in enumerate(("search", "means", "sij", "finalize")):
                c = t[:, k] * 1e3
                print(" %-10s %10.2f %10.2f %10.2f %10.3f" % (name, c.mean(), c.min(), c.max(), c.sum() / 1e3))
            s = t.sum(1) * 1e3
            print(" %-10s %10.2f %10.2f %10.2f %10.3f" % ("step", s.mean(), s.min(), s.max(), s.sum() / 1e3))
        return t, tot.value

    def time_kernels(self, reps):
        out = (C.c_float * 4)()
        self._chk(self._L.icp_time_kernels(self._h, reps, out))
        return dict(zip(("search", "means", "sij", "finalize"), [float(v) for v in out]))

    def state(self, batch_index=0):
        st = _State()
        self._chk(self._L.icp_state_b(self._h, batch_index, C.byref(st)))
        return st

    def _vec(self, name, shape=None):
        a = np.array(getattr(self.state(), name), dtype=np.float32)
        return a.reshape(shape) if shape else a

    R = property(lambda s: s._vec("R", (3, 3)))
    Rk = property(lambda s: s._vec("Rk", (3, 3)))
    q = property(lambda s: s._vec("q"))
    qk = property(lambda s: s._vec("qk"))
    t = property(lambda s: s._vec("t"))
    tk = property(lambda s: s._vec("tk"))
    s = property(lambda s: float(s.state().s))
    sk = property(lambda s: float(s.state().sk))
    k = property(lambda s: int(s.state().k))


class ICP(ICPStep):
    """Full registration loop — mirror of cl_algo::ICP::ICP<CR,CW> (include/ICP/algorithms.hpp:2433-2496)."""

    def init(self, m, nr, a=1e2, c=1e-6, max_iterations=40, angle_threshold=0.001,
             translation_threshold=0.01, batch=1):
        self._max_it, self._ang, self._tra = max_iterations, angle_threshold, translation_threshold
        super().init(m, nr, a, c, batch)

    def run(self):
        """ICP::run — iterate until check() stops (blocking); returns k."""
        k = C.c_uint32()
        self._chk(self._L.icp_run(self._h, C.byref(k)))
        return k.value

    def step(self, config=False):
        ICPStep.run(self, config)

    def run_stats(self):
        """(iteration launches enqueued, final k, launches past the last live iteration) of the last finished checked run."""
        n, k, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self._L.icp_run_stats(self._h, C.byref(n), C.byref(k), C.byref(d)))
        return n.value, k.value, d.value

    def run_timeline(self):
        """Host timeline of the last run() in us: [0, launches enqueued, first progress seen, decided, end enqueued, FINAL seen]."""
        t = (C.c_double * 6)()
        self._chk(self._L.icp_run_timeline(self._h, t))
        return [float(x) for x in t]

    def launch_stats(self, reset=False):
        """(longest launch call in us, calls slower than 10 us, calls) of the checked runs since init / the last reset."""
        mx, slow, tot = C.c_double(), C.c_uint64(), C.c_uint64()
        self._chk(self._L.icp_launch_stats(self._h, C.byref(mx), C.byref(slow), C.byref(tot), int(reset)))
        return mx.value, slow.value, tot.value

    def set_output_mode(self, every_iteration=False):
        """Per-query outputs of checked runs: stored by every iteration, or (default) reproduced on the first read (icp_set_output_mode)."""
        self._chk(self._L.icp_set_output_mode(self._h, int(every_iteration)))

    def set_run_depth(self, depth=3, adaptive=True):
        """Launches kept queued behind the one in flight by checked runs; adaptive=False: one graph of max_iterations launches."""
        self._chk(self._L.icp_set_run_depth(self._h, depth, int(adaptive)))

    def getMaxIterations(self):
        v = C.c_uint32()
        self._chk(self._L.icp_get_max_iterations(self._h, C.byref(v)))
        return v.value

    def setMaxIterations(self, n):
        self._chk(self._L.icp_set_max_iterations(self._h, n))
        self._max_it = n

    def getAngleThreshold(self):
        v = C.c_double()
        self._chk(self._L.icp_get_angle_threshold(self._h, C.byref(v)))
        return v.value

    def setAngleThreshold(self, deg):
        self._chk(self._L.icp_set_angle_threshold(self._h, deg))
        self._ang = deg

    def getTranslationThreshold(self):
        v = C.c_double()
        self._chk(self._L.icp_get_translation_threshold(self._h, C.byref(v)))
        return v.value

    def setTranslationThreshold(self, mm):
        self._chk(self._L.icp_set_translation_threshold(self._h, mm))
        self._tra = mm


def batch_partition(registrations, n_slots, i):
    """(slot, index inside the slot, registrations of that slot) of registration i — icp_batch_partition (no device needed)."""
    slot, idx, cnt = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().icp_batch_partition(registrations, n_slots, i, C.byref(slot), C.byref(idx), C.byref(cnt))
    if rc:
        raise ICPError(rc, "icp_batch_partition: bad arguments")
    return slot.value, idx.value, cnt.value


class ICPBatch:
    """B independent registrations over a list of devices inside the library (icp_batch_*, include/icp_amd.h):
    registration i -> device slot i mod n, one host thread and one stream per slot, no collective."""

    def __init__(self, devices, CR=ICPStepConfigT.POWER_METHOD, CW=ICPStepConfigW.WEIGHTED):
        self._L = lib()
        self._b = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._L.icp_batch_create(C.byref(self._b), devs, len(devices), CR, CW)
        if rc:
            msg = self._L.icp_batch_last_error(None).decode()
            self._b = None
            raise ICPError(rc, msg)
        self.m = 0

    def _chk(self, rc):
        if rc:
            raise ICPError(rc, self._L.icp_batch_last_error(self._b).decode())

    def slot_cpus(self, slot):
        """The CPUs the host thread of a device slot was pinned to ([] = not pinned)."""
        buf = C.create_string_buffer(16384)
        self._chk(self._L.icp_batch_slot_cpus(self._b, slot, buf, len(buf)))
        return [int(x) for x in buf.value.decode().split(",") if x]

    def close(self):
        if getattr(self, "_b", None):
            self._L.icp_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, registrations, m, nr, a=1e2, c=1e-6, max_iterations=40, angle_threshold=0.001, translation_threshold=0.01):
        self._chk(self._L.icp_batch_init(self._b, registrations, m, nr, a, c, max_iterations, angle_threshold, translation_threshold))
        self.m, self.registrations = m, registrations

    def set_modes(self, reduce_mode, power_mode):
        self._chk(self._L.icp_batch_set_modes(self._b, reduce_mode, power_mode))

    def write(self, i, mem, ptr):
        arr = np.ascontiguousarray(ptr, dtype=np.float32)
        want = 8 if mem == Memory.T else self.m * 8
        if arr.size != want:
            raise ValueError("write(%d): expected %d floats, got %d" % (mem, want, arr.size))
        self._chk(self._L.icp_batch_write(self._b, i, mem, _p(arr)))

    def buildRBC(self):
        self._chk(self._L.icp_batch_build_rbc(self._b))

    def run(self):
        self._chk(self._L.icp_batch_run(self._b))

    def run_fixed(self, iterations, from_identity=True):
        self._chk(self._L.icp_batch_run_fixed(self._b, iterations, int(from_identity)))

    def time_run_fixed(self, iterations, reps):
        """Wall-clock seconds of `reps` passes on all slots at once."""
        s = C.c_double()
        self._chk(self._L.icp_batch_time_run_fixed(self._b, iterations, reps, C.byref(s)))
        return s.value

    def time_run_fixed_slots(self, iterations, reps, warmup=1):
        """(wall-clock seconds of `reps` passes on all slots at once, HIP-event ms of every slot's own passes)."""
        s = C.c_double()
        n = C.c_uint32()
        self._chk(self._L.icp_batch_size(self._b, None, C.byref(n)))
        ms = np.zeros(n.value, np.float32)
        self._chk(self._L.icp_batch_time_run_fixed_slots(self._b, iterations, reps, warmup, C.byref(s), _p(ms)))
        return s.value, ms

    def state(self, i):
        st = _State()
        self._chk(self._L.icp_batch_state(self._b, i, C.byref(st)))
        return st

    def read(self, i, mem):
        dt, cols = _MEM_DTYPE[mem]
        sizes = {Memory.T: 32, Memory.TK: 32, Memory.MEANS: 32, Memory.S: 44, Memory.NN_ID: self.m * 8, Memory.R: 36, Memory.RK: 36,
                 Memory.F: self.m * 32, Memory.M: self.m * 32, Memory.W: self.m * 4, Memory.RID: self.m * 4}
        nbytes = sizes[mem]
        out = np.empty(nbytes // np.dtype(dt).itemsize, dt)
        self._chk(self._L.icp_batch_read(self._b, i, mem, _p(out), nbytes))
        return out.reshape(-1, cols) if cols else out
