"""Synthetic workloads of BASELINE.json's configs, shared by bench.py, the parity tests and the golden
fixtures.  Host-side only.

Config 4 — independent frame pairs, 64 per GPU: registration i of a job has seed base + i; rotation and translation vary
with i so that registrations stop at different iterations."""
import zlib

import numpy as np

# name -> (landmark grid side, |R|): |F| = |M| = side^2.  A = configs[1] (the headline), B = configs[2], C = configs[4].
CONFIGS = {"A": (128, 256), "B": (256, 1024), "C": (1024, 4096)}

SIDE, NR, M_POINTS, PER_GPU = 128, 256, 16384, 64
A, C_ = 2e2, 1e-6
BASE_SEED = 0x1C9D5EED
CHECKED = (0, 9, 18, 27, 36, 45, 54, 63)          # the registrations the parity tests and the fixture check


def pair(engine, i):
    """Fixed / moving landmark sets of registration i of config 4 (host arrays); `engine` = the icp_amd module."""
    return engine.synth_pair(SIDE, seed=BASE_SEED + i, rot_deg=1.0 + 0.125 * (i % 32), t=(25.0 - (i % 7), -10.0 + (i % 5), 15.0))


def ids_digest(ids):
    """(crc32, sum) of a correspondence-id array: the fixture's size-independent check of all 16384 ids."""
    ids = np.ascontiguousarray(ids, np.uint32)
    return np.array([zlib.crc32(ids.tobytes()), int(ids.astype(np.uint64).sum())], np.uint64)


def bits_digest(a):
    """(crc32, byte count) of the raw bytes of an array: bit-for-bit check of a large float / index array through a fixture."""
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a.tobytes()), a.nbytes], np.uint64)


def algorithmic_bytes(m, nr):
    """Per registration-iteration: read M, read the permuted fixed set once, read R, write {dist, id}, T."""
    return 72 * m + 32 * nr + 64


def algorithmic_flop(m, nr):
    """Per registration-iteration (SURVEY.md §8d / BASELINE.md §3): 18 flop per distance, Q x R + balanced list scans."""
    return 18.0 * m * (nr + m / nr) + 100.0 * m


# Invalid points of real captures: a Kinect frame's invalid pixels are points at the origin with their
# colour kept (reference src/kinect_frame_grabber.cpp:246-262), and getLMs picks them on purpose (kernels/icp_kernels.cl:49-50).
# name -> (pattern, fraction, keep_rgb).  The "_rgb0" cases zero the colour too: all holes are then one identical point and ONE
# representative's list holds every one of them (the degenerate list the one-shot search must not fall off a cliff on).
HOLES = {
    "scattered10": (0, 0.10, True), "blobs10": (1, 0.10, True), "blobs30": (1, 0.30, True),
    "scattered10_rgb0": (0, 0.10, False), "blobs10_rgb0": (1, 0.10, False), "blobs30_rgb0": (1, 0.30, False),
}


def holes_pair(engine, name, side=SIDE, seed=BASE_SEED):
    """The benchmark pair of `side` x `side` landmarks with the invalid points of case `name` in both frames
    (independent patterns: a moving camera sees other shadows)."""
    pattern, fraction, keep = HOLES[name]
    F, M = engine.synth_pair(side, seed=seed)
    F = engine.punch_holes(F, side, side, pattern, fraction, keep, seed=seed + 101)
    M = engine.punch_holes(M, side, side, pattern, fraction, keep, seed=seed + 202)
    return F, M


# The reference's second example pair, data/kg_pc8d_wall (data/README.md:11-16): "non-salient surface geometry ... highlights the benefit
# of utilizing the photometric information ... change the a parameter to a really small strictly positive number" to see what happens
# without colour.  Stand-in: a textured plane at 600 mm moved in its own plane (2 degrees about its normal, (8, -4) mm).
WALL_ROT_DEG, WALL_T, WALL_MAX_ITERATIONS, WALL_A_SMALL = 2.0, (8.0, -4.0, 0.0), 300, 1e-6


def wall_pair(engine, side=SIDE, seed=BASE_SEED):
    """(F, M, T_true) of the wall scene."""
    return engine.synth_pair_scene(side, engine.SCENE_WALL, seed=seed, rot_deg=WALL_ROT_DEG, t=WALL_T)


def rotation_error_deg(T, T_true):
    """Angle of the rotation that separates two transforms' quaternions [x y z w ...], degrees."""
    a, b = np.asarray(T[:4], np.float64), np.asarray(T_true[:4], np.float64)
    c = abs(float(np.dot(a, b))) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(np.degrees(2.0 * np.arccos(min(1.0, c))))
