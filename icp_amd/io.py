"""Point-cloud files of the reference's demos: raw little-endian 640x480 float8 `[x y z 1 r g b 1]`
(`data/kg_pc8d_{1,2}.bin`, 9 830 400 bytes; reader examples/registration.cpp:285-337, writer
src/kinect_frame_grabber.cpp:252-272)."""
import numpy as np

VGA_POINTS = 640 * 480


def load_pc8d(path):
    a = np.fromfile(path, dtype="<f4")
    if a.size != VGA_POINTS * 8:
        raise ValueError("%s: expected %d bytes (640x480 float8), got %d" % (path, VGA_POINTS * 32, a.size * 4))
    return a.reshape(VGA_POINTS, 8)


def save_pc8d(path, cloud):
    cloud = np.ascontiguousarray(cloud, dtype="<f4")
    if cloud.size != VGA_POINTS * 8:
        raise ValueError("expected a 640x480 float8 cloud")
    cloud.tofile(path)
