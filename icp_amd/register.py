"""`python -m icp_amd.register fixed.bin moving.bin [-o out.bin]` — what the reference's `ICPReg::registerPC`
does (src/ocl_icp_reg.cpp:165-210) without the GL window: landmarks (getLMs), buildRBC, ICP::run, full-cloud
transform of the moving cloud, and the same printout."""
import argparse
import math
import time

import numpy as np

from . import ICP, Memory, PowerMode, ReduceMode
from .io import load_pc8d, save_pc8d


def register_clouds(fixed, moving, device=0, a=2e2, c=1e-6, max_iterations=40, angle_threshold=0.001,
                    translation_threshold=0.01, reduce_mode=ReduceMode.FUSED):
    """Returns (T[8], k, latency_ms, transformed moving cloud)."""
    reg = ICP(device)
    reg.init(16384, 256, a, c, max_iterations, angle_threshold, translation_threshold)   # src/ocl_icp_reg.cpp:81-88
    reg.setPowerMode(PowerMode.SQUARED)
    reg.setReduceMode(reduce_mode)
    reg.write_cloud(Memory.F, fixed)
    reg.write_cloud(Memory.M, moving)
    reg.buildRBC()
    reg.sync()
    t0 = time.perf_counter()
    k = reg.run()
    ms = (time.perf_counter() - t0) * 1e3
    T = reg.read(Memory.T)
    out = reg.transform_cloud(moving)
    reg.close()
    return T, k, ms, out


def track(frames, device=0, a=2e2, c=1e-6, warm_start=False, **kw):
    """Frame-to-frame registration (README.md:4 of the reference): frame i is the fixed set of frame i+1.
    Yields (T_i, k_i) mapping frame i+1 onto frame i.  One handle; every frame is uploaded once, its landmarks are
    extracted on the device and stay there to serve as the next hop's fixed set (icp_track_next); warm_start: each hop
    starts from the previous hop's transform instead of the identity."""
    reg = ICP(device)
    reg.init(16384, 256, a, c, kw.get("max_iterations", 40), kw.get("angle_threshold", 0.001), kw.get("translation_threshold", 0.01))
    reg.setPowerMode(PowerMode.SQUARED)
    reg.setReduceMode(kw.get("reduce_mode", ReduceMode.FUSED))
    for f in frames:
        k = reg.track_next(f, warm_start)
        if k is not None:
            yield reg.read(Memory.T), k
    reg.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("fixed")
    ap.add_argument("moving")
    ap.add_argument("-o", "--output", help="write the transformed moving cloud (raw 640x480 float8)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("-a", "--alpha", type=float, default=2e2)
    args = ap.parse_args(argv)
    T, k, ms, out = register_clouds(load_pc8d(args.fixed), load_pc8d(args.moving), args.device, a=args.alpha)
    q, t, s = T[:4], T[4:7], T[7]
    sinth_2 = float(np.linalg.norm(q[:3]))
    angle = 180.0 / math.pi * 2 * math.atan2(sinth_2, float(q[3]))
    axis = q[:3] / sinth_2 if sinth_2 else np.zeros(3)
    print("\n================\n")                                     # src/ocl_icp_reg.cpp:199-206
    print("    Iterations            :    %d" % k)
    print("    Latency               :    %.3f ms" % ms)
    print("    Rotation angle        :    %g degrees" % angle)
    print("    Rotation axis         :    %s" % np.array2string(axis, precision=6))
    print("    Translation vector    :    %s" % np.array2string(t, precision=4))
    print("    Scale                 :    %g" % s)
    if args.output:
        save_pc8d(args.output, out)


if __name__ == "__main__":
    main()
