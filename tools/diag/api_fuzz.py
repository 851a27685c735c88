"""Diagnostic (not a test): tests/test_gpu_facade.py::test_random_api_sequences_against_the_oracle over many seeds and shapes."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import icp_amd as engine
from oracle import oracle
import test_gpu_facade as T
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0, n, bad = time.time(), 0, 0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
shapes = [(32, 16, False), (32, 16, True), (64, 1024, False), (48, 64, False), (16, 4, False), (64, 64, True), (128, 256, False)]
while time.time() - t0 < budget:
    side, nr, ref = shapes[n % len(shapes)]
    try:
        T.test_random_api_sequences_against_the_oracle(engine, oracle, side, nr, ref, seed + n)
    except AssertionError as e:
        bad += 1
        print("FAIL", side, nr, ref, seed + n, str(e)[:300], flush=True)
    n += 1
print("cases %d failures %d in %.0f s" % (n, bad, time.time() - t0))
