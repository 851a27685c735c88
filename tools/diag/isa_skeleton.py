"""Diagnostic (not a test; needs hipcc only): the memory / synchronisation skeleton of a kernel's ISA — every scalar / vector / LDS access, wait,
barrier and branch target with its instruction index — the view in which round 4 found the chained search waiting for its argument block in
front of its first loads (docs/HISTORY.md §5).  usage: python tools/diag/isa_skeleton.py [SOURCE [NAME-SUBSTRING [FIRST-N-INSTRUCTIONS]]]
    python tools/diag/isa_skeleton.py icp_amd/csrc/icp_kernels.hip k_searchILb1ELb1ELi2ELi16ELb0ELi1ELi1024ELb0ELb0ELb0E 700"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import kernel_isa

src = sys.argv[1] if len(sys.argv) > 1 else "icp_amd/csrc/icp_kernels.hip"
pat = sys.argv[2] if len(sys.argv) > 2 else "k_searchILb1ELb1ELi2ELi16ELb0ELi1ELi1024ELb0ELb0ELb0E"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
KEEP = ("s_load", "s_buffer_load", "global_", "flat_", "buffer_", "scratch_", "ds_", "s_barrier", "s_waitcnt", "s_endpgm", "s_sleep", "v_writelane", "v_mfma")
for name, ins in kernel_isa(src).items():
    if pat not in name:
        continue
    print("== %s: %d instructions" % (name, len(ins)))
    for i, t in enumerate(ins[:first]):
        if t.startswith(KEEP):
            print("%5d  %s" % (i, t))
