"""Compiler-side guards on the gfx950 kernels (need hipcc only, no GPU).

A latency-bound 9 us kernel must not pay scratch set-up per dispatch, and the one-block-per-CU variants must keep
fitting their launch bounds (1024 threads -> at most 128 VGPRs, i.e. occupancy >= 4 waves per SIMD).
"""
import os
import re

import pytest

from kernel_resources import FLAGS, ROOT, kernel_isa, kernel_resources


@pytest.fixture(scope="module")
def res():
    r = dict(kernel_resources("icp_amd/csrc/icp_kernels.hip"))
    r.update(kernel_resources("icp_amd/csrc/icp_search_dense.hip"))   # the dense variants of the search
    r.update(kernel_resources("icp_amd/csrc/icp_build.hip"))        # state, getLMs, transforms, the RBC construction, the rotation solver
    return r


def test_flags_match_makefile():
    mk = open(os.path.join(ROOT, "Makefile")).read()
    line = re.search(r"^HIPFLAGS\s*\?=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950")
    for f in FLAGS:
        assert f in line.split(), (f, line)


def test_every_per_iteration_kernel_has_zero_scratch(res):
    hot = [n for n in res if n.startswith(("k_search", "k_finalize", "k_chain_end", "k_means", "k_sij", "k_sum_w", "k_gmean"))]
    assert len(hot) >= 12, hot
    for n in hot:
        assert res[n]["scratch"] == 0, (n, res[n])
        assert res[n].get("dynamic_stack") in (None, "False"), (n, res[n])


def test_no_kernel_spills_or_uses_scratch(res):
    for n, r in res.items():
        assert r["scratch"] == 0, (n, r)


def test_launch_bounds_hold(res):
    """1024-thread blocks (16 waves on 4 SIMDs) need occupancy >= 4; the dense 512-thread variants are built for two
    blocks per CU (MINW = 4)."""
    for n, r in res.items():
        if n.startswith("k_search<") and ", 2, 16" in n:
            assert r["occupancy"] >= 4 and r["vgprs"] <= 128, (n, r)
        if n.startswith("k_search<") and ", 4, 8" in n:
            assert r["occupancy"] >= 4, (n, r)
        if n.startswith("k_search"):
            assert r["lds"] <= 80 * 1024, (n, r)           # two blocks per CU must fit the 160 KiB


def test_chained_search_issues_its_first_loads_before_it_waits_for_its_arguments():
    """The chained k_search (one launch per iteration at the reference's size) gets the 14 dwords its first loads need preloaded in SGPRs;
    the rest of its arguments arrives by scalar loads, and the only wait for those is `s_waitcnt lgkmcnt (0)` — for all of them.  Such a
    wait (or a spill of freshly loaded arguments: v_writelane) in front of the state load and the eight block-moment loads starts every
    wave's first memory round trip an argument fetch late: a scalar-cache miss in a graph replay, a trip to memory in a plain launch
    (docs/HISTORY.md §5, "Where the kernel arguments are waited for": 0.3 - 0.5 us of 9).  Both instantiations, power-method rotation."""
    isa = kernel_isa("icp_amd/csrc/icp_kernels.hip")
    names = [n for n in isa if n.startswith("_Z8k_searchILb1ELb1ELi2ELi16ELb0ELi1ELi1024ELb0ELb0E")]
    assert len(names) == 2, names                                     # HOSTRUN = false (fixed-length graphs) and true (host-driven runs)
    for n in names:
        ins = isa[n]
        entry = next(i for i, t in enumerate(ins) if t.startswith("s_branch"))      # (in front of it: the loads of a loader without preload support)
        loads = [i for i, t in enumerate(ins) if i > entry and t.startswith("global_load")]
        assert len(loads) >= 9
        head = ins[entry:loads[8] + 1]                                 # up to the state load + the eight moment loads
        bad = [t for t in head if (t.startswith("s_waitcnt") and "lgkmcnt" in t) or t.startswith("v_writelane")]
        assert not bad, (n, bad)
