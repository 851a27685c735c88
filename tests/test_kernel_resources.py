"""Compiler-side guards on the gfx950 kernels (need hipcc only, no GPU).

A latency-bound 9 us kernel must not pay scratch set-up per dispatch, and the one-block-per-CU variants must keep
fitting their launch bounds (1024 threads -> at most 128 VGPRs, i.e. occupancy >= 4 waves per SIMD).
"""
import os
import re

import pytest

from kernel_resources import FLAGS, ROOT, kernel_resources


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
