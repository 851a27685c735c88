"""Compiler-reported resources of every gfx950 kernel of the engine (hipcc -Rpass-analysis=kernel-resource-usage).

Needs only hipcc (cross-compiles without a GPU).  Used by tests/test_kernel_resources.py and as a diagnostic:
    python tools/diag/kernel_resources.py            # table of all kernels
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the flags of the Makefile (kept in step by test_kernel_resources.py::test_flags_match_makefile)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]
_KEYS = {"Function Name": "name", "TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
         "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "Dynamic Stack": "dynamic_stack"}


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return out.stdout.split("\n")[:len(names)] if out.returncode == 0 else names


def kernel_resources(source="icp_amd/csrc/icp_kernels.hip", extra_flags=()):
    """{demangled kernel name: {vgprs, sgprs, scratch, occupancy, lds, ...}} for one translation unit."""
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, source), "-o", os.devnull]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stderr[-4000:])
    recs, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            recs.append(cur)
        elif cur is not None and key in _KEYS:
            try:
                cur[_KEYS[key]] = int(val)
            except ValueError:
                cur[_KEYS[key]] = val
    names = demangle([r["name"] for r in recs])
    return {re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")).replace("void ", ""): r for n, r in zip(names, recs)}


def kernel_isa(source="icp_amd/csrc/icp_kernels.hip", extra_flags=()):
    """{mangled kernel name: [instruction lines]} of one translation unit's device code (hipcc -S --cuda-device-only)."""
    p = subprocess.run([HIPCC] + FLAGS + list(extra_flags) + ["-S", "--cuda-device-only", "-o", "-", os.path.join(ROOT, source)],
                       capture_output=True, text=True, cwd=ROOT)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stderr[-4000:])
    out, cur = {}, None
    for line in p.stdout.splitlines():
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m and ".type" not in line:
            cur = out.setdefault(m.group(1), [])
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None and t and not t.startswith((";", ".")) and not t.endswith(":"):
            cur.append(t.split(";")[0].strip())
    return out


if __name__ == "__main__":
    srcs = sys.argv[1:] or ["icp_amd/csrc/icp_kernels.hip", "icp_amd/csrc/icp_search_dense.hip", "icp_amd/csrc/icp_build.hip", "icp_amd/csrc/icp_reduce_scan.hip"]
    for s in srcs:
        for n, r in kernel_resources(s).items():
            print("%-60s vgpr %3d sgpr %3d scratch %4d occupancy %d lds %6d" % (n[:60], r.get("vgprs", -1), r.get("sgprs", -1),
                                                                               r.get("scratch", -1), r.get("occupancy", -1), r.get("lds", -1)))
