"""Build libotvm_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object (include/otvm_hip.h) loaded through ctypes, so it is independent of the
torch wheel's ROCm version."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libotvm_hip.so")
OBJ = os.path.join(HERE, "build")

SOURCES = [
    ("error.cpp", []),
    ("conv_igemm.hip", []),
    ("conv_f16x3.hip", []),
    ("conv_f16x3_glds.hip", []),
    ("conv_f16x3_m16.hip", []),
    ("conv_f16x3_p1.hip", []),
    ("conv_f16x3_p1g.hip", []),
    ("conv_patch_f16x3.hip", []),
    ("conv_stem_f16x3.hip", []),
    ("conv_head16_f16x3.hip", []),
    ("bottleneck_f16x3.hip", []),
    ("groupnorm.hip", []),
    ("gram.hip", []),
    ("resample.hip", ["-ffp-contract=off"]),
    ("glue.hip", ["-ffp-contract=off"]),
    ("edt.hip", ["-ffp-contract=off"]),
    ("memory_read.hip", []),
    ("memory_read_f16x3.hip", []),
    ("metrics.hip", []),
    ("guard.hip", []),
    ("losses.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    deps = [os.path.join(HERE, h) for h in sorted(os.listdir(HERE)) if h.endswith(".h")]          # common.h, head_math.h, ...
    deps += [os.path.join(os.path.dirname(PKG), "include", "otvm_hip.h"), __file__]
    objs, relink, jobs = [], force, []
    for src, extra in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            jobs.append([hipcc] + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o])
            relink = True
    # the translation units are independent: compile them side by side (conv_f16x3.hip alone takes minutes), longest first
    if jobs:
        jobs.sort(key=lambda c: -os.path.getsize(c[-3]))
        running, failed = [], []
        limit = max(1, min(len(jobs), (os.cpu_count() or 2)))
        while jobs or running:
            while jobs and len(running) < limit:
                cmd = jobs.pop(0)
                if verbose:
                    print(" ".join(cmd), flush=True)
                running.append((cmd, subprocess.Popen(cmd)))
            cmd, pr = running.pop(0)
            if pr.wait() != 0:
                failed.append(cmd)
        if failed:
            raise subprocess.CalledProcessError(1, failed[0])
    if relink or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):   # (an object compiled by hand counts)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
