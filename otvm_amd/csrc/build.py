"""Build libotvm_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object (include/otvm_hip.h) loaded through ctypes, so it is independent of the
torch wheel's ROCm version."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libotvm_hip.so")
LIB_PROBES = os.path.join(PKG, "libotvm_hip_probes.so")
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
    ("bottleneck128_f16x3.hip", []),
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


def _sha(*parts):
    import hashlib
    h = hashlib.sha256()
    for x in parts:
        h.update(x if isinstance(x, bytes) else str(x).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return b""


def source_hashes(probes=False):
    """{source: sha256 over (compiler flags, the source, every header it may include)} -- what an object is fresh AGAINST.
    Content hashes, not mtimes: a checkout, a copy to another box or a `touch` neither hides a stale object nor rebuilds a
    fresh one."""
    deps = [os.path.join(HERE, h) for h in sorted(os.listdir(HERE)) if h.endswith(".h")]          # common.h, head_math.h, ...
    deps.append(os.path.join(os.path.dirname(PKG), "include", "otvm_hip.h"))
    dep_hash = _sha(*[_read(d) for d in deps])
    flags = COMMON + (["-DOTVM_PROBES"] if probes else [])
    return {src: _sha(" ".join(flags + extra), _read(os.path.join(HERE, src)), dep_hash) for src, extra in SOURCES}


def tree_hash(probes=False):
    """One hash over all translation units: what the linked library is fresh against (stored beside it as <lib>.srchash)."""
    hs = source_hashes(probes)
    return _sha(*[hs[src] for src, _ in SOURCES])


def is_fresh(probes=False):
    lib = LIB_PROBES if probes else LIB
    return os.path.exists(lib) and _read(lib + ".srchash").decode().strip() == tree_hash(probes)


def build(force=False, verbose=False, probes=False):
    """Compile what is stale and link.  probes=True builds libotvm_hip_probes.so (-DOTVM_PROBES: the kernels' OTVM_* ablation
    switches readable from the environment, common.h::otvm_probe_int) beside the shipping library, objects in build_probes/."""
    lib, obj_dir = (LIB_PROBES, OBJ + "_probes") if probes else (LIB, OBJ)
    total = tree_hash(probes)
    if not force and is_fresh(probes):
        return lib                                   # the library on disk was linked from exactly these sources
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hs = source_hashes(probes)
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or _read(o + ".srchash").decode().strip() != hs[src]:
            cmd = ([hipcc] + COMMON + (["-DOTVM_PROBES"] if probes else []) + extra + (["-x", "hip"] if src.endswith(".cpp") else [])
                   + ["-c", s, "-o", o])
            jobs.append((cmd, o, hs[src]))
    # the translation units are independent: compile them side by side (conv_f16x3.hip alone takes minutes), longest first
    if jobs:
        jobs.sort(key=lambda c: -os.path.getsize(c[0][-3]))
        running, failed = [], []
        limit = max(1, min(len(jobs), (os.cpu_count() or 2)))
        while jobs or running:
            while jobs and len(running) < limit:
                cmd, o, h = jobs.pop(0)
                if verbose:
                    print(" ".join(cmd), flush=True)
                if os.path.exists(o + ".srchash"):
                    os.remove(o + ".srchash")
                running.append((cmd, o, h, subprocess.Popen(cmd)))
            cmd, o, h, pr = running.pop(0)
            if pr.wait() != 0:
                failed.append(cmd)
            else:
                with open(o + ".srchash", "w") as f:
                    f.write(h)
        if failed:
            raise subprocess.CalledProcessError(1, failed[0])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(lib + ".srchash"):
        os.remove(lib + ".srchash")
    subprocess.check_call(cmd)
    with open(lib + ".srchash", "w") as f:
        f.write(total)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))
