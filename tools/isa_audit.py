"""Resource audit of every kernel in the built library, from the code objects themselves (no recompilation):

    python tools/isa_audit.py > profiles/rNN_isa_audit.txt

For each object under otvm_amd/csrc/build the gfx950 code object is unbundled (llvm-objcopy --dump-section .hip_fatbin,
clang-offload-bundler --unbundle) and its kernel metadata read (llvm-readelf --notes): VGPRs, AGPRs, SGPRs, static LDS, scratch
bytes (private_segment_fixed_size), spilled VGPRs; scratch INSTRUCTIONS are counted in the disassembly.  Kernels with scratch are
listed again at the end."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def short(name):
    try:
        d = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), name], text=True).strip()
    except Exception:
        d = name
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", d)[:110]


def main():
    build = os.path.join(ROOT, "otvm_amd", "csrc", "build")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(f for f in os.listdir(build) if f.endswith(".o")):
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
            r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, os.path.join(build, o)],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat):
                continue                                      # (host-only object)
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + fat, "--output=" + co, "--unbundle"])
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
            scr = {}
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur and "scratch_" in line:
                    scr[cur] = scr.get(cur, 0) + 1
            for blk in notes.split("  - .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                    return m.group(1) if m else "?"
                name = f("name")
                rows.append((o, name, int(f("vgpr_count")), int(blk.split()[0]), int(f("sgpr_count")), int(f("group_segment_fixed_size")),
                             int(f("private_segment_fixed_size")), int(f("vgpr_spill_count")), scr.get(name, 0)))
            os.remove(fat)
    print("# kernel resources of otvm_amd/libotvm_hip.so (gfx950), read from the built code objects by tools/isa_audit.py")
    print("# object | kernel | VGPR | AGPR | SGPR | static LDS B | scratch B | spilled VGPRs | scratch instructions")
    for r in rows:
        print("%-22s %-112s %4d %4d %4d %7d %5d %4d %4d" % (r[0][:-2], short(r[1]), r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
    bad = [r for r in rows if r[6] or r[8]]
    print("\n# %d kernels, %d with scratch:" % (len(rows), len(bad)))
    for r in bad:
        print("#   %s: %d B, %d instructions" % (short(r[1]), r[6], r[8]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
