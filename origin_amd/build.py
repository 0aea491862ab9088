"""Build liborigin_hip.so (hipcc, gfx950 only) in-tree.

    python -m origin_amd.build          # or origin_amd.build.build()

hipcc cross-compiles without a GPU.  The built library sits next to this file so that it
travels with the source tree; it is git-ignored.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborigin_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(os.path.dirname(HERE), "include", "origin_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liborigin_hip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value",
             "-Wno-unused-result"] + os.environ.get("ORIGIN_HIPCC_FLAGS", "").split()
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    tmp = LIB + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
