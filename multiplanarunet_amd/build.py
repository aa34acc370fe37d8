"""
Builds libmpunet_hip.so (gfx950 only) in-tree with hipcc. No CMake, no JIT
cache: the .so lands in multiplanarunet_amd/lib/ so that it travels with the
repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libmpunet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable"]
# geometry.hip restates fp64 NumPy arithmetic op by op: no FMA contraction.
PER_FILE = {"geometry.hip": ["-ffp-contract=off"], "augment.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=True, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "mpunet_hip.h"))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if force or _newer(obj, [path] + headers):
            cmd = [HIPCC] + COMMON + PER_FILE.get(src, []) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
