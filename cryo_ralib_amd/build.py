"""Builds libralign_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libralign_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "ralign_engine.hip")]
HEADERS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h")) + \
          [os.path.join(ROOT, "include", "ralign.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP alignment engine cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
