"""TEST INFRASTRUCTURE: compile tests/native/*.hip for gfx950 into tests/native/libbuffer_direct.so (in-tree, so it travels to
the GPU box with the snapshot).  Holds the direct-form Cylindrical_Net kernel that the model tests use as a cross-check."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "convnet_direct.hip")
OUT = os.path.join(HERE, "libbuffer_direct.so")
ROOT = os.path.dirname(os.path.dirname(HERE))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC] + [os.path.join(ROOT, "buffer_amd", "csrc", f) for f in ("core.hip", "common.h")] + [os.path.join(ROOT, "include", "buffer_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fno-fast-math", "-Wall", "-Wno-unused-function", "-o", OUT, SRC])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
