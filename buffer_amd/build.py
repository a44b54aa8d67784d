"""Compile buffer_amd/csrc into buffer_amd/libbuffer_hip.so for gfx950 (in-tree, so the built
library travels with the repository snapshot to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "buffer_hip.hip")
OUT = os.path.join(HERE, "libbuffer_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for root in (os.path.join(HERE, "csrc"), os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    return False


def build(force=False, verbose=False, out=None):
    if out is None:
        out = OUT
    if out == OUT and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           # MFMA accumulators stay VGPRs: k_cyl_net_wg parks held outputs in AGPRs by hand (no scratch); with AGPR-form
           # accumulators the compiler rotates them through v_accvgpr copies inside the pass-0 loops (+3 % on that kernel)
           "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           # no packed-fp32 vector instructions: beside another wavefront's f16 MFMA they return wrong values in lanes 0..15 now and
           # then (round 5: tools/race_probe3.py, profiles/r05_packed_fp32_hazard.txt); nothing got slower without them
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
           "-o", out, SRC] + os.environ.get("BUF_EXTRA_HIPCC_FLAGS", "").split()
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    # (the host half of the compilation does not know the device feature and says so once per file: not a diagnostic of ours)
    err = "\n".join(ln for ln in r.stderr.splitlines() if "is not a recognized feature for this target" not in ln)
    if err.strip():
        print(err, file=sys.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    if "+packed-fp32-ops" not in os.environ.get("BUF_EXTRA_HIPCC_FLAGS", ""):     # (tools/pk_bisect.sh builds packed variants on purpose)
        verify_no_packed_fp32(out)
    return out


def verify_no_packed_fp32(so):
    """The warning filter above would also hide a toolchain that IGNORES -packed-fp32-ops on the device side: the built code
    object itself is checked.  Any v_pk_{mul,add,fma}_f32 in it fails the build (profiles/r06_pk_hazard.txt: such kernels return wrong values
    beside the library's f16-MFMA kernels).  Skipped only where the image has no llvm-objdump."""
    import glob
    import re
    import tempfile
    objdump = os.path.join(os.path.dirname(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), "..", "lib", "llvm", "bin", "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        print("buffer_amd.build: no llvm-objdump, the packed-fp32 check of the code object was skipped", file=sys.stderr)
        return
    with tempfile.TemporaryDirectory() as tmp:
        link = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(so), link)
        subprocess.run([objdump, "--offloading", link], capture_output=True, cwd=tmp, check=True)
        cos = [f for f in glob.glob(os.path.join(tmp, "*")) if "gfx950" in os.path.basename(f)]
        if not cos:
            raise RuntimeError(f"{so}: no gfx950 code object found")
        asm = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True, check=True).stdout
    hits = re.findall(r"v_pk_(?:mul|add|fma)_f32", asm)
    if hits:
        raise RuntimeError(f"{so}: {len(hits)} packed-fp32 instructions in the gfx950 code object: the toolchain ignored "
                           "-target-feature -packed-fp32-ops (see profiles/r06_pk_hazard.txt)")


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(OUT)
