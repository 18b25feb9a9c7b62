"""Build libraffthip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libraffthip.so")
SOURCES = ["rafft_api.hip", "rafft_kernels.hip", "rafft_expand_small.hip", "rafft_kin.hip", "rafft_kernels.h", "rafft_device.h", "rafft_params.h", "rafft_config.h", "../../params/turner2004_tables.h"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "rafft_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
           "-Wno-unused-function", "-Wno-missing-braces", os.path.join(SRC, "rafft_api.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=SRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
