"""Builds the HIP update engine (liborlengine.so) in-tree for gfx950 with hipcc.
The shared library is a plain C-ABI object (include/orl_engine.h); no torch headers involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborlengine.so")
SOURCES = ["engine.hip"]
HEADERS = ["engine.h", "gemm.h", "kernels.h", "ws_gemm.h", "mlp_fused.h", "algo_cql.inc", "algo_iql.inc", "algo_td3bc.inc", "algo_edac.inc",
           os.path.join("..", "..", "include", "orl_engine.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
