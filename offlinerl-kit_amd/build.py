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


STAMP = LIB + ".srchash"      # content hash of the sources the library was built from (file times do not survive every copy)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    h = source_hash()                      # hash what is about to be compiled
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(h + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
