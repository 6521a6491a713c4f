"""Builds the HIP update engine (liborlengine.so) in-tree for gfx950 with hipcc.
The shared library is a plain C-ABI object (include/orl_engine.h); no torch headers involved.

The sources are several translation units (host side + small kernels, the tiled GEMM instantiations in groups, one unit per
weight-stationary kernel) compiled in parallel and linked once; an object is rebuilt only when the content hash of the unit and
of the headers it includes changed (file times do not survive every copy)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "liborlengine.so")
STAMP = LIB + ".srchash"
ABI = os.path.join("..", "..", "include", "orl_engine.h")
GEMM_H = ["gemm.h", "gemm_kernel.h"]
WS_H = ["gemm.h", "ws_gemm.h", "ws_device.h"]
# translation unit -> headers it depends on
UNITS = {
    "engine.hip": ["engine.h", "gemm.h", "ws_gemm.h", "small_fwd.h", "small_bwd.h", "sample.h", "scalars.h", "kernels.h", "algo_cql.inc", "algo_iql.inc", "algo_td3bc.inc", "algo_edac.inc", "algo_sac.inc", "algo_mcq.inc", ABI],
    "gemm_inst_fwd.hip": GEMM_H,
    "gemm_inst_plain.hip": GEMM_H,
    "gemm_inst_rank1.hip": GEMM_H,
    "gemm_inst_wgrad.hip": GEMM_H,
    "gemm_inst_tune.hip": GEMM_H,
    "ws_fwd.hip": WS_H,
    "ws_dgrad.hip": WS_H,
    "ws_fwd3.hip": WS_H,
    "ws_dgrad3.hip": WS_H,
    "ws_wgrad3p.hip": WS_H,
    "ws_wgrad.hip": WS_H,
    "small_fwd.hip": ["small_fwd.h", "sample.h", "gemm.h"],
    "small_bwd.hip": ["small_bwd.h", "scalars.h", "gemm.h"],
}
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _hash(files):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def unit_hash(unit):
    return _hash([unit] + UNITS[unit])


def source_hash():
    return _hash(sorted(set(sum(([u] + d for u, d in UNITS.items()), []))))


def _obj(unit):
    return os.path.join(OBJ, unit.replace(".hip", ".o"))


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def _compile(unit, hipcc, verbose, objdir=None, defines=()):
    obj = _obj(unit) if objdir is None else os.path.join(objdir, unit.replace(".hip", ".o"))
    want = unit_hash(unit) + "|" + " ".join(defines)
    stamp = obj + ".srchash"
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj
    cmd = [hipcc] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, unit), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(want + "\n")
    return obj


def build(force=False, verbose=True, jobs=None):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for u in UNITS:
            if os.path.exists(_obj(u) + ".srchash"):
                os.remove(_obj(u) + ".srchash")
    h = source_hash()                      # hash what is about to be compiled
    jobs = jobs or min(len(UNITS), max(1, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda u: _compile(u, hipcc, verbose), UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(h + "\n")
    return LIB


def build_variant(name, defines, verbose=False):
    """Experiment builds: liborlengine_<name>.so compiled with extra -D defines (objects under build/<name>/); select it at run
    time with ORL_ENGINE_LIB=<path> (offlinerlkit/_engine.py) to A/B two kernels inside ONE gpurun call -- boxes differ by ~10 %."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(OBJ, name)
    os.makedirs(objdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(UNITS), max(1, (os.cpu_count() or 2) - 1))) as ex:
        objs = list(ex.map(lambda u: _compile(u, hipcc, verbose, objdir, tuple(defines)), UNITS))
    lib = os.path.join(HERE, f"liborlengine_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force="--force" in sys.argv)
