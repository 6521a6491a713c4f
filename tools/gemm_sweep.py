"""Tile-configuration sweep for the three hot GEMM kinds (run on the GPU box): prints TFLOP/s per config."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "offlinerl-kit_amd")]
from offlinerlkit import _engine
lib = _engine.load_library()
NAMES = {0: "64x256 8w", 1: "64x64", 2: "16x64 tk64", 3: "64x16", 4: "128x128", 7: "64x256 4w", 11: "64x128"}
CFGS = [int(x) for x in os.environ.get("CFGS", ",".join(str(k) for k in sorted(NAMES))).split(",")]
def t(cfg, kind, M, N, K, nz, ks, reps=30):
    ms = C.c_float()
    rc = lib.orl_debug_gemm_time(cfg, kind, M, N, K, nz, ks, reps, C.byref(ms))
    if rc: return None
    return ms.value
for R in [int(x) for x in os.environ.get("RUNS", "1,8").split(",")]:
    nz = 2 * R
    print(f"== runs {R} (nz={nz}) ==")
    for kind, (M, N, K, kss) in {0: (7936, 256, 256, [1]), 1: (7936, 256, 256, [1]), 2: (256, 256, 7936, [8, 16, 32] if R == 1 else [4])}.items():
        fl = 2.0 * M * (N + (1 if kind == 2 else 0)) * K * nz
        for cfg in CFGS:
            for ks in kss:
                ms = t(cfg, kind, M, N, K, nz, ks)
                if ms is None: print("kind", kind, NAMES[cfg & 31], "ERR", _engine.last_error()); continue
                nm = NAMES[cfg & 31] + (" bf16x3" if cfg & 32 else "")
                print(f"kind {kind} {nm:24s} ks={ks:2d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
