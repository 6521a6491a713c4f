"""Tile sweep for the 256-row phases at many runs: wgrad (256x256 outputs, K = 256 rows), dgrad / forward (256 rows x 256 x 256)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "offlinerl-kit_amd")]
from offlinerlkit import _engine
lib = _engine.load_library()
NAMES = {0: "64x256 8w", 1: "64x64", 2: "16x64 tk64", 4: "128x128", 7: "64x256 4w", 11: "64x128"}
def t(cfg, kind, M, N, K, nz, ks, reps=30):
    ms = C.c_float()
    rc = lib.orl_debug_gemm_time(cfg, kind, M, N, K, nz, ks, reps, C.byref(ms))
    return None if rc else ms.value
for nz in (128, 256):
    for kind, (M, N, K) in {2: (256, 256, 256), 1: (256, 256, 256), 0: (256, 256, 256)}.items():
        for cfg in NAMES:
            for ks in ((1, 2) if kind == 2 else (1,)):
                ms = t(cfg | 32, kind, M, N, K, nz, ks)
                if ms is None: print("ERR", _engine.last_error()); continue
                print(f"nz {nz} kind {kind} {NAMES[cfg]:12s} ks={ks}  {ms*1e3:8.1f} us", flush=True)
