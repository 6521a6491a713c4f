import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "offlinerl-kit_amd")]
from offlinerlkit import _engine
lib = _engine.load_library()
NAMES = {0: "64x256 8w", 1: "64x64", 4: "128x128", 7: "64x256 4w", 11: "64x128"}
def t(cfg, kind, M, N, K, nz, ks, reps=20):
    ms = C.c_float(); rc = lib.orl_debug_gemm_time(cfg, kind, M, N, K, nz, ks, reps, C.byref(ms)); return None if rc else ms.value
for nz in (2, 32):
    for K in (24,):
        for cfg in (0, 1, 4, 7, 11):
            for pb in (0, 32):
                ms = t(cfg | pb, 0, 7936, 256, K, nz, 1)
                print(f"nz={nz} K={K} {NAMES[cfg]:12s} {'bf16x3' if pb else 'fp32  '} {ms*1e3:8.1f} us  write {7936*256*4*nz/ms/1e9:6.2f} TB/s", flush=True)
