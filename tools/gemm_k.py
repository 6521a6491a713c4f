import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "offlinerl-kit_amd")]
from offlinerlkit import _engine
lib = _engine.load_library()
def t(cfg, kind, M, N, K, nz, ks, reps=20):
    ms = C.c_float()
    rc = lib.orl_debug_gemm_time(cfg, kind, M, N, K, nz, ks, reps, C.byref(ms))
    return None if rc else ms.value
for cfg in (0, 7, 9):
    for K in (64, 256, 1024, 4096):
        for M in (7936, 63488):
            ms = t(cfg, 0, M, 256, K, 2, 1)
            print(f"cfg {cfg} fwd M={M} K={K}: {ms*1e3:8.1f} us {2.0*M*256*K*2/ms/1e9:7.1f} TF", flush=True)
