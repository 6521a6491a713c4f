"""GPU: unit tests of the run-batched MFMA GEMM template (csrc/gemm.h) through the C ABI
(orl_debug_gemm) against numpy, for every tile configuration and prologue/epilogue mode.
fp32 MFMA is an exact fp32 fma chain, so the tolerance is summation-order only (1e-5 rel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL = 0, 1, 2, 3
SHAPES = [(64, 256, 32), (70, 40, 23), (256, 256, 256), (130, 17, 100), (16, 1, 256), (1, 33, 77), (300, 257, 129), (128, 64, 96), (200, 24, 64), (513, 256, 260)]


def _close(got, ref, tol=2e-5):
    scale = max(np.abs(ref).max(), 1e-6)
    assert np.abs(got - ref).max() / scale < tol, np.abs(got - ref).max() / scale


@pytest.mark.parametrize("cfg", [CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL, CFG_BIG | 16, CFG_MID | 16, CFG_SMALL | 16, CFG_TALL | 16])
@pytest.mark.parametrize("shape", SHAPES)
def test_forward_bias_relu(cfg, shape):
    from offlinerlkit._engine import debug_gemm
    M, N, K = shape
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)   # asymmetric operands catch transposed outputs
    b = rng.standard_normal(N).astype(np.float32)
    got = debug_gemm(cfg, 0, A, W, b, M=M, N=N, K=K).reshape(M, N)
    ref = np.maximum(A.astype(np.float64) @ W.T.astype(np.float64) + b, 0)
    _close(got, ref)


@pytest.mark.parametrize("cfg", [CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL, CFG_BIG | 16, CFG_MID | 16, CFG_SMALL | 16, CFG_TALL | 16])
@pytest.mark.parametrize("shape", SHAPES)
def test_dgrad_masked_and_rank1(cfg, shape):
    from offlinerlkit._engine import debug_gemm
    M, N, K = shape
    rng = np.random.RandomState(M + N * 5 + K * 11)
    dY = rng.standard_normal((M, K)).astype(np.float32)
    Wm = rng.standard_normal((K, N)).astype(np.float32)
    H = rng.standard_normal((M, N)).astype(np.float32)
    got = debug_gemm(cfg, 1, dY, Wm, H, M=M, N=N, K=K).reshape(M, N)
    ref = (dY.astype(np.float64) @ Wm.astype(np.float64)) * (H > 0)
    _close(got, ref)
    # rank-1 virtual operand: dz = (Hk > 0) * dq[m] * w[k]
    Hk = rng.standard_normal((M, K)).astype(np.float32)
    dq = rng.standard_normal(M).astype(np.float32)
    w = rng.standard_normal(K).astype(np.float32)
    got = debug_gemm(cfg, 3, Hk, Wm, dq, w, M=M, N=N, K=K).reshape(M, N)
    dz = (Hk > 0) * np.outer(dq, w)
    _close(got, dz.astype(np.float64) @ Wm.astype(np.float64))


@pytest.mark.parametrize("cfg", [CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL, CFG_BIG | 16, CFG_MID | 16, CFG_SMALL | 16, CFG_TALL | 16])
@pytest.mark.parametrize("shape", [(64, 64, 512), (256, 23, 1000), (1, 256, 300), (12, 256, 256), (40, 33, 77), (256, 256, 1024), (32, 24, 640), (64, 256, 100)])
@pytest.mark.parametrize("ksplit", [1, 3, 8])
def test_wgrad_with_bias_column_and_splitk(cfg, shape, ksplit):
    from offlinerlkit._engine import debug_gemm
    M, N, K = shape    # dW[M x N] = dY[K x M]^T X[K x N]; db[M] = column sums of dY
    rng = np.random.RandomState(M * 13 + N + K * 2 + ksplit)
    dY = rng.standard_normal((K, M)).astype(np.float32)
    X = rng.standard_normal((K, N)).astype(np.float32)
    got = debug_gemm(cfg, 2, dY, X, ksplit=ksplit, M=M, N=N, K=K)
    _close(got[:M * N].reshape(M, N), dY.T.astype(np.float64) @ X.astype(np.float64))
    _close(got[M * N:], dY.astype(np.float64).sum(0))
    # rank-1 virtual dY = (H > 0) * dq[k] * w[m]
    H = rng.standard_normal((K, M)).astype(np.float32)
    dq = rng.standard_normal(K).astype(np.float32)
    w = rng.standard_normal(M).astype(np.float32)
    got = debug_gemm(cfg, 4, H, X, dq, w, ksplit=ksplit, M=M, N=N, K=K)
    dz = ((H > 0) * np.outer(dq, w)).astype(np.float64)
    _close(got[:M * N].reshape(M, N), dz.T @ X.astype(np.float64))
    _close(got[M * N:], dz.sum(0))


@pytest.mark.parametrize("precision", [1, 2])
@pytest.mark.parametrize("cfg", [CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL, CFG_MID | 16])
@pytest.mark.parametrize("shape", [(64, 256, 32), (70, 40, 23), (256, 256, 256), (300, 257, 129), (513, 256, 260)])
def test_split_bf16_forward_dgrad(cfg, shape, precision):
    """precision=1: hi/lo split, 3 MFMA products, fp32 accumulate (the bar dates from the bf16 planes: ~16 mantissa bits per operand);
    precision=2: three fp16 planes, 6 products -- held to the exact-fp32 kernel's summation-order tolerance."""
    from offlinerlkit._engine import debug_gemm
    tol = 2e-4 if precision == 1 else 2e-5
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    got = debug_gemm(cfg, 0, A, W, b, M=M, N=N, K=K, precision=precision).reshape(M, N)
    _close(got, np.maximum(A.astype(np.float64) @ W.T.astype(np.float64) + b, 0), tol=tol)
    Wm = rng.standard_normal((K, N)).astype(np.float32)
    H = rng.standard_normal((M, N)).astype(np.float32)
    got = debug_gemm(cfg, 1, A, Wm, H, M=M, N=N, K=K, precision=precision).reshape(M, N)
    _close(got, (A.astype(np.float64) @ Wm.astype(np.float64)) * (H > 0), tol=tol)


@pytest.mark.parametrize("precision", [1, 2])
@pytest.mark.parametrize("cfg", [CFG_BIG, CFG_MID, CFG_SMALL, CFG_TALL])
@pytest.mark.parametrize("shape", [(64, 64, 512), (256, 23, 1000), (1, 256, 300), (256, 256, 1024)])
def test_split_bf16_wgrad(cfg, shape, precision):
    from offlinerlkit._engine import debug_gemm
    tol = 2e-4 if precision == 1 else 2e-5
    M, N, K = shape
    rng = np.random.RandomState(M * 13 + N + K * 2)
    dY = rng.standard_normal((K, M)).astype(np.float32)
    X = rng.standard_normal((K, N)).astype(np.float32)
    got = debug_gemm(cfg, 2, dY, X, ksplit=4, M=M, N=N, K=K, precision=precision)
    _close(got[:M * N].reshape(M, N), dY.T.astype(np.float64) @ X.astype(np.float64), tol=tol)
    _close(got[M * N:], dY.astype(np.float64).sum(0), tol=tol)
    H = rng.standard_normal((K, M)).astype(np.float32)
    dq = rng.standard_normal(K).astype(np.float32)
    w = rng.standard_normal(M).astype(np.float32)
    got = debug_gemm(cfg, 4, H, X, dq, w, ksplit=4, M=M, N=N, K=K, precision=precision)
    dz = ((H > 0) * np.outer(dq, w)).astype(np.float64)
    _close(got[:M * N].reshape(M, N), dz.T @ X.astype(np.float64), tol=tol)
