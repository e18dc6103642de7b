import numpy as np
import torch


def h16(a):
    """Round to fp16 and back (the HIP path stores activations / weights in fp16)."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def assert_close(got, ref, tol, what=""):
    assert np.isfinite(np.asarray(got)).all(), f"{what}: non-finite output"
    e = rel_err(got, ref)
    assert e < tol, f"{what}: max-abs error / max|ref| = {e:.3e} >= {tol:.1e}"
    return e


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))
