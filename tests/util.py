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
    if what:
        report(what, e, tol=tol, kind="max|err|/max|ref|")
    assert e < tol, f"{what}: max-abs error / max|ref| = {e:.3e} >= {tol:.1e}"
    return e


def assert_abs(got, ref, tol, what):
    """Absolute bound for quantities that live in a fixed range (decoded frames in [0,1])."""
    assert np.isfinite(np.asarray(got)).all(), f"{what}: non-finite output"
    e = float(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)).max())
    report(what, e, tol=tol, kind="max|err|")
    assert e < tol, f"{what}: max-abs error = {e:.3e} >= {tol:.1e}"
    return e


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def report(name, value, **extra):
    """Print a measured parity number and append it to gpurun_out/parity_measured.jsonl (copied to profiles/ per round)
    so that every tolerance in tests/ can be read next to the value that was actually measured on the GPU."""
    import json, os
    rec = {"name": name, "value": float(value), **extra}
    print("PARITY", json.dumps(rec))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return float(value)
