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


import contextlib


@contextlib.contextmanager
def fp16_storage(*modules):
    """fp16-storage emulation of an fp16 run of the CPU oracle: every leaf module's output (convolution, linear, norm) is rounded to fp16
    while the arithmetic inside it stays fp32 - what an fp16 GPU run does (fp16 tensors, fp32 accumulators).  torch-CPU's own fp16 kernels
    are far too slow for the full architecture; on the tiny configuration the emulation is checked against a real ``.half()`` run
    (tests/test_stages_gpu.py).  north_star's 1e-3 is stated against an fp16 reference run: HIP vs THIS is the quantity it bounds."""
    hooks = []

    def rnd(_m, _inp, out):
        if torch.is_tensor(out) and out.is_floating_point():
            return out.half().float()
        return None

    for mod in modules:
        for m in mod.modules():
            if not list(m.children()):
                hooks.append(m.register_forward_hook(rnd))
    try:
        yield
    finally:
        for h in hooks:
            h.remove()


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


def e4m3_rne(y):
    """float -> nearest OCP e4m3 value (round-to-nearest-even, saturating at +-448), returned as float32."""
    y = np.asarray(y, dtype=np.float64)
    a = np.minimum(np.abs(y), 448.0)
    ex = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    ex = np.maximum(ex, -6.0)                                   # subnormals share the quantum of the smallest normal binade
    quantum = 2.0 ** (ex - 3)
    q = np.rint(a / quantum) * quantum                          # np.rint = half-to-even
    return np.copysign(np.minimum(q, 448.0), y).astype(np.float32)          # -0.0 keeps its sign bit, as the hardware conversion does


def e4m3_encode(v):
    """e4m3 value (as produced by e4m3_rne) -> byte."""
    v = np.asarray(v, dtype=np.float64)
    a = np.abs(v)
    ex = np.floor(np.log2(np.maximum(a, 2.0 ** -20))).astype(np.int64)
    normal = a >= 2.0 ** -6
    e = np.where(normal, ex + 7, 0)
    m = np.where(normal, np.rint((a / 2.0 ** np.where(normal, ex, 0) - 1.0) * 8), np.rint(a / 2.0 ** -9)).astype(np.int64)
    return ((np.signbit(v).astype(np.int64) << 7) | (e << 3) | m).astype(np.uint8)


def mx8_quantise(x):
    """OCP-MX fp8 quantisation as kernels/mx8.hip does it: x [M,K] -> (dequantised float32 [M,K], bytes [M,K], biased e8m0 [M,K/32])."""
    x = np.asarray(x, dtype=np.float32)
    M, K = x.shape
    xb = x.reshape(M, K // 32, 32).astype(np.float64)
    amax = np.abs(xb).max(-1)
    ex = np.where(amax > 0, np.floor(np.log2(np.maximum(amax, 1e-300))) - 8, 0.0)
    ex = np.clip(ex, -127, 127)
    q = e4m3_rne(xb * 2.0 ** (-ex[..., None]))
    deq = (q.astype(np.float64) * 2.0 ** ex[..., None]).reshape(M, K).astype(np.float32)
    return deq, e4m3_encode(q).reshape(M, K), (ex + 127).astype(np.uint8)
