"""MI355X-native DepthCrafter / StableNormal path behind UniGeo's `model/` plugin surface.

Importing the package configures the HIP runtime for this process BEFORE its first HIP call:

* ``HIP_FORCE_DEV_KERNARG=1`` - kernel-argument buffers in device memory.  Every launch of this engine starts with scalar loads of its argument block
  (GemmP is ~400 bytes); with the runtime's default placement (host memory, read over PCIe) that first dependent round trip costs each of the ~19 000
  launches of a clip 1 - 2 us of its ramp: measured on one MI355X 916.4 -> 886.6 ms per 25 x 384 x 512 clip (-3.2 %) and 95.1 -> 83.8 ms per StableNormal
  image (-12 %), `profiles/r06_hip_force_dev_kernarg.txt`.  A documented ROCm runtime switch; `setdefault`, so an explicit setting of the caller wins.

The runtime reads its switches once, at its first API call.  A process that has already used HIP through another library (e.g. `torch.cuda.*` before this import)
keeps whatever it had; `unigeo_amd.runtime_switches()` reports what this import found / set.  Merely importing torch first (as the reference's eval.py does) does not
initialise HIP, so the reference harness gets the setting.
"""
import os as _os

_FOUND = {k: _os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG",)}
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def runtime_switches():
    """{switch: (value found at import, value in effect for a runtime initialised after the import)}"""
    return {k: (v, _os.environ.get(k)) for k, v in _FOUND.items()}
