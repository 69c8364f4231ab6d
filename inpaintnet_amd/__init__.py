"""inpaintnet_amd -- MI355X-native (gfx950) hot path of InpaintNet.

Python surface mirrors the reference's MeasureVAE / LatentRNN / Trainer classes;
all arithmetic runs in hand-written HIP kernels behind the C-ABI declared in
include/inpaintnet_hip.h (loaded with ctypes by inpaintnet_amd._lib).

Submodules are imported lazily so that data-only helpers
(inpaintnet_amd.synthetic) work where the HIP library is not built.
"""
__all__ = ["synthetic"]
