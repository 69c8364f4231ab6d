"""inpaintnet_amd -- MI355X-native (gfx950) hot path of InpaintNet.

Python surface mirrors the reference's MeasureVAE / LatentRNN / Trainer classes;
all arithmetic runs in hand-written HIP kernels behind the C-ABI declared in
include/inpaintnet_hip.h (loaded with ctypes by inpaintnet_amd._lib).

Submodules are imported lazily so that data-only helpers
(inpaintnet_amd.synthetic) work where the HIP library is not built.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every launch of the step reads its kernarg segment
# once per wave, and the persistent chain kernels re-read parts of it per time step; with the default placement that read
# crosses PCIe.  Measured on the B=256 MeasureVAE step: 4.25 -> 4.20 ms (profiles/r03_i_kernarg_ab.txt; 0.5 ms in round 2,
# when a step had 124 launches).  The HIP runtime reads the
# variable when it initialises, i.e. at the first HIP call of the process -- importing this package before touching the
# GPU is enough; a value the user has set is respected.  
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__all__ = ["synthetic"]
