"""MI355X-native WSI-patch x omics fusion path (MCAT / NaCAGaT hot path).

Host side is Python on PyTorch-ROCm; every kernel is hand-written HIP for gfx950
behind the C-ABI declared in include/mpo_hip.h (libmpo_hip.so, loaded by `_lib`).
Importing the package is cheap and GPU-free; the first call of any op loads the
library and raises if it is missing -- there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
