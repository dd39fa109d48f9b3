"""msm_zprize_amd -- MI355X-native Pippenger MSM behind the reference's curve API.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/msmz.h), the ctypes
binding (`_native`), the host mirror of `src/parallel.ts` (`parallel`) and the curve parameter
modules (`curves`).
"""
from . import curves  # noqa: F401
from .parallel import TwistedEdwards, Weierstrass, startThreads, stopThreads  # noqa: F401
