"""pfnl_amd — MI355X-native PFNL forward path: HIP kernels + C-ABI (csrc/, lib/), ctypes binding
(_capi), engine, and the drop-in model class (model.PFNL).  See DESIGN.md."""
from .spec import PFNLGeometry  # noqa: F401

__all__ = ["PFNLGeometry"]
