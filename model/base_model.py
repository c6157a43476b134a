"""Drop-in for the reference's `model/base_model.py` (class VSR: attributes, save/load)."""
from pfnl_amd.model import VSR  # noqa: F401
