"""Drop-in for the reference's `model/pfnl.py`: `from model.pfnl import PFNL` (reference main.py:8)
resolves to the MI355X-native implementation in `pfnl_amd/model.py`."""
from pfnl_amd.model import PFNL, VSR  # noqa: F401

if __name__ == '__main__':
    model = PFNL()
    model.testvideos()
