"""Drop-in package path of the reference (`from model.pfnl import PFNL`, reference main.py:8)."""
