"""Drop-in for the reference's ``src/network`` package (network.py, CleanUMamba.py, layers.py)."""
from .network import Net  # noqa: F401
from .CleanUMamba import CleanUMamba  # noqa: F401
