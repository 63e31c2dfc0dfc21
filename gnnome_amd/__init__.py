"""gnnome_amd - MI355X-native SymGatedGCN edge scoring behind GNNome's own module API.

    from gnnome_amd import models            # models.SymGatedGCNModel(...)(graph, x, e) -> logits[E,1]

Compute runs in libgnnome_hip.so (hand-written HIP for gfx950, C ABI in include/gnnome_hip.h);
there is no CPU or PyTorch fallback.
"""
from . import layers, models  # noqa: F401
from .models import SymGatedGCNModel  # noqa: F401

__all__ = ["models", "layers", "SymGatedGCNModel"]
