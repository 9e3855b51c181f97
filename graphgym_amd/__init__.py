"""graphgym_amd — MI355X-native message-passing engine behind GraphGym's layer API.

Hot path only: CSR neighbour aggregation (sum / mean / max), the ID-GNN
two-branch aggregation, and the layer modules that call them, behind the reference's
own plugin interface (graphgym.register.register_layer).
"""
from ._lib import EngineError, LIB_PATH, lib  # noqa: F401
from .graph import CSRGraph  # noqa: F401
from . import ops  # noqa: F401

__version__ = "0.1.0"
