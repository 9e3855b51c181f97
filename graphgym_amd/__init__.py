"""graphgym_amd — MI355X-native message-passing engine behind GraphGym's layer API.

Hot path only: CSR neighbour aggregation (sum / mean / max), the ID-GNN
two-branch aggregation, and the layer modules that call them, behind the reference's
own plugin interface (graphgym.register.register_layer).
"""
from . import hostcpu
from ._lib import EngineError, LIB_PATH, lib  # noqa: F401
from .graph import CSRGraph  # noqa: F401
from . import ops  # noqa: F401

__version__ = "0.1.0"

# torch sizes its CPU thread pool by the machine, not by the container's CPU quota; over the quota the kernel freezes the
# whole process for tens of milliseconds at a time (hostcpu.py).  MP_KEEP_THREADS=1 / OMP_NUM_THREADS leave it alone.
hostcpu.fit_torch_threads()
