"""pecos_amd -- MI355X-native XR-Linear batch inference (drop-in for the predict path of amzn/pecos).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI), :mod:`pecos_amd.core`
(the ctypes binding that mirrors ``pecos.core.clib``), :mod:`pecos_amd.xlinear` (the
``XLinearModel`` surface) and :mod:`pecos_amd.distributed` (query sharding + RCCL all-gather).
"""
from .core import clib  # noqa: F401
from .xlinear import HierarchicalMLModel, MLModel, XLinearModel  # noqa: F401

__all__ = ["clib", "XLinearModel", "HierarchicalMLModel", "MLModel"]
