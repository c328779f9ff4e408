"""hr-viton_amd: MI355X (gfx950)-native HR-VITON hot path.

Import as ``hr_viton_amd`` (see the shim module at the repo root).  The package
holds only what the hot path needs: ``csrc/`` (HIP kernels + the C ABI declared
in ``include/hrviton_hip.h``), ``_lib`` (ctypes binding), ``ops`` (tensor-level
wrappers) and host-side mirrors of the reference's ``networks.py`` /
``network_generator.py`` class API.
"""
from . import _lib  # noqa: F401
from ._lib import HrvError, LIB_PATH  # noqa: F401

__all__ = ["HrvError", "LIB_PATH"]
