"""MI355X-native SocioReasoner inference hot path (host side).

The compute lives in ``libsocior.so`` (hand-written HIP for gfx950, C ABI in ``include/socior.h``); this package is
the thin Python host layer: ctypes binding, engine wrapper, the reference's host-side integer logic, and the
``InferenceStrategy`` mirror that makes the engine a drop-in behind the reference's plugin boundary.
"""
from .config import ModelGeometry, geometry_3b, geometry_tiny  # noqa: F401

__all__ = ["ModelGeometry", "geometry_3b", "geometry_tiny"]
