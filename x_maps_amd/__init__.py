"""x_maps_amd -- MI355X-native (gfx950) implementation of the X-maps per-event disparity-lookup /
depth-reprojection hot path, behind the reference's DepthReprojectionProcessor / DepthReprojectionPipe
event-callback API.  Compute lives in libxmaps_hip.so (hand-written HIP, C-ABI in include/xmaps.h);
there is no CPU fallback.
"""
from ._native import XMapsNativeError, build_native, load_library  # noqa: F401
from .engine import FrameStats, XMapsEngine  # noqa: F401

__all__ = ["XMapsEngine", "FrameStats", "XMapsNativeError", "build_native", "load_library"]
