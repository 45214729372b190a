"""x_maps_amd -- MI355X-native (gfx950) implementation of the X-maps per-event disparity-lookup /
depth-reprojection hot path, behind the reference's DepthReprojectionProcessor / DepthReprojectionPipe
event-callback API.  Compute lives in libxmaps_hip.so (hand-written HIP, C-ABI in include/xmaps.h);
there is no CPU fallback.
"""
import os as _os

# Takes effect only if the HIP runtime has not been initialised yet (it reads its settings once): ROCclr drains every stream
# on the CPU after each 1000 commands by default, a multi-millisecond stall of the frame pipeline every ~1300 frames.
# Long-running hosts should export it themselves.
_os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")

from ._native import XMapsNativeError, build_native, load_library  # noqa: E402,F401
from .engine import FrameStats, XMapsEngine  # noqa: E402,F401

__all__ = ["XMapsEngine", "FrameStats", "XMapsNativeError", "build_native", "load_library"]
