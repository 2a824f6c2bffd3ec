"""MI355X-native (gfx950) ray-marching backend for OVR's renderer API.

Only what the hot path needs: the C-ABI shared library (csrc/ -> libovr_hip.so), its ctypes binding (_lib), the
host-side mirror of the reference's renderer interface (renderer), the image-plane sharding helpers for
one-process-per-GPU runs (tiles) and the synthetic inputs the reference does not ship (synth)."""
from . import _lib
from ._lib import (GRID_CELL_CENTRED, GRID_VERTEX_CENTRED, JITTER_BLUE_NOISE, JITTER_TEA, SHADE_FULL, SHADE_GRADIENT, SHADE_NONE)
from .renderer import (Camera, CrossDeviceBuffer, DeviceHIP, FrameBufferData, Scene, TransferFunction, create_renderer)
from . import imageio, synth, tiles, vidi3d

__all__ = ["Camera", "CrossDeviceBuffer", "DeviceHIP", "FrameBufferData", "Scene", "TransferFunction", "create_renderer",
           "imageio", "synth", "tiles", "vidi3d", "JITTER_TEA", "JITTER_BLUE_NOISE", "SHADE_NONE", "SHADE_GRADIENT", "SHADE_FULL", "GRID_CELL_CENTRED", "GRID_VERTEX_CENTRED"]
