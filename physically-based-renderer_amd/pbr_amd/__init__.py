"""pbr_amd — host side of the MI355X path-tracing core.

  pbr_amd.ptc     ctypes binding of include/ptc.h (libptc.so, HIP, gfx950 only) and `PathTracer`
  pbr_amd.scene   the reference's scene/mesh/camera interface mirrored in numpy
  pbr_amd.scenes  procedural stand-ins for BASELINE.json's configs
  pbr_amd.dist    one-process-per-GPU tile sharding + RCCL framebuffer reduce
"""
from . import dist, gltf, scene, scenes  # noqa: F401
from .ptc import (DEVICE_NONE, INTEGRATOR_PATH, INTEGRATOR_RASTER_COMPAT, INTEGRATOR_RASTER_GBUFFER16, Group, PathTracer, PtcError,  # noqa: F401
                  comm_unique_id, load_library)
