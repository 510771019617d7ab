"""rt355 — MI355X-native wavefront path tracer hot path (generate -> extend -> shade -> connect)
behind the Renderer::Tick()/Scene API of merijn-23/magr-ray-tracer.

Python here is harness glue (ctypes + numpy); the product is the HIP library
(librt355.so, include/rt355.h) and the C++ host mirror (librt355_host.so, include/rt355_host.h).
"""
from . import _lib  # noqa: F401
from ._lib import (ACCEL_BVH2, ACCEL_BVH4, SAMPLING_COSINE, SAMPLING_HEMISPHERE, SHADING_NEE,  # noqa: F401
                   SHADING_SIMPLE, NativeLibraryMissing)

__all__ = ["Device", "Renderer", "Scene", "scenes"]


def __getattr__(name):
    if name in ("Device", "Renderer", "RtError"):
        from . import renderer
        return getattr(renderer, name)
    if name in ("Scene", "SceneArrays", "make_camera", "material"):
        from . import scene
        return getattr(scene, name)
    if name == "scenes":
        import importlib
        return importlib.import_module(".scenes", __name__)
    raise AttributeError(name)
