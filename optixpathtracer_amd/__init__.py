"""optixpathtracer_amd — MI355X-native wavefront path tracer behind the reference's
SampleRenderer / LaunchParams / Model surface (bipul-mohanto/OptixPathTracer, SimplePathtracer.h:38-176).

The compute path is libptamd.so (hand-written HIP for gfx950, C ABI in include/pt_amd.h).  There is no
CPU fallback: importing `renderer` (or calling `load_library`) raises if the shared library is missing
or cannot be loaded.
"""
from . import scenes  # noqa: F401
from ._lib import LIB_PATH, build_library, load_library  # noqa: F401

__all__ = ["scenes", "load_library", "build_library", "LIB_PATH"]
