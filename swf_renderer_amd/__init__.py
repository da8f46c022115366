"""swf_renderer_amd -- MI355X-native SWF vector rasterizer behind the reference's render(Stage) API.

Python is host glue only: it marshals swf-tree JSON into the C structs of include/swfr.h and calls
libswfr.so (hand-written HIP kernels + C++ host).  There is no CPU rasterization path in this package;
importing works without a GPU, rendering does not.
"""
from .api import Renderer, SwfrError, load_library, library_path  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["Renderer", "SwfrError", "load_library", "library_path", "synth"]
