"""mygpuraytracer_amd -- MI355X-native Monte-Carlo path tracer.

Drop-in for the ``pathtraceInit / pathtrace(iter) / pathtraceFree`` path and the ``scenes/*.txt`` loader of
nkkk98/MyGPURaytracer.  The compute path is hand-written HIP for gfx950 in ``csrc/`` behind the C ABI declared in
``include/mi355x_pathtracer.h`` and ``include/mi355x_stream_compaction.h``; this package is the thin Python host
side over that ABI (ctypes), used by the tests, ``bench.py`` and the multi-GPU driver.

There is no CPU fallback: importing works anywhere (so that CPU-only tooling can inspect the ABI), but creating a
tracer without the built library or without a HIP device raises.
"""
from .api import (  # noqa: F401
    LIB_PATH,
    PathTracerError,
    Options,
    Scene,
    Tracer,
    MultiTracer,
    StreamCompaction,
    build_library,
    load_library,
)

__all__ = ["LIB_PATH", "PathTracerError", "Options", "Scene", "Tracer", "MultiTracer", "StreamCompaction", "build_library",
           "load_library"]
