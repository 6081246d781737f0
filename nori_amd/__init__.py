"""nori_amd -- MI355X (gfx950) hot path of the Nori renderer.

The product is the HIP library behind include/nori_hip.h (`nori_amd/lib/
libnori_hip.so`, built by `__graft_entry__.build()`), the C++ host that keeps
Nori's NoriObject/XML surface, and this thin Python glue used by tests, bench.py
and the multi-GPU launcher.  Importing the package does not load the library;
`Renderer()` does and raises if it is missing.
"""
from ._capi import NoriError  # noqa: F401
from .scene import Bsdf, Camera, Integrator, Mesh, RFilter, Scene  # noqa: F401
from .render import Renderer, develop_host  # noqa: F401
