"""MI355X-native path-tracing core behind the reference renderer's scene structs
and render-call surface.  All rendering happens in librt_mi355x.so (HIP, gfx950);
importing this package fails if that library has not been built."""
from ._abi import (LIB_PATH, SEM_CPU, SEM_GLSL, SAMPLE_JITTER, SAMPLE_GRID, SAMPLE_GRID_COMPAT, RNG_HASH, RNG_GLSL,
                   FLAG_METAL_UNNORMALIZED, FLAG_BOX_SIDES, FLAG_UNIT_GAUSS, FLAG_LITERAL_GEOMETRY, MATERIAL_DTYPE, SHAPE_DTYPE, TRIANGLE_DTYPE, RtError, lib)
from .host import (Material, Light, Lambertian, Metal, Dielectric, DebugMaterial, Shape, Circle, Box, Cube, SimpleTriangle,
                   Camera, SimpleMesh, RayTracingComputeShader, device_info, pack)
from . import scenes

__all__ = [n for n in dir() if not n.startswith("_")]
