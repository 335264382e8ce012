"""raytracer_amd -- MI355X (gfx950) implementation of the Soulztorm/Raytracer per-pixel tracing path.

`Renderer`, `Scene`, `Camera` mirror the reference's classes of the same names over the C ABI in
include/rt_mi355x.h; `scenes` holds the reference's default scene and the synthetic benchmark meshes.
"""
from .renderer import Camera, FlatKDTree, GroupRenderer, Image, KDTree, Renderer, Scene, Settings  # noqa: F401
from . import scenes  # noqa: F401
