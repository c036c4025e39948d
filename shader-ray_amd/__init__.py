"""shader-ray hot path for MI355X: host loaders + BVH (C++), HIP tracer (C ABI).

The directory name carries a hyphen (it is fixed by the project layout), so import it
through `__graft_entry__.load_package()`, which registers it as `shader_ray_amd`.
"""
from . import _native, host, multigpu, scenes, tracer  # noqa: F401
from .host import World, load_background  # noqa: F401
from .tracer import Scene  # noqa: F401
