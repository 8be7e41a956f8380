"""Import alias for the package directory ``raytracing.jl_amd/``.

The package directory carries the reference's name (``RayTracing.jl`` + ``_amd``) and a dot
is not legal in a Python module name, so this one-file loader registers the directory
package under the importable name ``raytracing_jl_amd``:

    import raytracing_jl_amd as rt
    tg = rt.TrackGenerator(model, 8, 0.02); rt.trace(tg); rt.segmentize(tg)
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracing.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
