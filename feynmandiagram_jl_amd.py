"""Import shim: the package directory is named ``feynmandiagram.jl_amd`` (a
dot is not importable), so this module turns itself into that package:
``import feynmandiagram_jl_amd as fd`` / ``from feynmandiagram_jl_amd import
Compilers``."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "feynmandiagram.jl_amd")
__path__ = [_pkg]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(_pkg, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg, "__init__.py"), "exec"))
del _f
