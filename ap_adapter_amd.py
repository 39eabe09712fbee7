"""Import shim: the package directory is named ``ap-adapter_amd`` (not a valid Python identifier), so
``import ap_adapter_amd`` lands here and turns this module into that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ap-adapter_amd")]
__package__ = "ap_adapter_amd"
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"), globals())
