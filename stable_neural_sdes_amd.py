"""Import shim: the package directory is named ``stable-neural-sdes_amd`` (not a valid identifier), so this
module makes it importable as ``stable_neural_sdes_amd`` (sub-modules resolve through ``__path__``)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'stable-neural-sdes_amd')]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
