"""Import shim: the package directory is ``manisdp-matlab_amd/`` (a hyphen is not
a legal Python identifier), so this one-file module turns itself into the
package ``manisdp_matlab_amd`` by pointing ``__path__`` at that directory and
executing its ``__init__.py``."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_here, "manisdp-matlab_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r") as _fh:
    exec(compile(_fh.read(), __file__, "exec"))
