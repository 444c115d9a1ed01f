"""Import shim: the package directory is `boss.jl_amd/` (not a valid dotted module name), so this
module turns itself into that package: `import boss_jl_amd`, `from boss_jl_amd import api`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "boss.jl_amd")]
__package__ = "boss_jl_amd"
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
