"""Import alias: the package lives in `nerf-hugs_amd/` (not a valid Python identifier);
`import nerf_hugs_amd` resolves its submodules from there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'nerf-hugs_amd')
__path__.insert(0, _real)
with open(_os.path.join(_real, '__init__.py')) as _f:
  exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
