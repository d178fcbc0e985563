"""Import alias: the package lives in `normalizing-flows_amd/` (a directory name Python cannot
import directly); `import normalizing_flows_amd` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'normalizing-flows_amd')
__path__[:] = [_real]
__file__ = _os.path.join(_real, '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
del _f
