"""Import shim: the product package lives in ``audio-metrics_amd/`` (hyphenated,
as the repository layout prescribes), which Python cannot import by name."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "audio-metrics_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
