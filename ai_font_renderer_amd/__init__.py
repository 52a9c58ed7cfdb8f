"""Import alias: the product package lives in ``ai-font-renderer_amd/`` (a name Python cannot
import directly because of the hyphens); this stub makes it importable as ``ai_font_renderer_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ai-font-renderer_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
