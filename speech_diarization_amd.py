"""Import alias: `import speech_diarization_amd` -> the package in `speech-diarization_amd/`.

The package directory keeps the repository's name (which contains a hyphen and is
therefore not importable as-is); this module gives it an importable name by
pointing its `__path__` at that directory and executing the package's `__init__`.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "speech-diarization_amd")
__path__ = [_PKG_DIR]
__file__ = _os.path.join(_PKG_DIR, "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
