"""Flat-name shim: put this directory first on sys.path and the reference's callers
(`from ecapa_annote import ...`) bind the MI355X implementation instead of the reference module."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from speech_diarization_amd.ecapa_annote import *  # noqa: E402,F401,F403
from speech_diarization_amd import ecapa_annote as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
