"""MI355X-native distribution-distance engine behind the audio-metrics API.

Importable as ``audio_metrics_amd`` (the directory name carries a hyphen; the
top-level ``audio_metrics_amd/`` shim points Python at it).
"""
from . import _lib, hip_ops                        # noqa: F401
from ._build import build_library, LIB_PATH        # noqa: F401

__version__ = "0.1.0"
