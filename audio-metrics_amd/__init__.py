"""MI355X-native distribution-distance engine behind the audio-metrics API.

Importable as ``audio_metrics_amd`` (the directory name carries a hyphen; the
top-level ``audio_metrics_amd/`` shim points Python at it).
"""
from . import _lib, hip_ops                        # noqa: F401
from ._build import build_library, LIB_PATH        # noqa: F401

from .data import AudioMetricsData, ensure_tensor, ensure_ndarray            # noqa: F401
from . import metrics                                                         # noqa: F401
from .metrics.fad import frechet_distance                                     # noqa: F401
from .metrics.kd import kernel_distance, kid_features_to_metric               # noqa: F401
from .metrics.prdc import prdc, nearest_neighbour_distances                   # noqa: F401
from .metrics.apa import apa, apa_compute_d_x_xp                              # noqa: F401

from .embed import ItemCategory, embedding_pipeline                            # noqa: F401
from .projection import IncrementalPCA                                         # noqa: F401
from .audio_metrics import AudioMetrics                                        # noqa: F401

__version__ = "0.1.0"
