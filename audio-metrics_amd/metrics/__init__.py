from .fad import frechet_distance                                   # noqa: F401
from .kd import kernel_distance, kid_features_to_metric              # noqa: F401
from .prdc import prdc, nearest_neighbour_distances                  # noqa: F401
from .apa import apa, apa_compute_d_x_xp, _apa                       # noqa: F401
