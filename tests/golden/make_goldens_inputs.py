"""Input cases shared by make_goldens.py (needs /root/reference) and the tests (do not)."""
import numpy as np


def mix_inputs():
    """(name, audio) cases of the peak mixers: f32 / f64, two channels and one, a nearly silent stem, a silent pair."""
    rng = np.random.default_rng(404)
    cases = []
    for i, (n, dt, scale) in enumerate([(4000, np.float32, 1.0), (4001, np.float64, 0.3), (16000, np.float32, 2.5), (333, np.float32, 0.01)]):
        cases.append((f"pair_{i}", (rng.standard_normal((n, 2)) * scale).astype(dt)))
    quiet = (rng.standard_normal((2000, 2))).astype(np.float32)
    quiet[:, 1] *= 1e-7
    cases.append(("quiet_stem", quiet))
    cases.append(("silent", np.zeros((500, 2), dtype=np.float32) + np.float32(1e-6)))
    cases.append(("mono", rng.standard_normal((1500, 1)).astype(np.float32)))
    return cases
