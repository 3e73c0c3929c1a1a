"""The error bound of the f16 filter passes (csrc/pairwise_fast.h: |a - t| <= fast_c(D) (|x|^2 + G)) under stress.

The A/B build can round the f16 copies to fewer mantissa bits (AM_HALF_DROP_BITS = n) and scales the bound's leading term
with it (u = 2^-(11-n) + 2^-11).  If the bound and the queue / verify / fallback protocol around it are right, the OUTPUTS
do not depend on n at all: a coarser filter only queues more pairs for the exact evaluation (and, far enough out, sends rows
or the whole call to the exact kernels).  One wrong constant, one pair dropped by a gate that the element-wise test would
have kept, and the checksums differ.  Subprocesses: the library reads the knob once per process."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, pattern, env):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], env=dict(os.environ, **env), capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-1500:]
    m = re.search(pattern, res.stdout)
    assert m, res.stdout[-1500:]
    return m.group(1)


@pytest.mark.parametrize("data,rows,dim,k", [("randn", 40000, 128, 5), ("unit", 33000, 192, 10)])
def test_outputs_do_not_depend_on_the_filter_precision(data, rows, dim, k):
    base = {"AM_HIP_LIBRARY": "dev", "AB_ROWS": str(rows), "AB_ROWS2": str(rows), "AB_DIM": str(dim), "AB_K": str(k), "AB_REPS": "1",
            "AB_DATA": data, "AB_SEED": "7"}
    radii, counts = {}, {}
    for drop in (0, 2, 4):
        env = dict(base, AM_HALF_DROP_BITS=str(drop))
        radii[drop] = _run("ab_knn.py", r"radii sha1 ([0-9a-f]+)", env)
        counts[drop] = _run("ab_cross.py", r"sha1 ([0-9a-f]+)", env)
    exact = dict(base, AM_KNN_FAST="0", AM_PRDC_FAST="0")                  # the exact f32 kernels, no filter at all
    assert len(set(radii.values())) == 1 and radii[0] == _run("ab_knn.py", r"radii sha1 ([0-9a-f]+)", exact), radii
    assert len(set(counts.values())) == 1 and counts[0] == _run("ab_cross.py", r"sha1 ([0-9a-f]+)", exact), counts
