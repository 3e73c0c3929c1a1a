"""Register budget of the two wide f16 filter kernels, checked at compile time (no GPU, no clock).

They run at one workgroup of 8 waves per CU, i.e. 256 registers per lane, with 128 of them holding accumulators: a change
that tips them into scratch memory costs a factor of 2-3 in speed (it happened once to the k = 10 instantiation) and shows
nowhere but in a timing.  hipcc's resource-usage remarks make it a deterministic check: the membership filter and the
k <= 5 sweep must not touch scratch at all; the k <= 10 sweep (22 more list registers) is allowed its known handful of
spilled registers (cold paths), not more."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio-metrics_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_wide_kernels_stay_within_the_register_file():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    import importlib.util
    spec = importlib.util.spec_from_file_location("am_build", os.path.join(ROOT, "audio-metrics_amd", "_build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)                                         # the flags the shipped library is built with
    r = subprocess.run([hipcc, *build.HIPCC_FLAGS, "-c", "pairwise_wide.hip", "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "ScratchSize \\[bytes/lane\\]", "VGPRs Spill", "Occupancy \\[waves/SIMD\\]"):
            m = re.search(r"remark:\s+%s: (\d+)" % key, line)
            if m and name:
                usage[name][key.split(" ")[0].replace("\\", "")] = int(m.group(1))
    kernels = {n: u for n, u in usage.items() if "wide_kernel" in n}
    assert len(kernels) == 4, list(usage)
    for n, u in kernels.items():
        assert u["VGPRs"] <= 256 and u["Occupancy"] >= 2, (n, u)          # two waves per SIMD = the 8-wave workgroup fits
        if "knn_wide_kernelILi11E" in n:
            assert u["ScratchSize"] <= 128, (n, u)
        else:
            assert u["ScratchSize"] == 0, (n, u)
