"""Register budget of the two wide f16 filter kernels, checked at compile time (no GPU, no clock).

They run at one workgroup of 8 waves per CU, i.e. 256 registers per lane, with 128 of them holding accumulators: a change
that tips them into scratch memory costs a factor of 2-3 in speed (it happened once to the k = 10 instantiation) and shows
nowhere but in a timing.  hipcc's resource-usage remarks make it a deterministic check: none of the four may touch
scratch (round 2 allowed the k <= 10 sweep a handful of spilled registers; the round-3 epilogue needs none)."""
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
        assert u["ScratchSize"] == 0, (n, u)              # (round 3: also the k <= 10 sweep, 240 registers, no spill)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_operand_stationary_kernels_do_not_spill():
    """pairwise_pstat.hip (round 5): 32 instantiations - membership filter and k-NN sweep for 1 .. 8 slabs of 64 f16 per row,
    with / without the row minimum, lists of 6 / 11 - at two waves per SIMD.  At 512 columns 96 registers hold P fragments, 64
    accumulators, 16 Q fragments: the epilogues have ~70 left, and the k <= 10 sweep used every one of them (the seventh and
    eighth slab of the P rows live in LDS for exactly that reason).  None may touch scratch."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    import importlib.util
    spec = importlib.util.spec_from_file_location("am_build", os.path.join(ROOT, "audio-metrics_amd", "_build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    r = subprocess.run([hipcc, *build.HIPCC_FLAGS, "--cuda-device-only", "-c", "pairwise_pstat.hip", "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    name, usage = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key, label in (("VGPRs", "VGPRs"), ("ScratchSize \\[bytes/lane\\]", "scratch"), ("Occupancy \\[waves/SIMD\\]", "occupancy")):
            m = re.search(r"remark:\s+%s: (\d+)" % key, line)
            if m and name:
                usage[name][label] = int(m.group(1))
    kernels = {n: u for n, u in usage.items() if "pstat_kernel" in n}
    assert len(kernels) == 32, sorted(usage)
    for n, u in kernels.items():
        assert u["VGPRs"] <= 256 and u["occupancy"] >= 2 and u["scratch"] == 0, (n, u)
    # round 6: the 256-thread form for rows of up to two slabs (D <= 128) - TWO workgroups per CU, so again two waves per SIMD:
    # 64 accumulators + the P fragments of two row tiles (32 registers per slab) + the epilogue
    narrow = {n: u for n, u in usage.items() if "pstat64_kernel" in n}
    assert len(narrow) == 8, sorted(usage)
    for n, u in narrow.items():
        assert u["VGPRs"] <= 256 and u["occupancy"] >= 2 and u["scratch"] == 0, (n, u)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_kernel_distance_on_the_wide_engine_does_not_spill():
    """kd_wide_kernel: 128 accumulators + an f64 epilogue.  With a run-time power loop per accumulator element the register
    allocator spilled accumulators inside the MAIN loop (684 bytes per lane); the degree-3 form must stay at zero scratch."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    import importlib.util
    spec = importlib.util.spec_from_file_location("am_build", os.path.join(ROOT, "audio-metrics_amd", "_build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    r = subprocess.run([hipcc, *build.HIPCC_FLAGS, "--cuda-device-only", "-c", "kd.hip", "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, scratch = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    wide = {n: v for n, v in scratch.items() if "kd_wide_kernel" in n}
    assert len(wide) == 1 and list(wide.values()) == [0], scratch
