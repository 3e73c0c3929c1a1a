#!/usr/bin/env python3
"""Builds an experimental variant of the library: tools/build_variant.py <name> [-DFLAG ...] -> audio-metrics_amd/lib/libam_<name>.so
(select it with AM_HIP_LIBRARY=libam_<name>.so; the dev knobs are compiled in)."""
import importlib.util
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("am_build", os.path.join(root, "audio-metrics_amd", "_build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
name, flags = sys.argv[1], ["-DAM_DEV_KNOBS", *sys.argv[2:]]
hipcc = b._hipcc()
path = os.path.join(b.LIB_DIR, f"libam_{name}.so")
procs = b._start_objects(hipcc, os.path.join(b.LIB_DIR, "obj_" + name), flags)
print(b._finish(hipcc, procs, path, False, tuple(flags)))
