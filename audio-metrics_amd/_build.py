"""Build the gfx950 C-ABI library in-tree with hipcc (cross-compiles without a GPU)."""
import glob
import hashlib
import json
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libaudio_metrics_hip.so")
# A/B build of the same sources with -DAM_DEV_KNOBS: environment-variable knobs, older engine schedules and debug dumps.
# Never loaded by the product path; tools/ and the tests that force fallback paths ask for it with AM_HIP_LIBRARY=dev.
DEV_LIB_PATH = os.path.join(LIB_DIR, "libaudio_metrics_hip_dev.so")
# -pragma-unroll-threshold: the epilogues of the 256 x 256 filter kernels are `#pragma unroll` loops over 128 accumulator
# registers; past the default budget (16384) the compiler leaves a loop rolled, indexes the accumulator array dynamically and
# puts it in scratch (576 bytes per lane, 10x slower kernels)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-mllvm", "-pragma-unroll-threshold=65536"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libaudio_metrics_hip.so")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _inputs():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        sorted(glob.glob(os.path.join(os.path.dirname(PKG_DIR), "include", "*.h")))


def source_digest(extra=()):
    """sha256 over the CONTENT of every source, header and flag that goes into a library: what decides whether a built
    .so still belongs to the tree (modification times do not survive a checkout or a copy to another machine)."""
    h = hashlib.sha256()
    for flag in (*HIPCC_FLAGS, *extra):
        h.update(flag.encode() + b"\0")
    for path in _inputs():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_path(lib_path):
    return lib_path + ".stamp.json"


def is_stale(lib_path=LIB_PATH, extra=None):
    """True when the library is missing or was built from other sources / flags than the tree holds now."""
    if extra is None:
        extra = ("-DAM_DEV_KNOBS",) if lib_path == DEV_LIB_PATH else ()
    if not os.path.exists(lib_path) or not os.path.exists(_stamp_path(lib_path)):
        return True
    try:
        with open(_stamp_path(lib_path)) as f:
            return json.load(f).get("sources_sha256") != source_digest(extra)
    except (OSError, ValueError):
        return True


def _start_objects(hipcc, obj_dir, extra):
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, *extra, "-c", src, "-o", obj]
        procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    return procs


def _finish(hipcc, procs, lib_path, verbose, extra=()):
    objs = []
    for cmd, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(link), r.stdout))
    with open(_stamp_path(lib_path), "w") as f:
        json.dump({"sources_sha256": source_digest(extra), "flags": [*HIPCC_FLAGS, *extra]}, f)
    return lib_path


def build_library(force=False, verbose=False, dev=True):
    """Compile every csrc/*.hip into the shipped shared object and (dev=True) its -DAM_DEV_KNOBS twin.  Objects are
    built in parallel (one hipcc per file and flavour) and linked with hipcc -shared."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    if force or is_stale(LIB_PATH):
        jobs.append((_start_objects(hipcc, os.path.join(LIB_DIR, "obj"), []), LIB_PATH, ()))
    if dev and (force or is_stale(DEV_LIB_PATH)):
        jobs.append((_start_objects(hipcc, os.path.join(LIB_DIR, "obj_dev"), ["-DAM_DEV_KNOBS"]), DEV_LIB_PATH, ("-DAM_DEV_KNOBS",)))
    for procs, path, extra in jobs:
        _finish(hipcc, procs, path, verbose, extra)
    return LIB_PATH


RCCL_ADAPTER_SRC = os.path.join(CSRC, "rccl", "am_rccl.cpp")
RCCL_ADAPTER_PATH = os.path.join(LIB_DIR, "libaudio_metrics_rccl.so")


def build_rccl_adapter(force=False):
    """libaudio_metrics_rccl.so: am_collectives over an ncclComm_t (csrc/rccl/am_rccl.cpp) - the hooks a host that is not
    Python hands to am_evaluate_sharded_f32.  Its own library: libaudio_metrics_hip.so links no collective library.  Returns
    the path, or None where the image has no RCCL headers."""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        return None
    h = hashlib.sha256()
    for path in (RCCL_ADAPTER_SRC, os.path.join(os.path.dirname(PKG_DIR), "include", "audio_metrics_hip.h")):
        with open(path, "rb") as f:
            h.update(f.read())
    digest = h.hexdigest()
    stamp = _stamp_path(RCCL_ADAPTER_PATH)
    if not force and os.path.exists(RCCL_ADAPTER_PATH) and os.path.exists(stamp):
        try:
            with open(stamp) as f:
                if json.load(f).get("sources_sha256") == digest:
                    return RCCL_ADAPTER_PATH
        except (OSError, ValueError):
            pass
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", RCCL_ADAPTER_SRC, "-o", RCCL_ADAPTER_PATH,
           "-L" + os.path.join(rocm, "lib"), "-lrccl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stdout))
    with open(stamp, "w") as f:
        json.dump({"sources_sha256": digest}, f)
    return RCCL_ADAPTER_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
