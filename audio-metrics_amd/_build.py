"""Build the gfx950 C-ABI library in-tree with hipcc (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libaudio_metrics_hip.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libaudio_metrics_hip.so")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(PKG_DIR), "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_library(force=False, verbose=False):
    """Compile every csrc/*.hip into one shared object.  Objects are built in
    parallel (one hipcc per file) and linked with hipcc -shared."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, "-c", src, "-o", obj]
        procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for cmd, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(link), r.stdout))
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
