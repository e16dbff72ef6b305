"""Build libvitsmi.so (HIP, gfx950 only) in-tree with hipcc.  No torch involved."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvitsmi.so")
SOURCES = ["vitsmi.hip", "g2p.hip", "model.cpp", "onnx_reader.cpp"]
HEADERS = ["kernels.hip.hpp", "conv_engine.hip.hpp", "conv_sx_engine.hip.hpp", "conv_sx_pair.hip.hpp", "sx_split.hip.hpp", "model.hpp",
           "g2p_model.hpp", "onnx_reader.hpp", "../../include/vitsmi.h", "../../include/g2pmi.h"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libvitsmi.so)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def write_build_info():
    """phoonnx_amd/_build_info.json: the commit this tree was built from (the GPU box receives no .git)."""
    import json
    import time
    root = os.path.dirname(HERE)
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5)
        if head.returncode != 0 or not head.stdout.strip():
            return
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                               text=True, timeout=10).stdout.strip() != ""
        with open(os.path.join(HERE, "_build_info.json"), "w") as f:
            json.dump({"commit": head.stdout.strip(), "dirty": dirty, "time": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}, f)
    except Exception:
        pass


def build(force=False, verbose=False):
    write_build_info()
    if not force and not stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd[1:1] = os.environ.get("VITSMI_CXXFLAGS", "").split()  # kernel experiments (-DSX_EXP_...)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvitsmi.so")
    if verbose:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
