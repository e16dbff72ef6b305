"""Build libvitsmi.so (HIP, gfx950 only) in-tree with hipcc.  No torch involved."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvitsmi.so")
SOURCES = ["vitsmi.hip", "tu_conv_f32.hip", "tu_sx.hip", "tu_sx_s16.hip", "tu_sx_s16p.hip", "tu_sx_s32.hip", "tu_sx_bf16.hip", "tu_sx_h1.hip", "tu_pair.hip", "tu_pair16.hip", "g2p.hip",
           "model.cpp", "onnx_reader.cpp"]
# every header under csrc/ (a header missing from a hand-kept list once left the library unrebuilt after an edit), and the C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + ["../../include/vitsmi.h", "../../include/g2pmi.h"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libvitsmi.so)")


def source_sha():
    """sha256 over every file the library is compiled from (and the extra compiler flags): what `libvitsmi.so` must match."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(f.encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(os.environ.get("VITSMI_CXXFLAGS", "").encode())
    return h.hexdigest()[:16]


INFO = os.path.join(HERE, "_build_info.json")


def read_build_info():
    import json
    try:
        with open(INFO) as f:
            return json.load(f)
    except Exception:
        return {}


def stale():
    """The library is missing, or was built from other sources than the ones in the tree.  Decided by CONTENT (the recorded
    source hash), not by mtime: built `.so` files travel to the GPU box next to sources whose mtimes mean nothing there."""
    if not os.path.exists(LIB):
        return True
    rec = read_build_info().get("source_sha")
    if rec is None:  # (a library without a record: fall back to mtimes)
        t = os.path.getmtime(LIB)
        return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)
    return rec != source_sha()


def write_build_info(compiled, sha):
    """phoonnx_amd/_build_info.json: whether this call compiled, the hash of the sources the library was built from, and
    the commit of the tree (the GPU box receives no .git)."""
    import json
    import time
    root = os.path.dirname(HERE)
    info = dict(read_build_info())
    info.update({"compiled": bool(compiled), "source_sha": sha})
    if compiled:
        info["time"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5)
        if head.returncode == 0 and head.stdout.strip() and compiled:
            info["commit"] = head.stdout.strip()
            info["dirty"] = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"],
                                           capture_output=True, text=True, timeout=10).stdout.strip() != ""
    except Exception:
        pass
    try:
        with open(INFO, "w") as f:
            json.dump(info, f)
    except Exception:
        pass
    return info


def spilling_kernels(remarks):
    """{demangled-ish kernel name: spilled VGPRs} from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    import re
    out, name = {}, None
    for ln in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", ln)
        if m and name and int(m.group(1)) > 0:
            out[name] = int(m.group(1))
    return out


def build_variant(out, cxxflags):
    """Kernel experiments: the library once more with extra -D switches, as `out` (loaded through VITSMI_LIB); the product
    library and its record are left alone."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build", "variant_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + cxxflags.split()

    def one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        return obj, subprocess.run(base + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        res = list(ex.map(one, SOURCES))
    for obj, r in res:
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed compiling " + obj)
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _ in res], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    return out


def build(force=False, verbose=False):
    sha = source_sha()
    if not force and not stale():
        info = write_build_info(False, sha)
        print(f"phoonnx_amd.build: reused libvitsmi.so (source_sha {sha} matches the record; built from "
              f"{info.get('commit', '?')}{'+dirty' if info.get('dirty') else ''})", file=sys.stderr)
        return LIB
    # one hipcc process per translation unit, side by side, then one link (the HIP files dominate: minutes each)
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
            "-Rpass-analysis=kernel-resource-usage"]  # (the remarks are parsed below: register spills are an ERROR here)
    base[1:1] = os.environ.get("VITSMI_CXXFLAGS", "").split()  # kernel experiments (-DSX_EXP_...)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        r = subprocess.run(base + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        return src, obj, r

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    spilled = {}
    for src, _, r in results:
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"hipcc failed compiling {src}")
        if verbose:
            sys.stderr.write(r.stderr)
        spilled.update(spilling_kernels(r.stderr))
    # The conv engines feed their MFMA loops through ASYNCHRONOUS inline-asm loads (global_load / ds_read whose results are
    # only valid behind a counted s_waitcnt): if the compiler spills such a register between the load and the wait it saves
    # the OLD contents and later restores them - silently wrong results (seen: conv_sx_pair16_kernel<64, 2, 256>, 41 spilled
    # registers, inf in the generator).  A spilling instantiation of those kernels is therefore a build error.
    # (enforced for the split-operand engine's kernels - the generator, flow and encoder of every full-size voice; the
    # 32x32x16 pair kernel, now reached through test hooks only, and the f32 engine's small tiles spill 3-34 registers - none
    # of them an asynchronous destination so far, every parity test green - and are reported, not refused)
    no_spill = ("conv_sx_kernel", "conv_sx_pair16", "conv_sx_small_kernel", "attention_relpos16_kernel")
    bad = {k: v for k, v in spilled.items() if any(n in k for n in no_spill)}
    if bad:
        raise RuntimeError("kernels with asynchronous inline-asm loads must not spill registers: " + ", ".join(f"{k} ({v})" for k, v in bad.items()))
    if spilled:
        print("phoonnx_amd.build: note: " + ", ".join(f"{k.split('vitsmi')[-1][:40]} spills {v}" for k, v in spilled.items()), file=sys.stderr)
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [o for _, o, _ in results],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libvitsmi.so")
    write_build_info(True, sha)
    print(f"phoonnx_amd.build: compiled libvitsmi.so (source_sha {sha})", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m phoonnx_amd.build --variant <out.so> "<-D flags>"
        i = sys.argv.index("--variant")
        print(build_variant(os.path.abspath(sys.argv[i + 1]), sys.argv[i + 2]))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
