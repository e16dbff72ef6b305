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
        if head.returncode == 0 and head.stdout.strip():
            dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"],
                                   capture_output=True, text=True, timeout=10).stdout.strip() != ""
            # (a reused library whose sources hash as recorded, in a clean tree: HEAD holds exactly these sources - name it)
            if compiled or not dirty:
                info["commit"] = head.stdout.strip()
                info["dirty"] = dirty
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


def _regs(tok):
    """'v[116:119]' -> {116..119}, 'v174' -> {174}; anything else (SGPRs, AGPRs, literals) -> empty"""
    import re
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def spill_hazards(asm_text, kernel_filter=None):
    """The hazard behind the no-spill rule, checked on the ISA itself (DESIGN 5.1g hazard 1): the conv engines issue
    ASYNCHRONOUS inline-asm loads (global_load / ds_read inside ';;#ASMSTART' blocks) whose destination registers only
    become valid behind a counted s_waitcnt the compiler knows nothing about.  If the register allocator spills such a
    register between the load and its first use it saves the OLD contents (and restores them later): silently wrong
    numbers.  For every kernel of `asm_text` (hipcc -S output) that spills at all this is a forward dataflow analysis over
    the kernel's control-flow graph - basic blocks cut at labels and branches, a block's entry state the UNION of its
    predecessors' exit states, iterated to the fixed point, so a load at the bottom of a loop body meets a spill at its top
    and a spill reached through a forward branch into a loop is seen like any other.  The state is the set of VGPRs that are
    destinations of asm loads not yet READ by any instruction (the hand-written wait sits in front of the first use) and not
    yet behind a full drain of their counter (s_waitcnt vmcnt(0) / lgkmcnt(0)); a scratch_store (spill) of a register in
    that set is a finding.  A spill anywhere else - of values the compiler itself produced, e.g. in an epilogue after the
    last asm load has been consumed - is harmless and is what the kernels that still spill do.  Returns {kernel: [(line
    number within the kernel, instruction)]} for kernels with findings, and the number of kernels examined that spill."""
    import re
    out, spilling = {}, 0
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", asm_text):
        name = m.group(1)
        if kernel_filter and not kernel_filter(name):
            continue
        end = asm_text.find(".Lfunc_end", m.end())
        lines = asm_text[m.end():end if end > 0 else len(asm_text)].split("\n")
        if not any("scratch_store" in l for l in lines):
            continue
        spilling += 1
        # instructions (line number, text, inside an asm block) and the labels in front of them
        ins, label_at, in_asm = [], {}, False
        for i, raw in enumerate(lines):
            st = raw.strip()
            if st.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if st.startswith(";;#ASMEND"):
                in_asm = False
                continue
            l = raw.split(";")[0].strip()
            if not l:
                continue
            lm = re.match(r"^(\.?[\w$.]+):$", l)
            if lm:
                label_at[lm.group(1)] = len(ins)
                continue
            if l.startswith("."):
                continue
            ins.append((i, l, in_asm))
        n = len(ins)
        if n == 0:
            continue

        def branch_of(l):
            t = l.split()
            if len(t) == 2 and t[0].startswith(("s_cbranch", "s_branch")) and t[1] in label_at:
                return t[0], label_at[t[1]]
            return None, None

        leaders = {0}
        for k, (_, l, _) in enumerate(ins):
            op, tgt = branch_of(l)
            if op:
                leaders.add(tgt)
                leaders.add(k + 1)
            elif l.startswith(("s_endpgm", "s_setpc")):
                leaders.add(k + 1)
        leaders = sorted(x for x in leaders if x < n)
        block_of = {b: j for j, b in enumerate(leaders)}
        bounds = [(b, leaders[j + 1] if j + 1 < len(leaders) else n) for j, b in enumerate(leaders)]
        succ = []
        for lo, hi in bounds:
            _, l, _ = ins[hi - 1]
            op, tgt = branch_of(l)
            sset = set()
            if op and tgt < n:
                sset.add(block_of[tgt])
            if not (op and op.startswith("s_branch")) and not l.startswith(("s_endpgm", "s_setpc")) and hi < n:
                sset.add(block_of[hi])
            succ.append(sset)
        findings = []

        def transfer(j, state):
            state = dict(state)
            lo, hi = bounds[j]
            for k in range(lo, hi):
                i, l, ia = ins[k]
                parts = l.replace(",", " ").split()
                op, args = parts[0], parts[1:]
                if op == "s_waitcnt":
                    if "vmcnt(0)" in l:
                        for r in [r for r, kd in state.items() if kd == "vm"]:
                            del state[r]
                    if "lgkmcnt(0)" in l:
                        for r in [r for r, kd in state.items() if kd == "lgkm"]:
                            del state[r]
                    continue
                if op.startswith("scratch_store"):
                    hit = set().union(*[_regs(a) for a in args]) & set(state)
                    if hit and (i, lines[i].strip()) not in findings:
                        findings.append((i, lines[i].strip()))
                    continue
                is_load = op.startswith(("global_load", "ds_read", "buffer_load", "flat_load", "scratch_load"))
                dst = _regs(args[0]) if args and (is_load or op.startswith("v_")) else set()
                srcs = set().union(*[_regs(a) for a in (args[1:] if dst else args)]) if args else set()
                for r in srcs:      # a read: the hand-written wait in front of the first use has passed
                    state.pop(r, None)
                if ia and is_load and not op.startswith("scratch"):
                    kind = "lgkm" if op.startswith("ds_") else "vm"
                    for r in dst:
                        state[r] = kind
                else:
                    for r in dst:   # overwritten by something the compiler tracks
                        state.pop(r, None)
            return state

        entry = [dict() for _ in bounds]
        work, seen = [0], {0}
        # (also seed every block: code only reachable through computed jumps is still examined, from an empty state)
        work += [j for j in range(1, len(bounds))]
        seen |= set(work)
        while work:
            j = work.pop()
            seen.discard(j)
            ex = transfer(j, entry[j])
            for sj in succ[j]:
                merged = dict(entry[sj])
                changed = False
                for r, kd in ex.items():
                    if r not in merged:
                        merged[r] = kd
                        changed = True
                if changed:
                    entry[sj] = merged
                    if sj not in seen:
                        seen.add(sj)
                        work.append(sj)
        if findings:
            out[name] = sorted(findings)
    return out, spilling


def sgpr_vmem_hazards(asm_text):
    """Hazard 5 of the hand-scheduled kernels, checked on the ISA: a vector-memory instruction needs five wait
    states after a VALU instruction (v_readlane / v_readfirstlane) wrote the scalar register it takes its base from.  hipcc's
    hazard recogniser inserts them for the memory instructions it emits itself - not for those inside inline asm, and under
    scalar-register pressure it parks uniform values in VGPR lanes and fetches them with v_readlane right in front of their
    use.  Returns {kernel: [(line, wait states found, the VALU write, the memory instruction)]} for every inline-asm memory
    instruction that follows such a write of one of its scalar operands too closely (s_nop N counts N + 1)."""
    import re
    out = {}
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", asm_text):
        name = m.group(1)
        end = asm_text.find(".Lfunc_end", m.end())
        ins, in_asm = [], False
        for i, l in enumerate(asm_text[m.end():end if end > 0 else len(asm_text)].split("\n")):
            st = l.strip()
            if st.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if st.startswith(";;#ASMEND"):
                in_asm = False
                continue
            ls = l.split(";")[0].strip()
            if ls and not ls.endswith(":") and not ls.startswith("."):
                ins.append((i, ls, in_asm))
        bad = []
        for n, (i, ls, ia) in enumerate(ins):
            if not ia or not re.match(r"(global_|buffer_|flat_|scratch_)", ls):
                continue
            sg = set()
            for a, b in re.findall(r"s\[(\d+):(\d+)\]", ls):
                sg |= set(range(int(a), int(b) + 1))
            for a in re.findall(r"(?<![\w\[])s(\d+)\b", ls):
                sg.add(int(a))
            ws = 0
            for back in range(1, 8):
                if n - back < 0 or ws >= 5:
                    break
                _, pl, _ = ins[n - back]
                mm = re.match(r"(v_readlane_b32|v_readfirstlane_b32) s(\d+)", pl)
                if mm and int(mm.group(2)) in sg:
                    bad.append((i, ws, pl, ls))
                    break
                mn = re.match(r"s_nop (\d+)", pl)
                ws += (int(mn.group(1)) + 1) if mn else 1
        if bad:
            out[name] = bad
    return out


def isa_of(src, extra_flags=()):
    """hipcc -S (device only) of one translation unit, as text"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "tu.s")
        r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", *extra_flags, "-x", "hip",
                            "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError(f"hipcc -S failed on {src}")
        with open(o) as f:
            return f.read()


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
    # one builder at a time per tree: N ranks that find the library stale would otherwise compile the same objects into the
    # same files side by side (_ffi.load() calls this from every process); the others wait, then find it fresh
    import fcntl
    try:
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
        lk = open(os.path.join(HERE, "build", ".lock"), "w")
    except OSError:
        # a read-only install (site-packages of another user, a read-only image layer): nothing can be compiled into this tree
        # anyway - a fresh library is used as it is, a stale one is the error it would have been further down
        if not force and not stale():
            return _build_locked(False, verbose)
        raise
    with lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    sha = source_sha()
    if not force and not stale():
        if read_build_info().get("source_sha") == sha:
            info = write_build_info(False, sha)
            print(f"phoonnx_amd.build: reused libvitsmi.so (source_sha {sha} matches the record; built from "
                  f"{info.get('commit', '?')}{'+dirty' if info.get('dirty') else ''})", file=sys.stderr)
        else:
            # fresh by file times only (a library without a record): usable, but its content was never checked against these
            # sources, so no hash is recorded for it - the next call falls back to the file times again
            print("phoonnx_amd.build: reused libvitsmi.so (no build record: newer than every source by mtime only)", file=sys.stderr)
        return LIB
    # one hipcc process per translation unit, side by side, then one link (the HIP files dominate: minutes each)
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
            "-Rpass-analysis=kernel-resource-usage"]  # (the remarks are parsed below: register spills are an ERROR here)
    base[1:1] = os.environ.get("VITSMI_CXXFLAGS", "").split()  # kernel experiments (-DSX_EXP_...)

    # The ISA checks below read the assembly of THIS compile (-save-temps=obj leaves <unit>-hip-amdgcn-amd-amdhsa-gfx950.s
    # next to the object; each unit in a directory of its own, removed once read): the code that is checked is the code that
    # ships, and a stale library costs one compile per unit, not two.
    isas = {}

    def compile_one(src):
        stem = os.path.splitext(src)[0]
        obj = os.path.join(objdir, stem + ".o")
        if not src.endswith(".hip"):
            r = subprocess.run(base + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
            return src, obj, r
        tdir = os.path.join(objdir, "temps_" + stem)
        shutil.rmtree(tdir, ignore_errors=True)
        os.makedirs(tdir)
        tobj = os.path.join(tdir, stem + ".o")
        r = subprocess.run(base + ["-save-temps=obj", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", tobj], capture_output=True, text=True)
        if r.returncode == 0:
            os.replace(tobj, obj)
            with open(os.path.join(tdir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")) as f:
                isas[src] = f.read()
        shutil.rmtree(tdir, ignore_errors=True)
        return src, obj, r

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    spilled, spilled_src = {}, {}
    for src, _, r in results:
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"hipcc failed compiling {src}")
        if verbose:
            sys.stderr.write(r.stderr)
        for k, v in spilling_kernels(r.stderr).items():
            spilled[k] = v
            spilled_src[k] = src
    # The conv engines feed their MFMA loops through ASYNCHRONOUS inline-asm loads (global_load / ds_read whose results are
    # only valid behind a counted s_waitcnt): if the compiler spills such a register between the load and the wait it saves
    # the OLD contents and later restores them - silently wrong results (seen: conv_sx_pair16_kernel<64, 2, 256>, 41 spilled
    # registers, inf in the generator).  Two rules, both build ERRORS:
    #  1. the kernels whose whole life is such a pipeline (the 16x16x32 / short-launch conv engines, the fused pair16 kernel,
    #     the 16-bit attention) must not spill at all;
    #  2. every OTHER kernel that spills (today: conv_sx_pair_kernel's 64-channel and fused-chain instantiations - on the
    #     default f16x3 path - the f32 engine's small tiles, the fp32 attention) is checked on its ISA: no spill store may
    #     hit a register that is the destination of an asm load not yet consumed (spill_hazards).  Their spills sit in the
    #     epilogues, behind the last asm load's use; a compiler that moves one into the pipeline fails the build here.
    no_spill = ("conv_sx_kernel", "conv_sx_pair16", "conv_sx_small_kernel", "attention_relpos16_kernel")
    bad = {k: v for k, v in spilled.items() if any(n in k for n in no_spill)}
    if bad:
        raise RuntimeError("kernels with asynchronous inline-asm loads must not spill registers: " + ", ".join(f"{k} ({v})" for k, v in bad.items()))
    # the ISA of every HIP unit (device only, side by side): hazards 1 and 5 are properties of the generated code
    hip_units = [u for u in SOURCES if u.endswith(".hip")]
    missing = [u for u in hip_units if u not in isas]
    if missing:
        raise RuntimeError(f"no device assembly was kept for {missing} (-save-temps=obj): the ISA checks cannot run")
    h5 = {}
    for u in hip_units:
        h5.update(sgpr_vmem_hazards(isas[u]))
    if h5:
        raise RuntimeError("an inline-asm memory instruction reads a scalar base a VALU instruction wrote fewer than five wait states "
                           "before (stale base: wild addresses): " +
                           "; ".join(f"{k}: '{v[0][3]}' {v[0][1]} wait states after '{v[0][2]}' (+{len(v) - 1} more)" for k, v in h5.items()))
    if spilled:
        units = sorted(set(spilled_src.values()))
        hazards, checked = {}, 0
        for u in units:
            f, n = spill_hazards(isas[u])
            hazards.update(f)
            checked += n
        if checked < len(set(spilled)):
            raise RuntimeError(f"spill check: the remarks name {len(set(spilled))} spilling kernels, the ISA of {units} shows {checked}")
        if hazards:
            raise RuntimeError("a spilled register is the destination of an asynchronous inline-asm load still in flight: " +
                               "; ".join(f"{k}: {v[0][1]} (+{len(v) - 1} more)" for k, v in hazards.items()))
        print(f"phoonnx_amd.build: {len(spilled)} kernels spill registers ({', '.join(sorted(set(k.split('vitsmi')[-1][:28] for k in spilled)))}); "
              f"ISA-checked ({', '.join(units)}): no spill store hits an asynchronous load's destination", file=sys.stderr)
    tmp_lib = LIB + f".tmp{os.getpid()}"
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + [o for _, o, _ in results],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libvitsmi.so")
    os.replace(tmp_lib, LIB)  # (a process that has the old library mapped keeps its inode)
    write_build_info(True, sha)
    print(f"phoonnx_amd.build: compiled libvitsmi.so (source_sha {sha})", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m phoonnx_amd.build --variant <out.so> "<-D flags>"
        i = sys.argv.index("--variant")
        print(build_variant(os.path.abspath(sys.argv[i + 1]), sys.argv[i + 2]))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
