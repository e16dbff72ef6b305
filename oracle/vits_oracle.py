"""ctypes front for the C oracle (oracle/vits_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
It reads the `.onnx` with the oracle's own pure-Python walker (onnx_walk.py), resolves
every parameter by following graph nodes (weight-normed flow convs
are folded to anonymous `onnx::Conv_N` initializers and identical initializers are
de-duplicated, so names alone are not enough), and hands named tensors to the C code.
"""
import ctypes
import os
import subprocess

import numpy as np

from onnx_walk import OnnxModel

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build_lib(native=False, force=False):
    """Compile vits_oracle.c -> libvits_oracle[.native].so (gcc, OpenMP)."""
    src = os.path.join(_HERE, "vits_oracle.c")
    out = os.path.join(_HERE, "libvits_oracle_native.so" if native else "libvits_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        arch = "-march=native" if native else "-march=x86-64-v3"
        cmd = ["gcc", "-O3", arch, "-fopenmp", "-shared", "-fPIC", "-std=gnu11", "-o", out, src, "-lm"]
        subprocess.check_call(cmd)
    return out


def load_lib(native=False):
    if native in _LIBS:
        return _LIBS[native]
    lib = ctypes.CDLL(build_lib(native))
    c = ctypes
    lib.vo_new.restype = c.c_void_p
    lib.vo_free.argtypes = [c.c_void_p]
    lib.vo_error.restype = c.c_char_p
    lib.vo_error.argtypes = [c.c_void_p]
    lib.vo_set_tensor.argtypes = [c.c_void_p, c.c_char_p, c.c_void_p, c.c_int, c.POINTER(c.c_int64)]
    lib.vo_set_int.argtypes = [c.c_void_p, c.c_char_p, c.c_int64]
    lib.vo_infer.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                             c.c_void_p, c.c_void_p, c.c_int64]
    lib.vo_vocoder.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p]
    lib.vo_result.restype = c.POINTER(c.c_float)
    lib.vo_result.argtypes = [c.c_void_p, c.c_char_p, c.POINTER(c.c_int), c.POINTER(c.c_int64)]
    lib.vo_conv1d.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_int,
                              c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p]
    lib.vo_conv_transpose1d.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_int,
                                        c.c_int, c.c_int, c.c_int, c.c_void_p]
    lib.vo_num_threads.restype = c.c_int
    lib.vo_attention_core.argtypes = [c.c_void_p] * 5 + [c.c_int] * 5 + [c.c_void_p, c.c_void_p]
    _LIBS[native] = lib
    return lib


def _module_path(node_name):
    parts = [p for p in node_name.split("/") if p]
    return ".".join(parts[:-1]), (parts[-1] if parts else "")


def resolve_weights(model: OnnxModel):
    """-> (tensors {canonical name: float32 array}, ints {key: int})"""
    tensors, ints = {}, {}

    def put(name, arr):
        if arr is not None and name not in tensors:
            tensors[name] = np.ascontiguousarray(arr, dtype=np.float32)

    pad_seen = {}
    for n in model.nodes:
        if not n.name:
            continue
        mod, leaf = _module_path(n.name)
        if n.op in ("Conv", "ConvTranspose"):
            put(mod + ".weight", model.tensor(n.inputs[1]))
            if len(n.inputs) > 2 and n.inputs[2]:
                put(mod + ".bias", model.tensor(n.inputs[2]))
            if "dilations" in n.attrs:
                ints[mod + ".dilation"] = int(n.attrs["dilations"][0])
            if n.op == "ConvTranspose":
                ints[mod + ".stride"] = int(n.attrs.get("strides", [1])[0])
                ints[mod + ".pad"] = int(n.attrs.get("pads", [0, 0])[0])
            ints[mod + ".group"] = int(n.attrs.get("group", 1))
        elif n.op == "Gather" and n.inputs and n.inputs[0] in model.init and \
                model.init[n.inputs[0]] is not None and model.init[n.inputs[0]].ndim == 2 and \
                model.init[n.inputs[0]].dtype == np.float32:
            put(mod + ".weight", model.init[n.inputs[0]])
        elif n.op in ("Mul", "Add") and "norm" in mod:
            for i in n.inputs:
                t = model.init.get(i)
                if t is not None and t.ndim == 1 and t.dtype == np.float32:
                    put(mod + (".gamma" if n.op == "Mul" else ".beta"), t)
        elif n.op == "Pad" and "attn_layers" in mod:
            t = model.init.get(n.inputs[0])
            if t is not None and t.ndim == 3:
                k = pad_seen.get(mod, 0)
                put(mod + (".emb_rel_k" if k == 0 else ".emb_rel_v"), t)
                pad_seen[mod] = k + 1
        elif mod == "dp.flows.0" and n.op == "Sub":
            for i in n.inputs:
                if i in model.init:
                    put("dp.flows.0.m", model.init[i])
        elif mod == "dp.flows.0" and n.op == "Exp":
            for i in n.inputs:
                if i in model.init:  # exporter folded Neg(logs) into the initializer
                    put("dp.flows.0.logs", -model.init[i])
    # anything still only reachable by parameter name (e.g. rel-pos tables when T<=window at trace)
    for k, v in model.init.items():
        if v is not None and v.dtype == np.float32 and (k.endswith("emb_rel_k") or k.endswith("emb_rel_v")):
            put(k, v)
    return tensors, ints


class VitsOracle:
    def __init__(self, onnx_path, native=False):
        self.lib = load_lib(native)
        self.model = OnnxModel(onnx_path)
        self.input_names = list(self.model.inputs)
        self.meta = dict(self.model.meta)
        self.tensors, self.ints = resolve_weights(self.model)
        self.h = ctypes.c_void_p(self.lib.vo_new())
        for name, arr in self.tensors.items():
            dims = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            self.lib.vo_set_tensor(self.h, name.encode(), arr.ctypes.data_as(ctypes.c_void_p), arr.ndim, dims)
        for k, v in self.ints.items():
            self.lib.vo_set_int(self.h, k.encode(), int(v))
        self.inter_channels = self.tensors["enc_p.proj.weight"].shape[0] // 2
        self.n_speakers = self.tensors["emb_g.weight"].shape[0] if "emb_g.weight" in self.tensors else 1

    def __del__(self):
        try:
            self.lib.vo_free(self.h)
        except Exception:
            pass

    def _result(self, name):
        nd = ctypes.c_int()
        dims = (ctypes.c_int64 * 4)()
        p = self.lib.vo_result(self.h, name.encode(), ctypes.byref(nd), dims)
        if not p:
            return None
        shape = tuple(dims[i] for i in range(nd.value))
        n = int(np.prod(shape)) if shape else 1
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).copy()

    def infer(self, ids, lens, scales, sid=None, noise_dp=None, noise_z=None):
        ids = np.ascontiguousarray(ids, np.int64)
        lens = np.ascontiguousarray(lens, np.int64)
        scales = np.ascontiguousarray(scales, np.float32)
        B, T = ids.shape
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        sid = None if sid is None else np.ascontiguousarray(sid, np.int64)
        noise_dp = None if noise_dp is None else np.ascontiguousarray(noise_dp, np.float32)
        noise_z = None if noise_z is None else np.ascontiguousarray(noise_z, np.float32)
        stride = 0 if noise_z is None else noise_z.shape[2]
        rc = self.lib.vo_infer(self.h, vp(ids), vp(lens), B, T, vp(scales), vp(sid), vp(noise_dp), vp(noise_z), stride)
        if rc != 0:
            raise RuntimeError(self.lib.vo_error(self.h).decode())
        out = {k: self._result(k) for k in
               ("emb", "x", "m_p", "logs_p", "logw", "w_ceil", "y_lengths", "z_p", "z", "output")}
        out["y_lengths"] = out["y_lengths"].astype(np.int64)
        return out

    def vocoder(self, z, sid=None):
        z = np.ascontiguousarray(z, np.float32)
        B, C, F = z.shape
        sid = None if sid is None else np.ascontiguousarray(sid, np.int64)
        rc = self.lib.vo_vocoder(self.h, z.ctypes.data_as(ctypes.c_void_p), B, C, F,
                                 None if sid is None else sid.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(self.lib.vo_error(self.h).decode())
        return self._result("output")


def conv1d(x, w, bias=None, dil=1, pad_l=0, pad_r=0, groups=1, native=False):
    lib = load_lib(native)
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    To = T + pad_l + pad_r - dil * (K - 1)
    out = np.empty((B, Cout, To), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib.vo_conv1d(x.ctypes.data, B, Cin, T, w.ctypes.data, None if b is None else b.ctypes.data, Cout, K, dil,
                  pad_l, pad_r, groups, out.ctypes.data)
    return out


def conv_transpose1d(x, w, bias, stride, pad, native=False):
    lib = load_lib(native)
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    _, Cout, K = w.shape
    To = (T - 1) * stride - 2 * pad + K
    out = np.empty((B, Cout, To), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib.vo_conv_transpose1d(x.ctypes.data, B, Cin, T, w.ctypes.data, None if b is None else b.ctypes.data, Cout, K,
                            stride, pad, out.ctypes.data)
    return out


def attention_core(qkv, n_heads, rel_k, rel_v, lens, native=False):
    """qkv [B,3C,T] -> [B,C,T] (attentions.py:225-272 on projected q,k,v)."""
    lib = load_lib(native)
    qkv = np.ascontiguousarray(qkv, np.float32)
    B, C3, T = qkv.shape
    C = C3 // 3
    q = np.ascontiguousarray(qkv[:, :C]); k = np.ascontiguousarray(qkv[:, C:2 * C]); v = np.ascontiguousarray(qkv[:, 2 * C:])
    rel_k = np.ascontiguousarray(rel_k, np.float32); rel_v = np.ascontiguousarray(rel_v, np.float32)
    lens = np.ascontiguousarray(lens, np.int64)
    out = np.empty((B, C, T), np.float32)
    lib.vo_attention_core(q.ctypes.data, k.ctypes.data, v.ctypes.data, rel_k.ctypes.data, rel_v.ctypes.data, B, C, T,
                          C // n_heads, (rel_k.shape[0] - 1) // 2, lens.ctypes.data, out.ctypes.data)
    return out
