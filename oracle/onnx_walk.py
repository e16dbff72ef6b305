"""Pure-Python protobuf walker for `.onnx` files.  TEST INFRASTRUCTURE (oracle side).

Independent of the product's C++ reader (phoonnx_amd/csrc/onnx_reader.cpp) on
purpose: the two are cross-checked against each other in tests/.

Only the wire features the PyTorch exporter emits are handled (wire types 0, 1,
2, 5; packed or unpacked repeated scalars).  Field numbers follow onnx.proto3:
ModelProto{ir_version=1, opset_import=8, graph=7, metadata_props=14},
GraphProto{node=1, name=2, initializer=5, input=11, output=12},
NodeProto{input=1, output=2, name=3, op_type=4, attribute=5},
AttributeProto{name=1, f=2, i=3, s=4, t=5, floats=7, ints=8, type=20},
TensorProto{dims=1, data_type=2, float_data=4, int32_data=5, int64_data=7,
name=8, raw_data=9}, ValueInfoProto{name=1}, StringStringEntryProto{key=1,value=2}.
"""
import struct

import numpy as np


def _varint(buf, pos):
    r = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def fields(buf, start=0, end=None):
    """Yield (field_number, wire_type, value) where value is an int for wire 0/1/5
    (raw bits for 1/5) and a (start, end) span for wire 2."""
    pos = start
    end = len(buf) if end is None else end
    while pos < end:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
            yield fn, wt, v
        elif wt == 1:
            yield fn, wt, struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 5:
            yield fn, wt, struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            yield fn, wt, (pos, pos + ln)
            pos += ln
        else:
            raise ValueError(f"unsupported wire type {wt} at {pos}")


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(buf, span):
    pos, end = span
    out = []
    while pos < end:
        v, pos = _varint(buf, pos)
        out.append(_signed(v))
    return out


_DT = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64, 2: np.uint8, 3: np.int8}


def parse_tensor(buf, span):
    dims, dt, name, raw = [], 0, "", None
    f32, i64, i32 = [], [], []
    for fn, wt, v in fields(buf, *span):
        if fn == 1:
            dims += _packed_ints(buf, v) if wt == 2 else [_signed(v)]
        elif fn == 2:
            dt = v
        elif fn == 8:
            name = bytes(buf[v[0]:v[1]]).decode()
        elif fn == 9:
            raw = v
        elif fn == 4:
            if wt == 2:
                f32 += list(np.frombuffer(buf, np.float32, (v[1] - v[0]) // 4, v[0]))
            else:
                f32.append(struct.unpack("<f", struct.pack("<I", v))[0])
        elif fn == 7:
            i64 += _packed_ints(buf, v) if wt == 2 else [_signed(v)]
        elif fn == 5:
            i32 += _packed_ints(buf, v) if wt == 2 else [_signed(v)]
    npdt = _DT.get(dt)
    if npdt is None:
        return name, None
    if raw is not None:
        arr = np.frombuffer(buf, npdt, (raw[1] - raw[0]) // np.dtype(npdt).itemsize, raw[0])
    elif f32:
        arr = np.asarray(f32, np.float32)
    elif i64:
        arr = np.asarray(i64, np.int64)
    elif i32:
        arr = np.asarray(i32, npdt)
    else:
        arr = np.zeros(0, npdt)
    return name, arr.reshape(dims) if dims or arr.size == 1 else arr


def parse_attr(buf, span):
    name, val = "", None
    ints, floats = [], []
    for fn, wt, v in fields(buf, *span):
        if fn == 1:
            name = bytes(buf[v[0]:v[1]]).decode()
        elif fn == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fn == 3:
            val = _signed(v)
        elif fn == 4:
            val = bytes(buf[v[0]:v[1]])
        elif fn == 5:
            val = parse_tensor(buf, v)[1]
        elif fn == 8:
            ints += _packed_ints(buf, v) if wt == 2 else [_signed(v)]
        elif fn == 7:
            if wt == 2:
                floats += list(np.frombuffer(buf, np.float32, (v[1] - v[0]) // 4, v[0]))
            else:
                floats.append(struct.unpack("<f", struct.pack("<I", v))[0])
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


class Node:
    __slots__ = ("name", "op", "inputs", "outputs", "attrs")

    def __repr__(self):
        return f"Node({self.op} {self.name} in={self.inputs} out={self.outputs} {self.attrs})"


def parse_node(buf, span):
    n = Node()
    n.name, n.op, n.inputs, n.outputs, n.attrs = "", "", [], [], {}
    for fn, wt, v in fields(buf, *span):
        if fn == 1:
            n.inputs.append(bytes(buf[v[0]:v[1]]).decode())
        elif fn == 2:
            n.outputs.append(bytes(buf[v[0]:v[1]]).decode())
        elif fn == 3:
            n.name = bytes(buf[v[0]:v[1]]).decode()
        elif fn == 4:
            n.op = bytes(buf[v[0]:v[1]]).decode()
        elif fn == 5:
            k, val = parse_attr(buf, v)
            n.attrs[k] = val
    return n


class OnnxModel:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = memoryview(f.read())
        buf = self.buf
        self.nodes, self.init, self.inputs, self.outputs, self.meta = [], {}, [], [], {}
        self.opset = None
        graph = None
        for fn, wt, v in fields(buf):
            if fn == 7:
                graph = v
            elif fn == 14:
                k = val = ""
                for f2, w2, v2 in fields(buf, *v):
                    if f2 == 1:
                        k = bytes(buf[v2[0]:v2[1]]).decode()
                    elif f2 == 2:
                        val = bytes(buf[v2[0]:v2[1]]).decode()
                self.meta[k] = val
            elif fn == 8:
                for f2, w2, v2 in fields(buf, *v):
                    if f2 == 2:
                        self.opset = v2
        assert graph is not None, "no graph in model"
        for fn, wt, v in fields(buf, *graph):
            if fn == 1:
                self.nodes.append(parse_node(buf, v))
            elif fn == 5:
                name, arr = parse_tensor(buf, v)
                self.init[name] = arr
            elif fn in (11, 12):
                for f2, w2, v2 in fields(buf, *v):
                    if f2 == 1:
                        (self.inputs if fn == 11 else self.outputs).append(
                            bytes(buf[v2[0]:v2[1]]).decode())
        # Constant nodes carry tensors too (exporter sometimes folds params there)
        self.const = {}
        for n in self.nodes:
            if n.op == "Constant" and "value" in n.attrs and n.attrs["value"] is not None:
                self.const[n.outputs[0]] = n.attrs["value"]
        # Identity nodes: the exporter de-duplicates initializers with identical bytes (e.g. every layer-norm weight of a
        # freshly initialised model) and re-introduces the other names as Identity(kept name)
        for n in self.nodes:
            if n.op == "Identity" and n.inputs and n.outputs and n.inputs[0] in self.init and n.outputs[0] not in self.init:
                self.init[n.outputs[0]] = self.init[n.inputs[0]]
        self.inputs = [i for i in self.inputs if i not in self.init]

    def tensor(self, name):
        if name in self.init:
            return self.init[name]
        return self.const.get(name)
