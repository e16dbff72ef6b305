#!/usr/bin/env python3
"""Rewrite an exported VITS `.onnx` so that its nodes are named the way older (Piper-era) torch exporters named them -
`<OpType>_<n>` - instead of by module path (`/flow/flows.6/enc/in_layers.0/Conv`).  Nothing else changes: same nodes in
the same order, same initializers, same bytes everywhere but in NodeProto.name.  Used by the tests of the
structure-keyed weight resolution (phoonnx_amd/csrc/model.cpp structural_paths): the renamed file must load to the same
weight arena and render the same waveform as the original.

    python tools/rename_nodes.py in.onnx out.onnx [--strip]      (--strip: empty names instead of <op>_<n>)

Pure protobuf wire-format surgery (ModelProto.graph = 7, GraphProto.node = 1, NodeProto.name = 3, op_type = 4); no onnx
package needed.
"""
import sys


def _varint(buf, pos):
    r = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _enc(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf, start, end):
    """(field number, wire type, start of the whole field, payload start, payload end)"""
    pos = start
    while pos < end:
        f0 = pos
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            _, p2 = _varint(buf, pos)
            yield fn, wt, f0, pos, p2
            pos = p2
        elif wt == 1:
            yield fn, wt, f0, pos, pos + 8
            pos += 8
        elif wt == 5:
            yield fn, wt, f0, pos, pos + 4
            pos += 4
        elif wt == 2:
            ln, p1 = _varint(buf, pos)
            yield fn, wt, f0, p1, p1 + ln
            pos = p1 + ln
        else:
            raise ValueError(f"unsupported wire type {wt}")


def _ld(fn, payload):
    return _enc((fn << 3) | 2) + _enc(len(payload)) + payload


def rename(src: bytes, strip=False) -> bytes:
    out = bytearray()
    counter = {}
    for fn, wt, f0, a, b in _fields(src, 0, len(src)):
        if fn != 7 or wt != 2:
            out += src[f0:b]
            continue
        g = bytearray()
        for gfn, gwt, g0, ga, gb in _fields(src, a, b):
            if gfn != 1 or gwt != 2:
                g += src[g0:gb]
                continue
            op = ""
            for nfn, nwt, n0, na, nb in _fields(src, ga, gb):
                if nfn == 4 and nwt == 2:
                    op = bytes(src[na:nb]).decode()
            k = counter.get(op, 0)
            counter[op] = k + 1
            node = bytearray()
            named = False
            for nfn, nwt, n0, na, nb in _fields(src, ga, gb):
                if nfn == 3 and nwt == 2:
                    named = True
                    if not strip:
                        node += _ld(3, f"{op}_{k}".encode())
                else:
                    node += src[n0:nb]
            if not named and not strip:
                node += _ld(3, f"{op}_{k}".encode())
            g += _ld(1, bytes(node))
        out += _ld(7, bytes(g))
    return bytes(out)


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    data = rename(open(args[0], "rb").read(), strip="--strip" in sys.argv)
    open(args[1], "wb").write(data)
    print(f"{args[1]}: {len(data)} bytes")
