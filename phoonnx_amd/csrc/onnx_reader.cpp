// onnx_reader.cpp — protobuf wire walker (varint / length-delimited / fixed32 / fixed64).
// Field numbers from onnx.proto3; only what export_onnx.py's files contain is decoded.
#include "onnx_reader.hpp"

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace vitsmi {
namespace {

struct Span {
    const uint8_t *p, *e;
};

struct Field {
    uint32_t num = 0;
    int wire = -1;
    uint64_t val = 0;              // wire 0/1/5
    Span sub{nullptr, nullptr};    // wire 2
};

// A string / bytes / sub-message field must arrive length-delimited: a damaged or hostile file can send the same
// field number with wire type 0/1/5, in which case `sub` holds nothing of this field (never read it then).
const Span &ld(const Field &f, const char *what) {
    if (f.wire != 2) throw std::runtime_error(std::string("field '") + what + "' is not length-delimited");
    return f.sub;
}
uint64_t scalar(const Field &f, const char *what) {
    if (f.wire == 2) throw std::runtime_error(std::string("field '") + what + "' is not a scalar");
    return f.val;
}

uint64_t varint(Span &s) {
    uint64_t r = 0;
    int shift = 0;
    while (true) {
        if (s.p >= s.e) throw std::runtime_error("truncated varint");
        uint8_t b = *s.p++;
        r |= uint64_t(b & 0x7F) << shift;
        if (!(b & 0x80)) return r;
        shift += 7;
        if (shift > 63) throw std::runtime_error("varint too long");
    }
}

bool next(Span &s, Field &f) {
    if (s.p >= s.e) return false;
    uint64_t key = varint(s);
    f.num = uint32_t(key >> 3);
    f.wire = int(key & 7);
    f.val = 0;
    f.sub = {nullptr, nullptr};
    switch (f.wire) {
        case 0: f.val = varint(s); break;
        case 1:
            if (s.e - s.p < 8) throw std::runtime_error("truncated fixed64");
            std::memcpy(&f.val, s.p, 8);
            s.p += 8;
            break;
        case 5: {
            if (s.e - s.p < 4) throw std::runtime_error("truncated fixed32");
            uint32_t v;
            std::memcpy(&v, s.p, 4);
            f.val = v;
            s.p += 4;
            break;
        }
        case 2: {
            uint64_t n = varint(s);
            if (uint64_t(s.e - s.p) < n) throw std::runtime_error("truncated field");
            f.sub = {s.p, s.p + n};
            s.p += n;
            break;
        }
        default: throw std::runtime_error("unsupported protobuf wire type");
    }
    return true;
}

std::string str(const Span &s) { return std::string(reinterpret_cast<const char *>(s.p), size_t(s.e - s.p)); }

void ints_of(const Field &f, std::vector<int64_t> &out) {
    if (f.wire == 2) {
        Span s = f.sub;
        while (s.p < s.e) out.push_back(int64_t(varint(s)));
    } else {
        out.push_back(int64_t(f.val));
    }
}

OnnxTensor tensor(Span s) {
    OnnxTensor t;
    Field f;
    while (next(s, f)) {
        switch (f.num) {
            case 1: ints_of(f, t.dims); break;
            case 2: t.dtype = int(scalar(f, "TensorProto.data_type")); break;
            case 8: t.name = str(ld(f, "TensorProto.name")); break;
            case 9: {
                const Span &r = ld(f, "TensorProto.raw_data");
                t.raw = r.p;
                t.raw_bytes = size_t(r.e - r.p);
                break;
            }
            case 4:  // float_data, packed or not
                if (f.wire == 2) {
                    size_t n = size_t(f.sub.e - f.sub.p) / 4;
                    size_t o = t.f32.size();
                    t.f32.resize(o + n);
                    std::memcpy(t.f32.data() + o, f.sub.p, n * 4);
                } else {
                    uint32_t v = uint32_t(f.val);
                    float x;
                    std::memcpy(&x, &v, 4);
                    t.f32.push_back(x);
                }
                break;
            default: break;
        }
    }
    // raw_data sits at an arbitrary byte offset of the file: a float view of it needs 4-byte alignment
    if (t.dtype == 1 && t.raw && (reinterpret_cast<uintptr_t>(t.raw) & 3u)) {
        t.f32.resize(t.raw_bytes / 4);
        std::memcpy(t.f32.data(), t.raw, t.f32.size() * 4);
        t.raw = nullptr;
    }
    return t;
}

OnnxNode node(Span s) {
    OnnxNode n;
    Field f;
    while (next(s, f)) {
        switch (f.num) {
            case 1: n.inputs.push_back(str(ld(f, "NodeProto.input"))); break;
            case 2: n.outputs.push_back(str(ld(f, "NodeProto.output"))); break;
            case 3: n.name = str(ld(f, "NodeProto.name")); break;
            case 4: n.op = str(ld(f, "NodeProto.op_type")); break;
            case 5: {  // AttributeProto
                Span a = ld(f, "NodeProto.attribute");
                Field g;
                std::string an;
                std::vector<int64_t> iv;
                bool has = false, hasf = false;
                float fv = 0.f;
                while (next(a, g)) {
                    if (g.num == 1) an = str(ld(g, "AttributeProto.name"));
                    else if (g.num == 2 && g.wire == 5) {  // AttributeProto.f (fixed32)
                        const uint32_t bits = uint32_t(g.val);
                        std::memcpy(&fv, &bits, 4);
                        hasf = true;
                    } else if (g.num == 3) { iv.push_back(int64_t(scalar(g, "AttributeProto.i"))); has = true; }
                    else if (g.num == 8) { ints_of(g, iv); has = true; }
                }
                if (has) n.ints[an] = iv;
                if (hasf) n.floats[an] = fv;
                break;
            }
            default: break;
        }
    }
    return n;
}

}  // namespace

std::string OnnxModel::load(const std::string &path) {
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) return "cannot open " + path;
    std::fseek(fp, 0, SEEK_END);
    long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (sz <= 0) {
        std::fclose(fp);
        return "empty file " + path;
    }
    buf.resize(size_t(sz));
    size_t got = std::fread(buf.data(), 1, size_t(sz), fp);
    std::fclose(fp);
    if (got != size_t(sz)) return "short read on " + path;
    try {
        Span top{buf.data(), buf.data() + buf.size()};
        Field f;
        Span graph{nullptr, nullptr};
        while (next(top, f)) {
            if (f.num == 7 && f.wire == 2) graph = f.sub;
            else if (f.num == 14 && f.wire == 2) {
                Span s = f.sub;
                Field g;
                std::string k, v;
                while (next(s, g)) {
                    if (g.num == 1) k = str(ld(g, "metadata key"));
                    else if (g.num == 2) v = str(ld(g, "metadata value"));
                }
                meta[k] = v;
            } else if (f.num == 8 && f.wire == 2) {
                Span s = f.sub;
                Field g;
                while (next(s, g))
                    if (g.num == 2 && g.wire != 2) opset = int64_t(g.val);
            }
        }
        if (!graph.p) return "no GraphProto in " + path;
        std::vector<std::string> all_inputs;
        while (next(graph, f)) {
            if (f.wire != 2) continue;
            if (f.num == 1) nodes.push_back(node(f.sub));
            else if (f.num == 5) {
                OnnxTensor t = tensor(f.sub);
                init[t.name] = std::move(t);
            } else if (f.num == 11 || f.num == 12) {
                Span s = f.sub;
                Field g;
                while (next(s, g))
                    if (g.num == 1 && g.wire == 2) (f.num == 11 ? all_inputs : outputs).push_back(str(g.sub));
            }
        }
        for (auto &n : all_inputs)
            if (!init.count(n)) inputs.push_back(n);
        for (const auto &n : nodes)
            if (n.op == "Identity" && !n.inputs.empty() && !n.outputs.empty() && !init.count(n.outputs[0]))
                alias[n.outputs[0]] = n.inputs[0];
    } catch (const std::exception &e) {
        return std::string("malformed onnx file: ") + e.what();
    }
    return "";
}

}  // namespace vitsmi
