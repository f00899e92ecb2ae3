#include "onnx_reader.hpp"

#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace stn {
namespace onnx {

namespace {
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) throw std::runtime_error("ONNX: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        throw std::runtime_error("ONNX: varint too long");
    }
    Reader sub() {  // length-delimited field
        const uint64_t n = varint();
        if (n > (uint64_t)(end - p)) throw std::runtime_error("ONNX: length-delimited field overruns the buffer");
        Reader r{p, p + n};
        p += n;
        return r;
    }
    void skip(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: need(8); p += 8; break;
            case 2: sub(); break;
            case 5: need(4); p += 4; break;
            default: throw std::runtime_error("ONNX: unsupported wire type " + std::to_string(wire));
        }
    }
    void need(size_t n) const { if ((size_t)(end - p) < n) throw std::runtime_error("ONNX: truncated field"); }
    std::string str() { Reader r = sub(); return std::string(reinterpret_cast<const char*>(r.p), (size_t)(r.end - r.p)); }
};

template <typename T, typename F>
void repeated_scalar(Reader& r, int wire, std::vector<T>& out, F decode_one, size_t fixed) {
    // proto3 packs repeated scalars (wire 2); older writers emit one element per tag
    if (wire == 2) {
        Reader s = r.sub();
        while (!s.done()) out.push_back(decode_one(s));
    } else {
        (void)fixed;
        out.push_back(decode_one(r));
    }
}

Tensor parse_tensor(Reader r) {
    Tensor t;
    while (!r.done()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        switch (field) {
            case 1: repeated_scalar(r, wire, t.dims, [](Reader& s) { return (int64_t)s.varint(); }, 0); break;
            case 2: t.data_type = (int)r.varint(); break;
            case 4: repeated_scalar(r, wire, t.float_data, [](Reader& s) { s.need(4); float f; std::memcpy(&f, s.p, 4); s.p += 4; return f; }, 4); break;
            case 5: repeated_scalar(r, wire, t.int32_data, [](Reader& s) { return (int32_t)s.varint(); }, 0); break;
            case 7: repeated_scalar(r, wire, t.int64_data, [](Reader& s) { return (int64_t)s.varint(); }, 0); break;
            case 8: t.name = r.str(); break;
            case 9: { Reader s = r.sub(); t.raw = s.p; t.raw_size = (size_t)(s.end - s.p); break; }
            case 14: t.external = r.varint() == 1; break;
            default: r.skip(wire);
        }
    }
    return t;
}
// AttributeProto{name = 1, i = 3, s = 4, t = 5, ints = 8}: integers into n.ints, strings into n.strs, a tensor (Constant's "value") into *tensor_out
void parse_attribute(Reader r, Node& n, std::vector<Tensor>* tensor_out) {
    std::string name;
    std::vector<int64_t> vals;
    bool have = false;
    Tensor t;
    bool have_t = false, have_s = false;
    std::string sval;
    while (!r.done()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 1 && wire == 2) name = r.str();
        else if (field == 3 && wire == 0) { vals.push_back((int64_t)r.varint()); have = true; }
        else if (field == 8) { repeated_scalar(r, wire, vals, [](Reader& s) { return (int64_t)s.varint(); }, 0); have = true; }
        else if (field == 5 && wire == 2) { t = parse_tensor(r.sub()); have_t = true; }
        else if (field == 4 && wire == 2) { sval = r.str(); have_s = true; }
        else r.skip(wire);
    }
    if (have) n.ints[name] = std::move(vals);
    if (have_s) n.strs[name] = std::move(sval);
    if (have_t && tensor_out && name == "value") tensor_out->push_back(std::move(t));
}
Node parse_node(Reader r, std::vector<Tensor>* constants) {
    Node n;
    std::vector<Tensor> vals;
    while (!r.done()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        switch (field) {
            case 1: n.inputs.push_back(r.str()); break;
            case 2: n.outputs.push_back(r.str()); break;
            case 3: n.name = r.str(); break;
            case 4: n.op_type = r.str(); break;
            case 5: if (wire == 2) parse_attribute(r.sub(), n, &vals); else r.skip(wire); break;
            default: r.skip(wire);
        }
    }
    if (n.op_type == "Constant" && !vals.empty() && !n.outputs.empty() && constants) {
        vals[0].name = n.outputs[0];
        constants->push_back(std::move(vals[0]));
    }
    return n;
}
std::string parse_value_info_name(Reader r) {
    std::string name;
    while (!r.done()) {
        const uint64_t key = r.varint();
        if ((key >> 3) == 1 && (key & 7) == 2) name = r.str();
        else r.skip((int)(key & 7));
    }
    return name;
}
void parse_graph(Reader r, Model& m) {
    while (!r.done()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (wire != 2) { r.skip(wire); continue; }
        switch (field) {
            case 1: m.nodes.push_back(parse_node(r.sub(), &m.initializers)); break;
            case 5: m.initializers.push_back(parse_tensor(r.sub())); break;
            case 11: m.inputs.push_back(parse_value_info_name(r.sub())); break;
            case 12: m.outputs.push_back(parse_value_info_name(r.sub())); break;
            default: r.skip(wire);
        }
    }
}
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else { int e = -1; uint32_t m2 = man; do { ++e; m2 <<= 1; } while (!(m2 & 0x400)); u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m2 & 0x3FF) << 13); }
    } else if (exp == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
}  // namespace

int64_t Tensor::numel() const {
    int64_t n = 1;
    for (int64_t d : dims) n *= d;
    return n;
}
const Tensor* Model::find(const std::string& name) const {
    for (const Tensor& t : initializers) if (t.name == name) return &t;
    return nullptr;
}

Model parse_bytes(std::shared_ptr<std::vector<uint8_t>> bytes) {
    Model m;
    m.bytes = bytes;
    Reader r{bytes->data(), bytes->data() + bytes->size()};
    bool saw_graph = false;
    while (!r.done()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 1 && wire == 0) m.ir_version = (int64_t)r.varint();
        else if (field == 2 && wire == 2) m.producer = r.str();
        else if (field == 7 && wire == 2) { parse_graph(r.sub(), m); saw_graph = true; }
        else r.skip(wire);
    }
    if (!saw_graph) throw std::runtime_error("ONNX: no graph in model");
    return m;
}
Model parse_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open " + path);
    auto buf = std::make_shared<std::vector<uint8_t>>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse_bytes(buf);
}

std::vector<float> to_float(const Tensor& t) {
    if (t.external) throw std::runtime_error("ONNX tensor " + t.name + ": external data is not supported");
    const size_t n = (size_t)t.numel();
    std::vector<float> out(n);
    auto need_raw = [&](size_t esz) { if (t.raw_size != n * esz) throw std::runtime_error("ONNX tensor " + t.name + ": raw_data size does not match dims"); };
    switch (t.data_type) {
        case FLOAT:
            if (t.raw) { need_raw(4); std::memcpy(out.data(), t.raw, n * 4); }
            else { if (t.float_data.size() != n) throw std::runtime_error("ONNX tensor " + t.name + ": float_data size mismatch"); out = t.float_data; }
            break;
        case DOUBLE: need_raw(8); for (size_t i = 0; i < n; ++i) { double d; std::memcpy(&d, t.raw + 8 * i, 8); out[i] = (float)d; } break;
        case FLOAT16:
            if (t.raw) { need_raw(2); for (size_t i = 0; i < n; ++i) { uint16_t h; std::memcpy(&h, t.raw + 2 * i, 2); out[i] = half_to_float(h); } }
            else { if (t.int32_data.size() != n) throw std::runtime_error("ONNX tensor " + t.name + ": int32_data size mismatch"); for (size_t i = 0; i < n; ++i) out[i] = half_to_float((uint16_t)t.int32_data[i]); }
            break;
        case BFLOAT16:
            need_raw(2);
            for (size_t i = 0; i < n; ++i) { uint16_t h; std::memcpy(&h, t.raw + 2 * i, 2); const uint32_t u = (uint32_t)h << 16; std::memcpy(&out[i], &u, 4); }
            break;
        case INT64:
            if (t.raw) { need_raw(8); for (size_t i = 0; i < n; ++i) { int64_t v; std::memcpy(&v, t.raw + 8 * i, 8); out[i] = (float)v; } }
            else { if (t.int64_data.size() != n) throw std::runtime_error("ONNX tensor " + t.name + ": int64_data size mismatch"); for (size_t i = 0; i < n; ++i) out[i] = (float)t.int64_data[i]; }
            break;
        case INT32:
            if (t.raw) { need_raw(4); for (size_t i = 0; i < n; ++i) { int32_t v; std::memcpy(&v, t.raw + 4 * i, 4); out[i] = (float)v; } }
            else { if (t.int32_data.size() != n) throw std::runtime_error("ONNX tensor " + t.name + ": int32_data size mismatch"); for (size_t i = 0; i < n; ++i) out[i] = (float)t.int32_data[i]; }
            break;
        default: throw std::runtime_error("ONNX tensor " + t.name + ": unsupported data_type " + std::to_string(t.data_type));
    }
    return out;
}

std::string summary_json(const Model& m) {
    auto esc = [](const std::string& s) { std::string o; for (char c : s) { if (c == '"' || c == '\\') o.push_back('\\'); o.push_back(c); } return o; };
    std::ostringstream o;
    o << "{\"ir_version\":" << m.ir_version << ",\"producer\":\"" << esc(m.producer) << "\",\"inputs\":[";
    for (size_t i = 0; i < m.inputs.size(); ++i) o << (i ? "," : "") << "\"" << esc(m.inputs[i]) << "\"";
    o << "],\"outputs\":[";
    for (size_t i = 0; i < m.outputs.size(); ++i) o << (i ? "," : "") << "\"" << esc(m.outputs[i]) << "\"";
    std::map<std::string, int> ops;
    for (const Node& n : m.nodes) ++ops[n.op_type];
    o << "],\"n_nodes\":" << m.nodes.size() << ",\"ops\":{";
    bool first = true;
    for (auto& kv : ops) { o << (first ? "" : ",") << "\"" << esc(kv.first) << "\":" << kv.second; first = false; }
    o << "},\"initializers\":[";
    int64_t params = 0;
    for (size_t i = 0; i < m.initializers.size(); ++i) {
        const Tensor& t = m.initializers[i];
        o << (i ? "," : "") << "{\"name\":\"" << esc(t.name) << "\",\"dtype\":" << t.data_type << ",\"dims\":[";
        for (size_t d = 0; d < t.dims.size(); ++d) o << (d ? "," : "") << t.dims[d];
        o << "]}";
        params += t.numel();
    }
    o << "],\"n_params\":" << params << "}";
    return o.str();
}

}  // namespace onnx
}  // namespace stn
