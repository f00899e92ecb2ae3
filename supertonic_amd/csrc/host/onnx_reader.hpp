// onnx_reader.hpp — minimal reader for the parts of an ONNX ModelProto the engine needs: graph initializers (weights),
// nodes (op type, edges, integer attributes; the tensor of a Constant node joins the initializers under the node's output name) and
// graph input/output names.  Hand-written protobuf wire parsing; neither `onnx` nor
// `protobuf` C++ exist in this image.  Replaces what `Ort::Session(env, path, opts)` does with the file at model-load time
// (/root/reference/cpp/helper.cpp:776-795).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace stn {
namespace onnx {

enum DataType { FLOAT = 1, UINT8 = 2, INT8 = 3, INT32 = 6, INT64 = 7, FLOAT16 = 10, DOUBLE = 11, BFLOAT16 = 16 };

struct Tensor {
    std::string name;
    std::vector<int64_t> dims;
    int data_type = 0;
    const uint8_t* raw = nullptr;  // raw_data (little-endian), points into the file buffer
    size_t raw_size = 0;
    std::vector<float> float_data;    // field 4 (when raw_data is absent)
    std::vector<int64_t> int64_data;  // field 7
    std::vector<int32_t> int32_data;  // field 5 (also carries fp16/bf16 bit patterns)
    bool external = false;            // data_location == EXTERNAL: not supported
    int64_t numel() const;
};
struct Node {
    std::string op_type, name;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, std::vector<int64_t>> ints;  // INT and INTS attributes (group, dilations, kernel_shape, transB, axis ...)
    const std::vector<int64_t>* attr(const std::string& k) const { auto it = ints.find(k); return it == ints.end() ? nullptr : &it->second; }
    int64_t attr_i(const std::string& k, int64_t dflt) const { auto* v = attr(k); return v && !v->empty() ? (*v)[0] : dflt; }
    std::map<std::string, std::string> strs;           // STRING attributes (Gelu's "approximate")
    std::string attr_s(const std::string& k, const std::string& dflt) const { auto it = strs.find(k); return it == strs.end() ? dflt : it->second; }
};

struct Model {
    int64_t ir_version = 0;
    std::string producer;
    std::vector<Tensor> initializers;
    std::vector<Node> nodes;
    std::vector<std::string> inputs, outputs;
    std::shared_ptr<std::vector<uint8_t>> bytes;  // keeps Tensor::raw alive
    const Tensor* find(const std::string& name) const;
};

Model parse_file(const std::string& path);                // throws std::runtime_error ("Failed to open ..." / malformed)
Model parse_bytes(std::shared_ptr<std::vector<uint8_t>> bytes);
std::vector<float> to_float(const Tensor& t);             // FLOAT / FLOAT16 / BFLOAT16 / DOUBLE / INT64 / INT32 -> float32
std::string summary_json(const Model& m);                 // {"ir_version":..,"inputs":[..],"ops":{..},"initializers":[..]}

}  // namespace onnx
}  // namespace stn
