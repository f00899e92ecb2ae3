// graph_bind.hpp — bind the four ONNX graphs of an asset directory to the engine's canonical tensor list WITHOUT a manifest.
//
// The reference hands the files to ONNX Runtime, which executes whatever graph they hold (/root/reference/cpp/helper.cpp:784-795).
// This engine executes ONE layout — embedding / ConvNeXt blocks / attention blocks / projections, include/stn_arch.h — so loading
// means recognising that layout in the graph: the nodes that carry weights are read in graph (= topological = execution) order,
// classified by operator and weight shape (depthwise Conv, pointwise Conv / MatMul(+Add) / Gemm, LayerNormalization, layer-scale
// Mul, Gather), and parsed against the layout's grammar.  Widths, depths, kernel sizes, dilations and (from the Reshape shape
// constants inside an attention block) head counts come out of the shapes; the tensors are bound by position and role, never by
// name.  A graph that is not this layout fails with the first weighted node that does not fit, what was expected there and the
// descriptor derived up to that point.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../../include/stn_arch.h"
#include "onnx_reader.hpp"

namespace stn {
namespace graphbind {

struct Bound {
    const onnx::Tensor* t = nullptr;  // nullptr: the graph has no such tensor (a projection without bias) -> zeros
    bool transpose = false;           // stored [cols][rows] (MatMul / Gemm with transB = 0)
    std::string from;                 // "<file>: node #i <op> '<name>' input '<initializer>'"
};

struct Result {
    stn_arch arch;
    std::map<std::string, Bound> tensors;  // canonical name -> initializer
    std::string notes;                     // what could not be read from the graphs and was left at the descriptor's value
};

// `base`: the descriptor as tts.json (and the defaults) give it; the graphs override what their shapes determine and must agree
// with what tts.json states.  Throws std::runtime_error with the diff described above.
Result bind(const stn_arch& base, const onnx::Model& dp, const onnx::Model& te, const onnx::Model& ve, const onnx::Model& vo);

// The descriptor as tts.json states it: the four fields every host reads (/root/reference/cpp/helper.cpp:811-815) and the style /
// projection dims of /root/reference/go/helper.go:45-78 when present; everything else at stn_arch_default's values.
stn_arch arch_from_config(const std::string& tts_json_path);

// bind() over the four graph files of `dir` (+ check_io_names), as JSON: {"arch": {...}, "tensors": {canonical: {"from": "...",
// "transpose": bool}}, "notes": "..."} — what stn_load_dir would load, without a device.
std::string bind_dir_json(const std::string& dir);

// Graph input / output names every host feeds and fetches (/root/reference/cpp/helper.cpp:547-672); throws naming the difference.
void check_io_names(const onnx::Model& m, const std::string& file, const std::vector<std::string>& inputs,
                    const std::vector<std::string>& outputs);

void check_all_io_names(const onnx::Model& dp, const onnx::Model& te, const onnx::Model& ve, const onnx::Model& vo);

// Fetch one bound tensor as canonical row-major [rows][cols] fp32 (transposing / zero-filling as bound); checks the element count
// and, for matrices, the stored dims against rows x cols.
std::vector<float> fetch(const Bound& b, const std::string& canonical, int rows, int cols);

}  // namespace graphbind
}  // namespace stn
