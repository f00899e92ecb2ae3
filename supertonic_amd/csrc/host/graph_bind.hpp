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
    // a fused projection (one MatMul for q|k|v or k|v, split afterwards): this tensor is rows [row0, row0 + rows) of the
    // initializer's `rows_total` canonical rows; rows_total == 0: the whole initializer
    int row0 = 0, rows_total = 0, nrows = 0;
    // a one-element initializer standing for `broadcast` equal values (the bias of a one-channel ConvTranspose wave head: every sample of the
    // frame gets the same bias); 0: no
    int broadcast = 0;
    std::string from;                 // "<file>: node #i <op> '<name>' input '<initializer>'"
};

struct Result {
    stn_arch arch;
    std::map<std::string, Bound> tensors;  // canonical name -> initializer
    std::string notes;                     // remarks that do not stop a load (e.g. the GELU form the graphs spell out)
    std::string gelu;                      // "erf", "tanh", "op" (a Gelu node), "" (none seen): how the graphs write the activation
};

// `base`: the descriptor as tts.json (and the defaults) give it; the graphs override what their shapes determine and must agree
// with what tts.json states.  Throws std::runtime_error with the diff described above.
// Variants of the same layout that exporters produce are accepted: LayerNormalization as one node or decomposed
// (ReduceMean / Sub / Pow / ReduceMean / Add / Sqrt / Div / Mul gamma / Add beta), q / k / v as separate projections or fused
// (one projection of 3C or 2C rows split afterwards), pointwise projections as Conv k=1, MatMul(+Add) in either operand order
// with Transposes around it, or Gemm with either transB, GELU as a Gelu node or spelled out with Erf or Tanh.
// heads_explicit: base's *_heads fields were stated by the caller (stn_weight_map.json "arch"); otherwise a head count that the
// graphs do not carry (no [batch, length, heads, head_dim] Reshape constant inside an attention block) is an ERROR, not a default.
Result bind(const stn_arch& base, const onnx::Model& dp, const onnx::Model& te, const onnx::Model& ve, const onnx::Model& vo,
            bool heads_explicit = false);

// The descriptor as tts.json states it: the four fields every host reads (/root/reference/cpp/helper.cpp:811-815) and the style /
// projection dims of /root/reference/go/helper.go:45-78 when present; everything else at stn_arch_default's values.
stn_arch arch_from_config(const std::string& tts_json_path);

// bind() over the four graph files of `dir` (+ check_io_names), as JSON: {"arch": {...}, "tensors": {canonical: {"from": "...",
// "transpose": bool}}, "notes": "..."} — what stn_load_dir would load, without a device.
// An optional stn_weight_map.json WITHOUT a "tensors" table only states descriptor fields ("arch": {"ve_heads": 4, ...}) for the walk.
std::string bind_dir_json(const std::string& dir);
// {"arch": {...}} of <dir>/stn_weight_map.json applied to `a` when the file exists and has no "tensors" table; returns whether it did
bool apply_arch_overrides(const std::string& dir, stn_arch& a);

// Graph input / output names every host feeds and fetches (/root/reference/cpp/helper.cpp:547-672); throws naming the difference.
void check_io_names(const onnx::Model& m, const std::string& file, const std::vector<std::string>& inputs,
                    const std::vector<std::string>& outputs);

void check_all_io_names(const onnx::Model& dp, const onnx::Model& te, const onnx::Model& ve, const onnx::Model& vo);

// Fetch one bound tensor as canonical row-major [rows][cols] fp32 (transposing / zero-filling as bound); checks the element count
// and, for matrices, the stored dims against rows x cols.
std::vector<float> fetch(const Bound& b, const std::string& canonical, int rows, int cols);
// The same values without a descriptor (host tools / tests): canonical orientation and row block are taken from the initializer's own
// dims ([out][in...] or, transposed, [in][out]); an unbound tensor (zeros) comes back empty.
std::vector<float> fetch_canonical(const Bound& b, const std::string& canonical);
// bind() over <dir> as bind_dir_json does, returning one canonical tensor's values
std::vector<float> bound_tensor_of_dir(const std::string& dir, const std::string& canonical);

}  // namespace graphbind
}  // namespace stn
