#include "graph_bind.hpp"

#include <fstream>

#include "json_min.hpp"

#include <algorithm>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace stn {
namespace graphbind {

namespace {

using onnx::Model;
using onnx::Node;
using onnx::Tensor;

enum Kind { EMBED, DWCONV, CONVK, LINEAR, LN, SCALE, OTHER };
const char* kind_name(Kind k) {
    static const char* n[] = {"embedding table (Gather)", "depthwise Conv", "k-tap Conv", "pointwise projection (Conv k=1 / MatMul / Gemm)",
                              "LayerNormalization", "per-channel scale (Mul)", "unrecognised weighted operator"};
    return n[k];
}

// one node that carries weights, in graph order
struct Tok {
    Kind kind = OTHER;
    int node = -1;
    std::string where;            // "node #i Op 'name'"
    const Tensor* w = nullptr;    // weight / scale / table
    const Tensor* b = nullptr;    // bias (LN: beta), may be absent
    int out = 0, in = 0, k = 1, dil = 1;
    bool transposed = false;      // LINEAR stored [in][out]
    int heads = 0;                // head count of the first 4-D Reshape constant between this token and the next one
    bool bias_bcast = false;      // the bias is one value for all `out` outputs (one-channel ConvTranspose head)
    std::string out_name;         // SCALE: the value it produces (a decomposed LayerNorm's beta Add refers to it)
    bool norm_scale = false;      // SCALE whose input is a normalised value (x - mean) / sqrt(var + eps): the gamma of a decomposed LayerNorm
    // LINEAR whose result is cut into equal parts by a Split (or by Slices) on the channel axis — a fused q|k|v / k|v projection:
    int split_parts = 0;          // number of parts (0: no Split / Slice consumes the projection)
    std::string split_where;      // the Split / Slice node(s)
    std::string split_roles;      // what each part feeds, in part order: 'q', 'k', 'v' (or '?': not traceable to the attention's MatMuls)
    std::string split_err;        // why the cut is not the row-block layout the binder assumes (axis, sizes); empty = fine
};

std::string dims_str(const Tensor* t) {
    if (!t) return "(none)";
    std::ostringstream o;
    o << "[";
    for (size_t i = 0; i < t->dims.size(); ++i) o << (i ? "," : "") << t->dims[i];
    o << "]";
    return o.str();
}
bool is_float(const Tensor* t) { return t && (t->data_type == onnx::FLOAT || t->data_type == onnx::FLOAT16 || t->data_type == onnx::BFLOAT16 || t->data_type == onnx::DOUBLE); }
// dims with the 1s dropped
std::vector<int64_t> squeezed(const Tensor* t) {
    std::vector<int64_t> d;
    for (int64_t v : t->dims) if (v != 1) d.push_back(v);
    return d;
}

std::vector<Tok> weighted_nodes(const Model& m, const std::string& file, std::string* gelu_form = nullptr) {
    std::unordered_map<std::string, const Tensor*> init;
    for (const Tensor& t : m.initializers) init[t.name] = &t;
    std::unordered_map<std::string, std::string> made_by;  // value name -> operator that produced it
    for (const Node& n : m.nodes) for (const std::string& o : n.outputs) made_by[o] = n.op_type;
    auto from_op = [&](const std::string& v) -> const std::string& { static const std::string none; auto it = made_by.find(v); return it == made_by.end() ? none : it->second; };
    std::unordered_map<std::string, int> scale_tok;  // value name -> index of the SCALE token that produced it
    auto get = [&](const Node& n, size_t i) -> const Tensor* {
        if (i >= n.inputs.size()) return nullptr;
        auto it = init.find(n.inputs[i]);
        return it == init.end() ? nullptr : it->second;
    };
    std::vector<Tok> toks;
    std::unordered_map<std::string, int> producer;  // value name -> index of the LINEAR token that produced it
    for (size_t ni = 0; ni < m.nodes.size(); ++ni) {
        const Node& n = m.nodes[ni];
        Tok t;
        t.node = (int)ni;
        t.where = file + ": node #" + std::to_string(ni) + " " + n.op_type + " '" + n.name + "'";
        bool emit = false;
        if (gelu_form) {  // how the graph writes the activation (Result::gelu)
            if (n.op_type == "Gelu") { if (gelu_form->empty()) *gelu_form = n.attr_s("approximate", "none") == "tanh" ? "tanh" : "op"; }
            else if (n.op_type == "Erf") *gelu_form = "erf";
            else if (n.op_type == "Tanh" && *gelu_form != "erf" && !n.outputs.empty()) {
                // only the Tanh of 0.5 x (1 + tanh(..)): its output goes into an Add whose result goes into a Mul (a Tanh anywhere else in
                // the graph — a gate, a bounded output — says nothing about the activation)
                bool in_gelu = false;
                for (const Node& add : m.nodes) {
                    if (add.op_type != "Add" || add.outputs.empty()) continue;
                    bool takes = false;
                    for (const std::string& in : add.inputs) takes = takes || in == n.outputs[0];
                    if (!takes) continue;
                    for (const Node& mul : m.nodes) {
                        if (mul.op_type != "Mul") continue;
                        for (const std::string& in : mul.inputs) in_gelu = in_gelu || in == add.outputs[0];
                    }
                }
                if (in_gelu) *gelu_form = "tanh";
            }
        }
        if (n.op_type == "Gather") {
            const Tensor* w = get(n, 0);
            if (is_float(w) && w->dims.size() == 2) { t.kind = EMBED; t.w = w; t.out = (int)w->dims[0]; t.in = (int)w->dims[1]; emit = true; }
        } else if (n.op_type == "Conv") {
            const Tensor* w = get(n, 1);
            if (!is_float(w) || w->dims.size() != 3) { t.kind = OTHER; t.w = w; emit = true; t.where += " (weight " + dims_str(w) + ": a 1-D convolution with a constant [Cout][Cin/group][k] weight is required)"; }
            else {
                const int64_t group = n.attr_i("group", 1), co = w->dims[0], cig = w->dims[1], k = w->dims[2];
                t.w = w; t.b = get(n, 2); t.out = (int)co; t.k = (int)k; t.dil = (int)n.attr_i("dilations", 1);
                if (group > 1) {
                    if (group == co && cig == 1) { t.kind = DWCONV; t.in = (int)co; }
                    else { t.kind = OTHER; t.where += " (group = " + std::to_string(group) + " with weight " + dims_str(w) + ": only depthwise grouping is part of the layout)"; }
                } else if (k == 1) { t.kind = LINEAR; t.in = (int)cig; }
                else { t.kind = CONVK; t.in = (int)cig; }
                emit = true;
            }
        } else if (n.op_type == "ConvTranspose") {
            // the wave head as a transposed 1-D convolution (north_star's spelling): weight [Cin][Cout / group][k].  With ONE output channel and
            // stride == kernel (no overlap, no padding) frame t writes samples t*k .. t*k + k - 1 = W^T x_t + b: the frame -> chunk projection the
            // engine's head kernel computes, weight [Cin][k] = the transposed linear form, the one bias value for every sample.
            const Tensor* w = get(n, 1);
            const auto all_are = [&](const char* key, int64_t want) { const auto* v = n.attr(key); if (!v) return true; for (int64_t x : *v) if (x != want) return false; return true; };
            if (is_float(w) && w->dims.size() == 3 && w->dims[1] == 1 && n.attr_i("group", 1) == 1 && n.attr_i("strides", 1) == w->dims[2] &&
                all_are("pads", 0) && all_are("dilations", 1) && all_are("output_padding", 0)) {
                t.kind = LINEAR; t.w = w; t.b = get(n, 2); t.in = (int)w->dims[0]; t.out = (int)w->dims[2]; t.transposed = true; t.bias_bcast = true; emit = true;
            } else {
                t.kind = OTHER; t.w = w; emit = true;
                t.where += " (weight " + dims_str(w) + ", stride " + std::to_string(n.attr_i("strides", 1)) +
                           ": only a one-channel transposed convolution with stride == kernel, no padding — a frame -> chunk projection — is part of the layout; "
                           "an overlapping one is a different vocoder head)";
            }
        } else if (n.op_type == "MatMul") {
            const Tensor *w1 = get(n, 1), *w0 = get(n, 0);
            if (is_float(w1) && w1->dims.size() == 2) { t.kind = LINEAR; t.w = w1; t.in = (int)w1->dims[0]; t.out = (int)w1->dims[1]; t.transposed = true; emit = true; }
            else if (is_float(w0) && w0->dims.size() == 2) { t.kind = LINEAR; t.w = w0; t.out = (int)w0->dims[0]; t.in = (int)w0->dims[1]; emit = true; }
        } else if (n.op_type == "Gemm") {
            const Tensor* w = get(n, 1);
            if (is_float(w) && w->dims.size() == 2) {
                const bool tb = n.attr_i("transB", 0) != 0;
                t.kind = LINEAR; t.w = w; t.b = get(n, 2); t.transposed = !tb;
                t.out = (int)(tb ? w->dims[0] : w->dims[1]); t.in = (int)(tb ? w->dims[1] : w->dims[0]);
                emit = true;
            }
        } else if (n.op_type == "LayerNormalization") {
            const Tensor* g = get(n, 1);
            if (is_float(g)) { t.kind = LN; t.w = g; t.b = get(n, 2); t.out = t.in = (int)g->numel(); emit = true; }
        } else if (n.op_type == "Add" || n.op_type == "Mul") {
            const Tensor* c = is_float(get(n, 1)) ? get(n, 1) : (is_float(get(n, 0)) ? get(n, 0) : nullptr);
            if (c && n.inputs.size() == 2) {
                const std::string& other = n.inputs[get(n, 1) == c ? 0 : 1];
                if (n.op_type == "Add") {  // MatMul + Add = projection with bias (a one-row projection has a one-element bias)
                    auto p = producer.find(other);
                    if (p != producer.end() && !toks[p->second].b && toks[p->second].out == (int)c->numel() && squeezed(c).size() <= 1) {
                        toks[p->second].b = c;
                        if (!n.outputs.empty()) producer[n.outputs[0]] = p->second;
                        continue;
                    }
                    // decomposed LayerNorm: ... Div -> Mul gamma -> Add beta.  The Mul was recorded as a normalising scale; this Add
                    // completes it into a LayerNorm token
                    auto sc = scale_tok.find(other);
                    if (sc != scale_tok.end() && toks[sc->second].norm_scale && toks[sc->second].kind == SCALE && toks[sc->second].out == (int)c->numel() &&
                        squeezed(c).size() <= 1) {
                        Tok& g = toks[sc->second];
                        g.kind = LN; g.b = c;
                        g.where += " + node #" + std::to_string(ni) + " Add (decomposed LayerNormalization)";
                        continue;
                    }
                }
                if (c->numel() > 1) {  // scalars (attention scale, epsilons) are not weights
                    if (squeezed(c).size() != 1) { t.kind = OTHER; t.w = c; emit = true; }
                    else if (n.op_type == "Add") { t.kind = OTHER; t.w = c; emit = true; t.where += " (a constant vector added to something that is not a bias-free projection of that width)"; }
                    else {
                        t.kind = SCALE; t.w = c; t.out = t.in = (int)c->numel(); emit = true;
                        // (x - mean) / sqrt(var + eps) arrives from a Div, or from a Mul by a Reciprocal / Rsqrt-style value
                        const std::string& src = from_op(other);
                        t.norm_scale = src == "Div" || src == "InstanceNormalization" || src == "MeanVarianceNormalization";
                        if (!n.outputs.empty()) t.out_name = n.outputs[0];
                    }
                }
            }
        } else if (n.op_type == "Reshape") {
            const Tensor* s = get(n, 1);
            if (s && s->data_type == onnx::INT64 && s->numel() == 4 && !toks.empty() && toks.back().heads == 0) {
                const std::vector<float> v = onnx::to_float(*s);
                if (v[2] > 0) toks.back().heads = (int)v[2];  // [batch, length, heads, head_dim]
            }
        } else {
            for (size_t i = 0; i < n.inputs.size(); ++i) {
                const Tensor* c = get(n, i);
                if (is_float(c) && c->numel() > 1) { t.kind = OTHER; t.w = c; emit = true; break; }
            }
        }
        if (!emit) continue;
        if (t.kind == LINEAR && !n.outputs.empty()) producer[n.outputs[0]] = (int)toks.size();
        if (t.kind == SCALE && !t.out_name.empty()) scale_tok[t.out_name] = (int)toks.size();
        toks.push_back(t);
    }
    // ---- fused projections: who cuts a projection's result, on which axis, into what sizes, and which attention operand each part becomes.
    // The binder assigns the row blocks of a 3C / 2C projection to q | k | v in part order: that is only right when the graph's own Split
    // (or Slices) cuts the LAST axis into equal blocks and hands them to the attention in that order.
    {
        std::unordered_map<std::string, std::vector<int>> cons;   // value -> nodes that read it
        std::unordered_map<std::string, int> made_at;             // value -> node that writes it
        for (size_t ni = 0; ni < m.nodes.size(); ++ni) {
            for (const std::string& in : m.nodes[ni].inputs) cons[in].push_back((int)ni);
            for (const std::string& o : m.nodes[ni].outputs) made_at[o] = (int)ni;
        }
        auto consumers = [&](const std::string& v) -> const std::vector<int>& { static const std::vector<int> none; auto it = cons.find(v); return it == cons.end() ? none : it->second; };
        auto is_act_matmul = [&](const Node& n) { return n.op_type == "MatMul" && n.inputs.size() == 2 && !init.count(n.inputs[0]) && !init.count(n.inputs[1]); };
        // nearest activation x activation MatMul downstream of a value, and the operand slot the value's lineage enters it through
        auto first_matmul = [&](const std::string& start, int* slot) -> int {
            std::vector<std::string> frontier{start};
            std::unordered_map<std::string, bool> seen;
            for (int visited = 0; !frontier.empty() && visited < 256;) {
                std::vector<std::string> next;
                for (const std::string& v : frontier) {
                    if (seen[v]) continue;
                    seen[v] = true; ++visited;
                    for (int ci : consumers(v)) {
                        const Node& c = m.nodes[ci];
                        if (is_act_matmul(c)) { *slot = c.inputs[0] == v ? 0 : 1; return ci; }
                        if (c.op_type == "Softmax") continue;  // (a part never reaches the attention through a Softmax)
                        for (const std::string& o : c.outputs) next.push_back(o);
                    }
                }
                frontier.swap(next);
            }
            return -1;
        };
        auto softmax_downstream = [&](int mm) {  // the scores MatMul: its result reaches a Softmax before any other MatMul
            std::vector<std::string> frontier(m.nodes[mm].outputs.begin(), m.nodes[mm].outputs.end());
            for (int depth = 0; depth < 8 && !frontier.empty(); ++depth) {
                std::vector<std::string> next;
                for (const std::string& v : frontier)
                    for (int ci : consumers(v)) {
                        if (m.nodes[ci].op_type == "Softmax") return true;
                        if (m.nodes[ci].op_type == "MatMul") continue;
                        for (const std::string& o : m.nodes[ci].outputs) next.push_back(o);
                    }
                frontier.swap(next);
            }
            return false;
        };
        auto softmax_upstream = [&](int mm, int slot) {  // the P V MatMul: operand `slot` comes out of a Softmax
            std::vector<std::string> frontier{m.nodes[mm].inputs[slot]};
            for (int depth = 0; depth < 6 && !frontier.empty(); ++depth) {
                std::vector<std::string> next;
                for (const std::string& v : frontier) {
                    auto it = made_at.find(v);
                    if (it == made_at.end()) continue;
                    const Node& pn = m.nodes[it->second];
                    if (pn.op_type == "Softmax") return true;
                    if (pn.op_type == "MatMul") continue;
                    for (const std::string& in : pn.inputs) next.push_back(in);
                }
                frontier.swap(next);
            }
            return false;
        };
        std::unordered_map<int, std::string> tok_value;  // LINEAR token -> the last value it was seen to produce (behind its bias Add)
        for (const auto& kv : producer) {
            auto it = tok_value.find(kv.second);
            if (it == tok_value.end() || made_at[kv.first] > made_at[it->second]) tok_value[kv.second] = kv.first;
        }
        for (auto& tv : tok_value) {
            Tok& t = toks[tv.first];
            std::string cur = tv.second;
            for (int hop = 0; hop < 4; ++hop) {  // Transpose / Cast / Identity between the projection and its cut
                const std::vector<int>& cs = consumers(cur);
                if (cs.size() != 1) break;
                const Node& c = m.nodes[cs[0]];
                if ((c.op_type != "Transpose" && c.op_type != "Cast" && c.op_type != "Identity") || c.outputs.empty()) break;
                cur = c.outputs[0];
            }
            struct Part { std::string value; int64_t start; };
            std::vector<Part> parts;
            const std::vector<int>& cs = consumers(cur);
            std::vector<int64_t> sizes;
            int64_t axis = -1;
            bool all_slices = cs.size() >= 2;
            for (int ci : cs) all_slices = all_slices && m.nodes[ci].op_type == "Slice";
            if (cs.size() == 1 && m.nodes[cs[0]].op_type == "Split") {
                const Node& sp = m.nodes[cs[0]];
                t.split_where = "node #" + std::to_string(cs[0]) + " Split '" + sp.name + "'";
                axis = sp.attr_i("axis", 0);
                if (const auto* a = sp.attr("split")) sizes = *a;
                else if (const Tensor* st = get(sp, 1)) { if (st->data_type == onnx::INT64) for (float f : onnx::to_float(*st)) sizes.push_back((int64_t)f); }
                int64_t at = 0;
                for (size_t j = 0; j < sp.outputs.size(); ++j) { parts.push_back({sp.outputs[j], at}); at += j < sizes.size() ? sizes[j] : 0; }
            } else if (all_slices) {
                t.split_where = std::to_string(cs.size()) + " Slice nodes behind " + t.where;
                for (int ci : cs) {
                    const Node& sl = m.nodes[ci];
                    const Tensor *st = get(sl, 1), *en = get(sl, 2), *ax = get(sl, 3);
                    if (!st || !en || sl.outputs.empty() || st->numel() != 1 || en->numel() != 1) { t.split_err = "node #" + std::to_string(ci) + " Slice '" + sl.name + "': starts / ends are not single constants"; break; }
                    const int64_t s0 = (int64_t)onnx::to_float(*st)[0], e0 = (int64_t)onnx::to_float(*en)[0];
                    if (ax && ax->numel() == 1) axis = (int64_t)onnx::to_float(*ax)[0];
                    parts.push_back({sl.outputs[0], s0});
                    sizes.push_back(e0 - s0);
                }
                // (parts in the order of their start offsets: that is the order of the row blocks)
                std::vector<size_t> ord(parts.size());
                for (size_t j = 0; j < ord.size(); ++j) ord[j] = j;
                std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return parts[a].start < parts[b].start; });
                std::vector<Part> p2; std::vector<int64_t> s2;
                for (size_t j : ord) { p2.push_back(parts[j]); if (j < sizes.size()) s2.push_back(sizes[j]); }
                parts.swap(p2); sizes.swap(s2);
            } else continue;
            t.split_parts = (int)parts.size();
            if (t.split_err.empty() && axis != -1 && axis != 2)  // activations here are [batch, length, channels]
                t.split_err = t.split_where + " cuts axis " + std::to_string(axis) + ", not the channel (last) axis";
            if (t.split_err.empty() && !sizes.empty())
                for (int64_t z : sizes)
                    if (z * t.split_parts != t.out) { t.split_err = t.split_where + " cuts the " + std::to_string(t.out) + " channels into unequal parts"; break; }
            for (const Part& pt : parts) {
                int slot = 0;
                const int mm = first_matmul(pt.value, &slot);
                char role = '?';
                if (mm >= 0) {
                    if (softmax_upstream(mm, 1 - slot)) role = 'v';
                    else if (softmax_downstream(mm)) role = slot == 0 ? 'q' : 'k';
                }
                t.split_roles.push_back(role);
            }
        }
    }
    // a normalising scale that no beta followed is a LayerNorm without bias (beta = zeros)
    for (Tok& t : toks)
        if (t.kind == SCALE && t.norm_scale) { t.kind = LN; t.where += " (decomposed LayerNormalization without beta)"; }
    return toks;
}

std::string arch_so_far(const stn_arch& a, const char* stage) {
    std::ostringstream o;
    const std::string s = stage;
    if (s == "dp") o << "dp_dim=" << a.dp_dim << " dp_hidden=" << a.dp_hidden << " dp_kernel=" << a.dp_kernel << " dp_conv_blocks=" << a.dp_conv_blocks;
    if (s == "te") o << "te_dim=" << a.te_dim << " te_hidden=" << a.te_hidden << " te_kernel=" << a.te_kernel << " te_conv_blocks=" << a.te_conv_blocks
                     << " te_attn_blocks=" << a.te_attn_blocks << " te_ffn=" << a.te_ffn << " te_style_blocks=" << a.te_style_blocks;
    if (s == "ve") o << "ve_dim=" << a.ve_dim << " ve_hidden=" << a.ve_hidden << " ve_kernel=" << a.ve_kernel << " ve_main_blocks=" << a.ve_main_blocks
                     << " ve_dilated=" << a.ve_dilated << " ve_tail_blocks=" << a.ve_tail_blocks << " ve_time_dim=" << a.ve_time_dim;
    if (s == "vo") o << "vo_dim=" << a.vo_dim << " vo_hidden=" << a.vo_hidden << " vo_kernel=" << a.vo_kernel << " vo_in_kernel=" << a.vo_in_kernel
                     << " vo_blocks=" << a.vo_blocks;
    return o.str();
}

// cursor over one graph's weighted nodes; every expect_* either binds canonical names or throws the diff
struct Parser {
    const std::vector<Tok>& t;
    const std::string file;
    const char* stage;
    Result& r;
    size_t i = 0;

    const Tok* peek(size_t ahead = 0) const { return i + ahead < t.size() ? &t[i + ahead] : nullptr; }
    bool next_is(Kind k, size_t ahead = 0) const { const Tok* p = peek(ahead); return p && p->kind == k; }

    [[noreturn]] void fail(const std::string& canonical, const std::string& want) const {
        std::ostringstream o;
        o << file << ": the graph is not the embedding / ConvNeXt / attention layout this engine runs. At weighted node " << i << " of " << t.size()
          << " the layout needs " << canonical << " = " << want << "; the graph has ";
        if (const Tok* p = peek()) {
            o << kind_name(p->kind) << " " << p->out << " <- " << p->in;
            if (p->kind == DWCONV || p->kind == CONVK) o << ", k = " << p->k << ", dilation " << p->dil;
            o << " (" << p->where << ", weight " << dims_str(p->w) << ")";
        } else o << "no further weighted node";
        o << ". Derived so far: " << arch_so_far(r.arch, stage);
        throw std::runtime_error(o.str());
    }
    void bind(const std::string& name, const Tensor* ten, bool transpose, const Tok& tok, int row0 = 0, int rows_total = 0, int rows_per_part = 0) {
        Bound b;
        b.t = ten; b.transpose = transpose; b.row0 = row0; b.rows_total = rows_total; b.nrows = rows_total ? rows_per_part : 0;
        b.from = tok.where + (ten ? " initializer '" + ten->name + "' " + dims_str(ten) : " (no such input: zeros)");
        if (rows_total) b.from += " rows " + std::to_string(row0) + ".. of " + std::to_string(rows_total);
        r.tensors[name] = b;
    }
    static std::string shape(int out, int in) { return (out < 0 ? std::string("?") : std::to_string(out)) + " <- " + (in < 0 ? std::string("?") : std::to_string(in)); }

    // out / in < 0: taken from the graph
    const Tok& linear(const std::string& name, int out, int in) {
        const Tok* p = peek();
        if (!p || p->kind != LINEAR || (out >= 0 && p->out != out) || (in >= 0 && p->in != in)) fail(name, std::string(kind_name(LINEAR)) + " " + shape(out, in));
        bind(name + ".w", p->w, p->transposed, *p);
        bind(name + ".b", p->b, false, *p);
        if (p->bias_bcast && p->b) r.tensors[name + ".b"].broadcast = p->out;
        ++i;
        return *p;
    }
    void ln(const std::string& name, int c) {
        const Tok* p = peek();
        if (!p || p->kind != LN || p->out != c) fail(name, std::string(kind_name(LN)) + " over " + std::to_string(c) + " channels");
        bind(name + ".g", p->w, false, *p);
        bind(name + ".b", p->b, false, *p);
        ++i;
    }
    // ConvNeXt block: depthwise conv -> LayerNorm -> pw1 -> (GELU) -> pw2 -> layer scale.  c / hid / k < 0: set from the graph.
    int convnext(const std::string& name, int& c, int& hid, int& k, int want_dil) {
        const Tok* p = peek();
        if (!p || p->kind != DWCONV || (c >= 0 && p->out != c) || (k >= 0 && p->k != k) || (want_dil > 0 && p->dil != want_dil))
            fail(name + ".dw", std::string(kind_name(DWCONV)) + " over " + (c < 0 ? "?" : std::to_string(c)) + " channels, k = " + (k < 0 ? "?" : std::to_string(k)) +
                                   (want_dil > 0 ? ", dilation " + std::to_string(want_dil) : std::string()));
        c = p->out; k = p->k;
        const int dil = p->dil;
        bind(name + ".dw.w", p->w, false, *p);
        bind(name + ".dw.b", p->b, false, *p);
        ++i;
        ln(name + ".ln", c);
        hid = linear(name + ".pw1", hid, c).out;
        linear(name + ".pw2", c, hid);
        p = peek();
        if (!p || p->kind != SCALE || p->out != c) fail(name + ".gamma", std::string(kind_name(SCALE)) + " over " + std::to_string(c) + " channels");
        bind(name + ".gamma", p->w, false, *p);
        ++i;
        return dil;
    }
    // attention block: LayerNorm -> q, k, v projections -> (attention) -> output projection.  cctx < 0: from the graph.  Returns heads (0 = not in the graph).
    // Fused forms: ONE projection of 3c rows from the block's own input (q | k | v, self-attention) or q alone followed by ONE
    // projection of 2c rows from the context (k | v): their row blocks are bound as the separate canonical tensors.
    int attn(const std::string& name, int c, int& cctx) {
        ln(name + ".ln", c);
        int heads = 0;
        auto h = [&](const Tok& tk) { if (!heads) heads = tk.heads; };
        auto fused = [&](const Tok& tk, std::initializer_list<const char*> parts) {
            const int n = (int)parts.size();
            // the row blocks are only q | k | v in this order if the graph itself cuts the result that way: find the cut and hold it to that
            std::string want;
            for (const char* part : parts) want += part;
            std::string problem;
            if (tk.split_parts == 0) problem = "no Split (or set of Slices) consumes its result, so how the " + std::to_string(tk.out) + " rows divide into " + want + " cannot be read from the graph";
            else if (!tk.split_err.empty()) problem = tk.split_err;
            else if (tk.split_parts != n) problem = tk.split_where + " makes " + std::to_string(tk.split_parts) + " parts where " + std::to_string(n) + " (" + want + ") are needed";
            else if (tk.split_roles != want)
                problem = tk.split_where + " hands its parts to the attention as [" + tk.split_roles + "] ('?': not traceable to the score / value MatMuls); the binding of row blocks needs [" + want + "]";
            if (!problem.empty())
                throw std::runtime_error(file + ": fused " + want + " projection of block " + name + " (" + tk.where + ", " + std::to_string(tk.out) + " <- " + std::to_string(tk.in) + "): " + problem +
                                         ". State the tensors in stn_weight_map.json instead (include/stn.h).");
            int j = 0;
            for (const char* part : parts) {
                bind(name + "." + part + ".w", tk.w, tk.transposed, tk, j * c, n * c, c);
                bind(name + "." + part + ".b", tk.b, false, tk, j * c, tk.b ? n * c : 0, c);
                ++j;
            }
            h(tk);
            ++i;
        };
        const Tok* p = peek();
        if (p && p->kind == LINEAR && p->out == 3 * c && p->in == c && (cctx < 0 || cctx == c)) {
            cctx = c;
            fused(*p, {"q", "k", "v"});
        } else {
            h(linear(name + ".q", c, c));
            p = peek();
            if (p && p->kind == LINEAR && p->out == 2 * c && (cctx < 0 || p->in == cctx)) {
                cctx = p->in;
                fused(*p, {"k", "v"});
            } else {
                const Tok& kk = linear(name + ".k", c, cctx);
                cctx = kk.in;
                h(kk);
                h(linear(name + ".v", c, cctx));
            }
        }
        linear(name + ".o", c, c);
        return heads;
    }
    void done() {
        if (i != t.size()) fail("(end of the graph)", "no further weighted node");
    }
};

void set_heads(Result& r, int32_t& field, const char* fname, int found, int c, bool explicit_) {
    if (found > 0) {
        if (c % found) throw std::runtime_error(std::string(fname) + " = " + std::to_string(found) + " (from a Reshape constant) does not divide the width " + std::to_string(c));
        field = found;
        return;
    }
    // A head count changes every attention result and no weight shape shows it: never guessed.
    if (!explicit_)
        throw std::runtime_error(std::string(fname) + " is not readable from the graph (no [batch, length, heads, head_dim] Reshape constant inside an attention block) "
                                 "and was not stated: put {\"arch\": {\"" + fname + "\": <n>}} into stn_weight_map.json beside the graphs (a manifest without a "
                                 "\"tensors\" table only states descriptor fields; the graphs are still walked)");
    if (field <= 0 || c % field) throw std::runtime_error(std::string(fname) + " = " + std::to_string(field) + " (stated in stn_weight_map.json) does not divide the width " + std::to_string(c));
    r.notes += std::string(fname) + " = " + std::to_string(field) + " as stated in stn_weight_map.json (the graph does not carry it); ";
}
void agree(const char* what, int from_graph, int from_json, const std::string& file) {
    if (from_graph != from_json)
        throw std::runtime_error(file + ": " + what + " = " + std::to_string(from_graph) + " by the graph's weight shapes, but tts.json says " + std::to_string(from_json));
}

}  // namespace

Result bind(const stn_arch& base, const Model& dpm, const Model& tem, const Model& vem, const Model& vom, bool heads_explicit) {
    Result r;
    r.arch = base;
    stn_arch& a = r.arch;
    auto S = [](const char* fmt, int i, int j = 0) { char b[64]; snprintf(b, sizeof b, fmt, i, j); return std::string(b); };
    const int D = base.latent_dim * base.chunk_compress_factor;

    {   // ---- duration predictor: embedding, ConvNeXt x n, style cross-attention, LayerNorm, two projections -------------------
        std::string gelu;
        const std::vector<Tok> toks = weighted_nodes(dpm, "duration_predictor.onnx", &gelu);
        r.gelu = gelu;
        Parser p{toks, "duration_predictor.onnx", "dp", r};
        a.dp_conv_blocks = 0;
        const Tok* e = p.peek();
        if (!e || e->kind != EMBED) p.fail("dp.emb", std::string(kind_name(EMBED)) + " [vocab][dp_dim]");
        a.vocab_size = e->out; a.dp_dim = e->in;
        p.bind("dp.emb", e->w, false, *e);
        ++p.i;
        int c = a.dp_dim, hid = -1, k = -1;
        while (p.next_is(DWCONV)) { p.convnext(S("dp.conv%d", a.dp_conv_blocks), c, hid, k, 1); ++a.dp_conv_blocks; a.dp_hidden = hid; a.dp_kernel = k; }
        int cctx = -1;
        set_heads(r, a.dp_heads, "dp_heads", p.attn("dp.st", c, cctx), c, heads_explicit);
        agree("d_style_dp (key projection of dp.st)", cctx, base.d_style_dp, p.file);
        p.ln("dp.out_ln", c);
        p.linear("dp.fc1", c, c);
        p.linear("dp.fc2", 1, c);
        p.done();
    }
    {   // ---- text encoder: embedding, ConvNeXt x n, {self-attention, FFN} x n, style cross-attention x n, LayerNorm, projection ---
        std::string gelu;
        const std::vector<Tok> toks = weighted_nodes(tem, "text_encoder.onnx", &gelu);
        if (!gelu.empty() && !r.gelu.empty() && gelu != r.gelu) r.notes += "text_encoder.onnx writes GELU as '" + gelu + "', an earlier graph as '" + r.gelu + "'; ";
        if (r.gelu.empty()) r.gelu = gelu;
        Parser p{toks, "text_encoder.onnx", "te", r};
        a.te_conv_blocks = a.te_attn_blocks = a.te_style_blocks = 0;
        const Tok* e = p.peek();
        if (!e || e->kind != EMBED) p.fail("te.emb", std::string(kind_name(EMBED)) + " [vocab][te_dim]");
        if (e->out != a.vocab_size)
            throw std::runtime_error("text_encoder.onnx: embedding table has " + std::to_string(e->out) + " rows, duration_predictor.onnx has " + std::to_string(a.vocab_size) + " (one unicode_indexer.json serves both)");
        a.te_dim = e->in;
        p.bind("te.emb", e->w, false, *e);
        ++p.i;
        int c = a.te_dim, hid = -1, k = -1, heads = 0;
        while (p.next_is(DWCONV)) { p.convnext(S("te.conv%d", a.te_conv_blocks), c, hid, k, 1); ++a.te_conv_blocks; a.te_hidden = hid; a.te_kernel = k; }
        // an attention block (LN q k v o) followed by LN + two projections is a self-attention block with its FFN; without them it
        // is a style cross-attention block; LN + ONE projection ends the graph
        int ffn = -1;
        // projections of the attention block whose LayerNorm sits `at` tokens ahead: 4 (q, k, v, o), 3 (q, fused k|v, o), 2 (fused
        // q|k|v, o — told from an FFN's two projections by the shapes: 3c <- c then c <- c); 0: not an attention block
        auto attn_linears = [&](size_t at) -> int {
            if (!p.next_is(LN, at)) return 0;
            int n = 0;
            while (p.next_is(LINEAR, at + 1 + n)) ++n;
            if (n == 2) {
                const Tok *l1 = p.peek(at + 1), *l2 = p.peek(at + 2);
                return (l1->out == 3 * c && l1->in == c && l2->out == c && l2->in == c) ? 2 : 0;
            }
            return (n == 3 || n == 4) ? n : 0;
        };
        while (const int nl = attn_linears(0)) {
            const size_t f = 1 + (size_t)nl;  // where an FFN's LayerNorm would sit
            const bool has_ffn = p.next_is(LN, f) && p.next_is(LINEAR, f + 1) && p.next_is(LINEAR, f + 2) && !p.next_is(LINEAR, f + 3) && attn_linears(f) == 0;
            if (has_ffn) {
                if (a.te_style_blocks) p.fail("te.st" + std::to_string(a.te_style_blocks), "a style cross-attention block (self-attention blocks come first)");
                const std::string n = S("te.sa%d", a.te_attn_blocks);
                int cctx = c;
                const int h = p.attn(n, c, cctx);
                if (!heads) heads = h;
                p.ln(n + ".ffn_ln", c);
                ffn = p.linear(n + ".ffn1", ffn, c).out;
                p.linear(n + ".ffn2", c, ffn);
                a.te_ffn = ffn;
                ++a.te_attn_blocks;
            } else {
                int cctx = base.d_style_ttl;
                const int h = p.attn(S("te.st%d", a.te_style_blocks), c, cctx);
                if (!heads) heads = h;
                ++a.te_style_blocks;
            }
        }
        set_heads(r, a.te_heads, "te_heads", heads, c, heads_explicit);
        p.ln("te.out_ln", c);
        const int odim = p.linear("te.proj", -1, c).out;
        agree("te_out_dim (rows of te.proj)", odim, base.te_out_dim, p.file);
        p.done();
    }
    {   // ---- vector estimator ---------------------------------------------------------------------------------------------
        std::string gelu;
        const std::vector<Tok> toks = weighted_nodes(vem, "vector_estimator.onnx", &gelu);
        if (!gelu.empty() && !r.gelu.empty() && gelu != r.gelu) r.notes += "vector_estimator.onnx writes GELU as '" + gelu + "', an earlier graph as '" + r.gelu + "'; ";
        if (r.gelu.empty()) r.gelu = gelu;
        Parser p{toks, "vector_estimator.onnx", "ve", r};
        a.ve_main_blocks = a.ve_tail_blocks = 0;
        const Tok& in = p.linear("ve.in", -1, -1);
        a.ve_dim = in.out;
        agree("latent_dim * chunk_compress_factor (columns of ve.in)", in.in, D, p.file);
        int c = a.ve_dim;
        a.ve_time_dim = p.linear("ve.t1", c, -1).in;
        p.linear("ve.t2", c, c);
        int hid = -1, k = -1, heads = 0, dilated = -1;
        a.ve_dilated = 0;
        auto cn = [&](const std::string& name, int dil) { p.convnext(name, c, hid, k, dil); a.ve_hidden = hid; a.ve_kernel = k; };
        // Only a main block's time projection is a projection followed directly by a depthwise conv with nothing but ConvNeXt
        // blocks before it: while one lies ahead, the ConvNeXt run in front of it is a main block's dilated run (dilation 2^j);
        // the last run, followed by LayerNorm, is the tail.
        auto main_block_ahead = [&]() {
            for (size_t j = 0; p.peek(j + 1); ++j) {
                const Tok* t0 = p.peek(j);
                if (t0->kind == LINEAR && p.peek(j + 1)->kind == DWCONV && t0->out == c && t0->in == c) return true;
                if (t0->kind == LN && j > 0 && p.peek(j - 1)->kind != DWCONV) return false;  // an attention / output LayerNorm: past the run
            }
            return false;
        };
        while (p.next_is(DWCONV) || (p.next_is(LINEAR) && p.next_is(DWCONV, 1))) {  // (a main block without dilated blocks opens with its time projection)
            if (!main_block_ahead()) {
                while (p.next_is(DWCONV)) { cn(S("ve.tail%d", a.ve_tail_blocks), 1); ++a.ve_tail_blocks; }
                break;
            }
            const int b = a.ve_main_blocks;
            int j = 0;
            while (dilated < 0 ? p.next_is(DWCONV) : j < dilated) { cn(S("ve.m%d.dil%d", b, j), 1 << j); ++j; }
            if (dilated < 0) { dilated = j; a.ve_dilated = j; }
            p.linear(S("ve.m%d.time", b), c, c);
            cn(S("ve.m%d.cn_a", b), 1);
            int cctx = base.te_out_dim;
            int h = p.attn(S("ve.m%d.text", b), c, cctx);
            if (!heads) heads = h;
            cn(S("ve.m%d.cn_b", b), 1);
            cctx = base.d_style_ttl;
            h = p.attn(S("ve.m%d.style", b), c, cctx);
            if (!heads) heads = h;
            ++a.ve_main_blocks;
        }
        set_heads(r, a.ve_heads, "ve_heads", heads, c, heads_explicit);
        p.ln("ve.out_ln", c);
        p.linear("ve.out", D, c);
        p.done();
    }
    {   // ---- vocoder: k-tap input conv, ConvNeXt x n (per-block dilation), LayerNorm, head --------------------------------------
        std::string gelu;
        const std::vector<Tok> toks = weighted_nodes(vom, "vocoder.onnx", &gelu);
        if (!gelu.empty() && !r.gelu.empty() && gelu != r.gelu) r.notes += "vocoder.onnx writes GELU as '" + gelu + "', an earlier graph as '" + r.gelu + "'; ";
        if (r.gelu.empty()) r.gelu = gelu;
        Parser p{toks, "vocoder.onnx", "vo", r};
        a.vo_blocks = 0;
        const Tok* e = p.peek();
        if (!e || (e->kind != CONVK && e->kind != LINEAR) || e->in != base.latent_dim)
            p.fail("vo.in", std::string(kind_name(CONVK)) + " vo_dim <- " + std::to_string(base.latent_dim) + " (latent_dim of tts.json)");
        a.vo_dim = e->out; a.vo_in_kernel = e->k;
        p.bind("vo.in.w", e->w, false, *e);
        p.bind("vo.in.b", e->b, false, *e);
        if (e->kind == LINEAR && e->transposed) p.fail("vo.in", "a Conv (a MatMul cannot be the k-tap input convolution)");
        ++p.i;
        int c = a.vo_dim, hid = -1, k = -1;
        while (p.next_is(DWCONV)) {
            if (a.vo_blocks >= STN_MAX_VO_BLOCKS) p.fail("vo.blk" + std::to_string(a.vo_blocks), "at most " + std::to_string(STN_MAX_VO_BLOCKS) + " vocoder blocks (STN_MAX_VO_BLOCKS)");
            a.vo_dilations[a.vo_blocks] = p.convnext(S("vo.blk%d", a.vo_blocks), c, hid, k, 0);
            ++a.vo_blocks;
            a.vo_hidden = hid; a.vo_kernel = k;
        }
        p.ln("vo.out_ln", c);
        const int chunk = p.linear("vo.head", -1, c).out;
        agree("base_chunk_size (rows of vo.head)", chunk, base.base_chunk_size, p.file);
        p.done();
    }
    if (r.gelu == "tanh")
        r.notes += "the graphs spell GELU with Tanh (the tanh approximation): the engine computes that form (stn_set_gelu_form 1: exact in fp32 and f16, "
                   "the exp2 shortcut of the same function in bf16, kernels_dev.hpp); ";
    return r;
}

bool apply_arch_overrides(const std::string& dir, stn_arch& a) {
    std::ifstream f(dir + "/stn_weight_map.json", std::ios::binary);
    if (!f.is_open()) return false;
    const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const json::Value man = json::parse(text);
    if (man.has("tensors") || !man.has("arch")) return false;
    static const std::pair<const char*, int32_t stn_arch::*> fields[] = {
        {"te_heads", &stn_arch::te_heads}, {"dp_heads", &stn_arch::dp_heads}, {"ve_heads", &stn_arch::ve_heads}};
    for (const auto& kv : man.at("arch").obj) {
        bool ok = false;
        for (const auto& fld : fields) if (kv.first == fld.first) { a.*(fld.second) = kv.second.as_int(); ok = true; }
        if (!ok) throw std::runtime_error("stn_weight_map.json without a \"tensors\" table may state only the head counts (te_heads, dp_heads, ve_heads), which no weight shape "
                                          "shows; \"" + kv.first + "\" comes out of the graphs");
    }
    return true;
}

stn_arch arch_from_config(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open " + path);
    const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    using json::Value;
    const Value cfg = json::parse(text);
    stn_arch a;
    stn_arch_default(&a);
    a.sample_rate = cfg.at("ae").at("sample_rate").as_int();
    a.base_chunk_size = cfg.at("ae").at("base_chunk_size").as_int();
    a.chunk_compress_factor = cfg.at("ttl").at("chunk_compress_factor").as_int();
    a.latent_dim = cfg.at("ttl").at("latent_dim").as_int();
    auto opt = [](const Value& v, std::initializer_list<const char*> keys, int32_t& dst) {
        const Value* cur = &v;
        for (const char* k : keys) { if (!cur->is_object() || !cur->has(k)) return; cur = &cur->at(k); }
        if (cur->type == Value::Number) dst = (int32_t)cur->num;
    };
    opt(cfg, {"ttl", "style_encoder", "style_token_layer", "n_style"}, a.n_style_ttl);
    opt(cfg, {"ttl", "style_encoder", "style_token_layer", "style_value_dim"}, a.d_style_ttl);
    opt(cfg, {"ttl", "text_encoder", "proj_out", "odim"}, a.te_out_dim);
    opt(cfg, {"dp", "style_encoder", "style_token_layer", "n_style"}, a.n_style_dp);
    opt(cfg, {"dp", "style_encoder", "style_token_layer", "style_value_dim"}, a.d_style_dp);
    return a;
}

void check_all_io_names(const Model& dp, const Model& te, const Model& ve, const Model& vo) {
    // the names every host feeds and fetches (/root/reference/cpp/helper.cpp:512-513, 545-546, 620-623, 663-664)
    check_io_names(dp, "duration_predictor.onnx", {"text_ids", "style_dp", "text_mask"}, {"duration"});
    check_io_names(te, "text_encoder.onnx", {"text_ids", "style_ttl", "text_mask"}, {"text_emb"});
    check_io_names(ve, "vector_estimator.onnx", {"noisy_latent", "text_emb", "style_ttl", "text_mask", "latent_mask", "total_step", "current_step"},
                   {"denoised_latent"});
    check_io_names(vo, "vocoder.onnx", {"latent"}, {"wav_tts"});
}

std::string bind_dir_json(const std::string& dir) {
    stn_arch base = arch_from_config(dir + "/tts.json");
    const bool heads_explicit = apply_arch_overrides(dir, base);
    const Model dp = onnx::parse_file(dir + "/duration_predictor.onnx"), te = onnx::parse_file(dir + "/text_encoder.onnx"),
                ve = onnx::parse_file(dir + "/vector_estimator.onnx"), vo = onnx::parse_file(dir + "/vocoder.onnx");
    check_all_io_names(dp, te, ve, vo);
    const Result r = bind(base, dp, te, ve, vo, heads_explicit);
    auto esc = [](const std::string& s) { std::string o; for (char c : s) { if (c == '"' || c == '\\') o.push_back('\\'); o.push_back(c); } return o; };
    std::ostringstream o;
    const stn_arch& a = r.arch;
    o << "{\"arch\":{";
#define F(x) "\"" #x "\":" << a.x
    o << F(sample_rate) << "," << F(base_chunk_size) << "," << F(chunk_compress_factor) << "," << F(latent_dim) << "," << F(vocab_size) << ","
      << F(n_style_ttl) << "," << F(d_style_ttl) << "," << F(n_style_dp) << "," << F(d_style_dp) << ","
      << F(te_dim) << "," << F(te_hidden) << "," << F(te_kernel) << "," << F(te_conv_blocks) << "," << F(te_attn_blocks) << "," << F(te_heads) << ","
      << F(te_ffn) << "," << F(te_style_blocks) << "," << F(te_out_dim) << ","
      << F(dp_dim) << "," << F(dp_hidden) << "," << F(dp_kernel) << "," << F(dp_conv_blocks) << "," << F(dp_heads) << ","
      << F(ve_dim) << "," << F(ve_hidden) << "," << F(ve_kernel) << "," << F(ve_main_blocks) << "," << F(ve_dilated) << "," << F(ve_tail_blocks) << ","
      << F(ve_heads) << "," << F(ve_time_dim) << ","
      << F(vo_dim) << "," << F(vo_hidden) << "," << F(vo_kernel) << "," << F(vo_blocks) << "," << F(vo_in_kernel) << ",\"vo_dilations\":[";
#undef F
    for (int i = 0; i < a.vo_blocks; ++i) o << (i ? "," : "") << a.vo_dilations[i];
    o << "]},\"tensors\":{";
    bool first = true;
    for (const auto& kv : r.tensors) {
        o << (first ? "" : ",") << "\"" << kv.first << "\":{\"from\":\"" << esc(kv.second.from) << "\",\"transpose\":" << (kv.second.transpose ? "true" : "false")
          << ",\"zeros\":" << (kv.second.t ? "false" : "true") << ",\"row0\":" << kv.second.row0 << ",\"rows_total\":" << kv.second.rows_total << "}";
        first = false;
    }
    o << "},\"gelu\":\"" << esc(r.gelu) << "\",\"notes\":\"" << esc(r.notes) << "\"}";
    return o.str();
}

void check_io_names(const Model& m, const std::string& file, const std::vector<std::string>& inputs, const std::vector<std::string>& outputs) {
    // older exporters list the initializers among the graph inputs as well
    std::vector<std::string> in;
    for (const std::string& n : m.inputs) if (!m.find(n)) in.push_back(n);
    auto join = [](const std::vector<std::string>& v) { std::string s; for (auto& x : v) s += (s.empty() ? "" : ", ") + x; return "{" + s + "}"; };
    auto same_set = [](std::vector<std::string> x, std::vector<std::string> y) { std::sort(x.begin(), x.end()); std::sort(y.begin(), y.end()); return x == y; };
    if (!same_set(in, inputs)) throw std::runtime_error(file + ": graph inputs are " + join(in) + ", the host feeds " + join(inputs));
    if (!same_set(m.outputs, outputs)) throw std::runtime_error(file + ": graph outputs are " + join(m.outputs) + ", the host fetches " + join(outputs));
}

std::vector<float> fetch(const Bound& b, const std::string& canonical, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    if (!b.t) return std::vector<float>(n, 0.f);
    std::vector<float> v = onnx::to_float(*b.t);
    if (b.broadcast > 0 && v.size() == 1) v.assign((size_t)b.broadcast, v[0]);
    // a row block of a fused projection: the initializer holds rows_total x cols (1-D tensors, the biases: rows_total elements)
    const bool vec = rows == 1 && b.rows_total > 0;  // canonical vectors are [1][n]
    const int full_rows = b.rows_total ? (vec ? 1 : b.rows_total) : rows, full_cols = vec ? b.rows_total : cols;
    if (v.size() != (size_t)full_rows * full_cols)
        throw std::runtime_error(canonical + ": " + b.from + " has " + std::to_string(v.size()) + " elements, descriptor wants " + std::to_string(full_rows) + "x" + std::to_string(full_cols));
    if (b.transpose) {
        std::vector<float> w(v.size());
        for (int r = 0; r < full_rows; ++r) for (int c = 0; c < full_cols; ++c) w[(size_t)r * full_cols + c] = v[(size_t)c * full_rows + r];
        v.swap(w);
    }
    if (!b.rows_total) return v;
    if (vec) return std::vector<float>(v.begin() + b.row0, v.begin() + b.row0 + cols);
    return std::vector<float>(v.begin() + (size_t)b.row0 * cols, v.begin() + (size_t)(b.row0 + rows) * cols);
}

std::vector<float> fetch_canonical(const Bound& b, const std::string& canonical) {
    if (!b.t) return {};
    std::vector<int64_t> d;
    for (int64_t x : b.t->dims) if (x != 1) d.push_back(x);
    if (d.size() <= 1) {  // a vector (bias, LayerNorm parameter, scale) or a block of one
        const int n = b.broadcast > 0 && b.t->numel() == 1 ? b.broadcast : (int)b.t->numel();
        return fetch(b, canonical, 1, b.rows_total ? b.nrows : n);
    }
    int64_t rest = 1;
    for (size_t i = 1; i < d.size(); ++i) rest *= d[i];
    const int rows = (int)(b.transpose ? d[1] : d[0]), cols = (int)(b.transpose ? d[0] : rest);
    return fetch(b, canonical, b.rows_total ? b.nrows : rows, cols);
}

std::vector<float> bound_tensor_of_dir(const std::string& dir, const std::string& canonical) {
    stn_arch base = arch_from_config(dir + "/tts.json");
    const bool heads_explicit = apply_arch_overrides(dir, base);
    const Model dp = onnx::parse_file(dir + "/duration_predictor.onnx"), te = onnx::parse_file(dir + "/text_encoder.onnx"),
                ve = onnx::parse_file(dir + "/vector_estimator.onnx"), vo = onnx::parse_file(dir + "/vocoder.onnx");
    const Result r = bind(base, dp, te, ve, vo, heads_explicit);
    auto it = r.tensors.find(canonical);
    if (it == r.tensors.end()) throw std::runtime_error("no canonical tensor \"" + canonical + "\" in the binding of " + dir);
    return fetch_canonical(it->second, canonical);
}

}  // namespace graphbind
}  // namespace stn
