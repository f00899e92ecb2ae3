// engine.cpp — weights, workspace and the four stage executors (see engine.hpp).
#include "engine.hpp"
#include "dev_env.hpp"
#include "engine_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace stn {
using detail::up;

// =================================================================================================
// Arena
// =================================================================================================
Arena::~Arena() {
    for (auto& c : chunks_) (void)hipFree(c.p);
}
size_t Arena::capacity() const {
    size_t t = 0;
    for (auto& c : chunks_) t += c.cap;
    return t;
}
void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (;;) {
        if (cur_ < chunks_.size()) {
            Chunk& c = chunks_[cur_];
            if (off_ + bytes <= c.cap) {
                void* p = c.p + off_;
                off_ += bytes;
                return p;
            }
            ++cur_;
            off_ = 0;
            continue;
        }
        Chunk c;
        c.cap = std::max(bytes, size_t(256) << 20);
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&c.p), c.cap));
        chunks_.push_back(c);
    }
}

// =================================================================================================
// deterministic synthetic weights (spec shared with the oracle by definition, not by code):
//   value(name, i) = offs + scale * u,  u = top 24 bits of mix64(mix64(seed ^ fnv1a(name)) + i) / 2^23 - 1
// =================================================================================================
namespace {
enum Kind { K_W, K_BIAS, K_LN_G, K_LN_B, K_LSCALE, K_EMB };

uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
    return h;
}
uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
std::vector<float> synth(uint64_t seed, const std::string& name, Kind kind, int rows, int cols, float gain) {
    const size_t n = (size_t)rows * cols;
    std::vector<float> v(n);
    float scale = 1.f, offs = 0.f;
    switch (kind) {
        case K_W: scale = std::sqrt(3.0f / (float)cols) * gain; break;
        case K_BIAS: scale = 0.05f; break;
        case K_LN_G: scale = 0.1f; offs = 1.0f; break;
        case K_LN_B: scale = 0.05f; break;
        case K_LSCALE: scale = 0.1f; offs = 0.2f; break;
        case K_EMB: scale = std::sqrt(3.0f); break;
    }
    const uint64_t base = mix64(seed ^ fnv1a(name));
    for (size_t i = 0; i < n; ++i) {
        const uint64_t h = mix64(base + i);
        const float u = (float)(h >> 40) * (1.0f / 8388608.0f) - 1.0f;
        v[i] = offs + scale * u;
    }
    return v;
}
uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // RNE (weights are finite)
}
uint16_t f32_to_f16(float f) {  // IEEE binary16, round to nearest even (the compiler's conversion; weights are finite)
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
}  // namespace

// =================================================================================================
// Engine: construction / weights
// =================================================================================================
Engine::Engine(int device, int dtype) : device_(device), dt_(dtype) {
    if (dtype != F32 && dtype != BF16 && dtype != F16) throw std::runtime_error("dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) throw std::runtime_error("no HIP device available: the engine has no CPU fallback");
    if (device < 0 || device >= n) throw std::runtime_error("device index out of range");
    STN_HIP(hipSetDevice(device));
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) n_cu_ = cu; else (void)hipGetLastError(); }
    // Stream priorities, experiment switch STN_PRIO=<main><side>, each h / n / l (default nn: all streams at the default priority).  Measured:
    // the main stream at the highest priority gains 0.6 % for one batch at a time (11.59 -> 11.52 ms) and LOSES a third of the rate with two
    // handles in flight (53 k -> 34 k audio-s/s: two highest-priority queues no longer interleave) — not adopted
    int prio_least = 0, prio_greatest = 0;
    STN_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const char* pr = stn::dev_env("STN_PRIO");
    auto prio_of = [&](char c) { return c == 'h' ? prio_greatest : c == 'l' ? prio_least : 0; };
    const int prio_main = pr && pr[0] ? prio_of(pr[0]) : 0, prio_side = pr && pr[0] && pr[1] ? prio_of(pr[1]) : 0;
    STN_HIP(hipStreamCreateWithPriority(&own_s_, hipStreamNonBlocking, prio_main));
    s_ = own_s_;
    STN_HIP(hipStreamCreateWithPriority(&dp_s_, hipStreamNonBlocking, prio_side));
    STN_HIP(hipEventCreateWithFlags(&ev_te_, hipEventDisableTiming));
    STN_HIP(hipEventCreateWithFlags(&ev_copied_, hipEventDisableTiming));
    STN_HIP(hipEventCreateWithFlags(&ev_dp_, hipEventDisableTiming));
    STN_HIP(hipStreamCreateWithPriority(&te_s_, hipStreamNonBlocking, prio_side));
    if (const char* p = stn::dev_env("STN_DP_STREAM")) if (atoi(p) == 0) {  // A/B switch: everything on the main stream
        (void)hipStreamDestroy(dp_s_); dp_s_ = nullptr;
        (void)hipStreamDestroy(te_s_); te_s_ = nullptr;
    }
    if (const char* p = stn::dev_env("STN_NT")) nt_hints_ = atoi(p) != 0;  // A/B switch: non-temporal hints on the vocoder's hidden activation
    if (const char* p = stn::dev_env("STN_FFN")) fused_ffn_ = atoi(p);          // A/B switch: K4 stage mask (1 vocoder, 2 estimator, 4 text stages)
    if (const char* p = stn::dev_env("STN_FFN_MIN_ROWS")) ffn_min_rows_ = atoll(p);
    if (const char* p = stn::dev_env("STN_XATTN")) set_fused_xattn(atoi(p));  // A/B switch: cross-attention blocks head-split (default) or as four launches (0)
    if (const char* p = stn::dev_env("STN_FFN_SPLIT_MIN_ROWS")) ffn_split_min_rows_ = atoll(p);
    if (const char* p = stn::dev_env("STN_PACKED")) packed_ve_ = atoi(p) != 0;  // A/B switch for measurements (stn_set_row_layout overrides)
}

void Engine::set_stream(hipStream_t s) {
    sync();
    s_ = s ? s : own_s_;
}

void Engine::drop_graphs() {
    for (auto& g : graphs_) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        if (g.exec2) (void)hipGraphExecDestroy(g.exec2);
        if (g.graph2) (void)hipGraphDestroy(g.graph2);
    }
    graphs_.clear();
    warm_keys_.clear();
}

void Engine::free_weights() {
    // captured graphs point into the weights (and into the vocoder constants) that are about to be freed: none may survive
    if (s_) (void)hipStreamSynchronize(s_);
    drop_graphs();
    ++wgen_;
    for (void* p : owned_) (void)hipFree(p);
    owned_.clear();
    w_.clear();
    frag_w_.clear();
    ffn_w_.clear();
    loaded_ = false;
    params_ = 0;
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (s_) (void)hipStreamSynchronize(s_);
    if (dp_s_) (void)hipStreamSynchronize(dp_s_);
    if (te_s_) (void)hipStreamSynchronize(te_s_);
    free_weights();
    for (void* p : batch_owned_) (void)hipFree(p);
    for (void* p : batch_retired_) (void)hipFree(p);
    if (vo_quiet_) (void)hipFree(vo_quiet_);
    if (vo_edge_) (void)hipFree(vo_edge_);
    if (tcond_.buf) (void)hipFree(tcond_.buf);
    for (auto& sp : spans_) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto ev : ev_pool_) (void)hipEventDestroy(ev);
    drop_graphs();
    if (pin_llen_) (void)hipHostFree(pin_llen_);
    if (pin_seed_) (void)hipHostFree(pin_seed_);
    if (seed_dev_) (void)hipFree(seed_dev_);
    for (auto& f : fetch_) {
        if (f.busy && f.done) (void)hipEventSynchronize(f.done);
        if (f.dev) (void)hipFree(f.dev);
        if (f.pin) (void)hipHostFree(f.pin);
        if (f.ready) (void)hipEventDestroy(f.ready);
        if (f.done) (void)hipEventDestroy(f.done);
    }
    if (copy_s_) (void)hipStreamDestroy(copy_s_);
    if (own_s_) (void)hipStreamDestroy(own_s_);
    if (dp_s_) (void)hipStreamDestroy(dp_s_);
    if (te_s_) (void)hipStreamDestroy(te_s_);
    if (ev_te_) (void)hipEventDestroy(ev_te_);
    if (ev_copied_) (void)hipEventDestroy(ev_copied_);
    if (ev_dp_) (void)hipEventDestroy(ev_dp_);
}

DevTensor& Engine::tensor(const std::string& name) {
    auto it = w_.find(name);
    if (it == w_.end()) throw std::runtime_error("unknown weight tensor: " + name);
    return it->second;
}
Linear Engine::linear(const std::string& p) {
    Linear l;
    l.w = tensor(p + ".w");
    l.b = tensor(p + ".b").f32;
    l.N = l.w.rows;
    l.K = l.w.cols;
    return l;
}
LNorm Engine::lnorm(const std::string& p) { return LNorm{tensor(p + ".g").f32, tensor(p + ".b").f32}; }
ConvNeXt Engine::convnext_w(const std::string& p) {
    ConvNeXt c;
    c.dw_t = tensor(p + ".dw.wt").f32;
    c.dw_b = tensor(p + ".dw.b").f32;
    c.ln = lnorm(p + ".ln");
    c.pw1 = linear(p + ".pw1");
    c.pw2 = linear(p + ".pw2");
    c.gamma = tensor(p + ".gamma").f32;
    return c;
}
Attn Engine::attn_w(const std::string& p, bool self) {
    Attn a;
    a.ln = lnorm(p + ".ln");
    a.q = linear(p + ".q");
    a.kv = linear(p + ".kv");
    if (self) a.qkv = linear(p + ".qkv");
    a.o = linear(p + ".o");
    return a;
}

void Engine::load_synthetic(const stn_arch& a, uint64_t seed) {
    load_weights(a, [seed](const std::string& name, int kind, int rows, int cols, float gain) {
        return synth(seed, name, (Kind)kind, rows, cols, gain); }, nullptr);
}
void Engine::load_tensors(const stn_arch& a, const TensorSource& src) {
    load_weights(a, [&src](const std::string& name, int, int rows, int cols, float) {
        std::vector<float> v = src(name, rows, cols);
        if (v.size() != (size_t)rows * cols)
            throw std::runtime_error("tensor " + name + ": expected " + std::to_string((size_t)rows * cols) + " values, got " + std::to_string(v.size()));
        return v; }, nullptr);
}
std::vector<std::string> Engine::tensor_names(const stn_arch& a) {
    std::vector<std::string> names;
    load_weights(a, RawSource(), &names);
    return names;
}

void Engine::load_weights(const stn_arch& a, const RawSource& src, std::vector<std::string>* names_only) {
    // the descriptor is caller data (tts.json, a manifest, a test): everything a kernel's contract depends on is checked here,
    // with the field's name, so that a graph the engine cannot run is an error code at load time and never a failed launch
    {
        auto bad = [](const std::string& what) { throw std::invalid_argument("unsupported architecture descriptor: " + what); };
        auto pos = [&](const char* n, int v) { if (v <= 0) bad(std::string(n) + " = " + std::to_string(v) + " (must be > 0)"); };
        auto width = [&](const char* n, int v) {
            pos(n, v);
            if (v % 8) bad(std::string(n) + " = " + std::to_string(v) + " (widths must be multiples of 8: 16-byte rows)");
            if (v > 1024) bad(std::string(n) + " = " + std::to_string(v) + " (LayerNorm / depthwise kernels hold a row of <= 1024 channels)");
        };
        auto hidden = [&](const char* n, int v) {  // widths that only feed GEMMs
            pos(n, v);
            if (v % 8 || v > 16384) bad(std::string(n) + " = " + std::to_string(v) + " (hidden widths: multiples of 8, <= 16384)");
        };
        auto heads = [&](const char* n, int c, int h) {
            pos(n, h);
            if (c % h || (c / h) % 8 || c / h < 8 || c / h > 96) bad(std::string(n) + " = " + std::to_string(h) + " (head dim " + std::to_string(c / std::max(h, 1)) + ": multiple of 8 in [8, 96])");
        };
        auto kern = [&](const char* n, int k) { if (k < 1 || k > 15 || !(k & 1)) bad(std::string(n) + " = " + std::to_string(k) + " (odd, <= 15)"); };
        pos("sample_rate", a.sample_rate); pos("base_chunk_size", a.base_chunk_size); pos("chunk_compress_factor", a.chunk_compress_factor);
        pos("latent_dim", a.latent_dim); pos("vocab_size", a.vocab_size);
        pos("n_style_ttl", a.n_style_ttl); width("d_style_ttl", a.d_style_ttl); pos("n_style_dp", a.n_style_dp); width("d_style_dp", a.d_style_dp);
        width("te_dim", a.te_dim); hidden("te_hidden", a.te_hidden); hidden("te_ffn", a.te_ffn); width("te_out_dim", a.te_out_dim);
        width("dp_dim", a.dp_dim); hidden("dp_hidden", a.dp_hidden); width("ve_dim", a.ve_dim); hidden("ve_hidden", a.ve_hidden);
        width("vo_dim", a.vo_dim); hidden("vo_hidden", a.vo_hidden); width("ve_time_dim", a.ve_time_dim);
            heads("te_heads", a.te_dim, a.te_heads); heads("dp_heads", a.dp_dim, a.dp_heads); heads("ve_heads", a.ve_dim, a.ve_heads);
        kern("te_kernel", a.te_kernel); kern("dp_kernel", a.dp_kernel); kern("ve_kernel", a.ve_kernel); kern("vo_kernel", a.vo_kernel); kern("vo_in_kernel", a.vo_in_kernel);
        if (a.te_conv_blocks < 0 || a.te_attn_blocks < 0 || a.te_style_blocks < 0 || a.dp_conv_blocks < 0 || a.ve_main_blocks < 0 || a.ve_dilated < 0 ||
            a.ve_tail_blocks < 0 || a.vo_blocks < 0) bad("a block count is negative");
        if (a.ve_dilated > 8) bad("ve_dilated = " + std::to_string(a.ve_dilated) + " (dilations 2^j, j < 8)");
        if (a.vo_blocks > STN_MAX_VO_BLOCKS) bad("vo_blocks = " + std::to_string(a.vo_blocks) + " exceeds STN_MAX_VO_BLOCKS");
        for (int i = 0; i < a.vo_blocks; ++i)
            if (a.vo_dilations[i] < 1 || a.vo_dilations[i] > 64) bad("vo_dilations[" + std::to_string(i) + "] = " + std::to_string(a.vo_dilations[i]));
        if (a.base_chunk_size % 4) bad("base_chunk_size = " + std::to_string(a.base_chunk_size) + " (multiple of 4: 16-byte waveform rows)");
        if (a.ve_time_dim % 2) bad("ve_time_dim must be even (sin/cos pairs)");
        if (!(a.ln_eps > 0.f)) bad("ln_eps must be > 0");
    }
    if (!names_only) {
        STN_HIP(hipSetDevice(device_));
        sync();
        free_weights();
        a_ = a;
    }
    const int D = a.latent_dim * a.chunk_compress_factor;

    std::unordered_map<std::string, std::vector<float>> host;  // canonical copies kept for derived tensors
    auto upload = [&](const std::string& name, const std::vector<float>& v, int rows, int cols, bool want_bf16) {
        if (names_only) return;
        DevTensor t;
        t.rows = rows;
        t.cols = cols;
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&t.f32), std::max<size_t>(v.size(), 4) * 4));
        owned_.push_back(t.f32);
        STN_HIP(hipMemcpy(t.f32, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        if (want_bf16) {
            std::vector<uint16_t> h(v.size());
            for (size_t i = 0; i < v.size(); ++i) h[i] = dt_ == F16 ? f32_to_f16(v[i]) : f32_to_bf16(v[i]);
            STN_HIP(hipMalloc(reinterpret_cast<void**>(&t.bf16), std::max<size_t>(v.size(), 8) * 2));
            owned_.push_back(t.bf16);
            STN_HIP(hipMemcpy(t.bf16, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        }
        w_[name] = t;
    };
    auto decl = [&](const std::string& name, Kind kind, int rows, int cols, float gain, bool matrix, bool keep) {
        if (names_only) { names_only->push_back(name); if (keep) host[name] = std::vector<float>((size_t)rows * cols, 0.f); return; }
        std::vector<float> v = src(name, (int)kind, rows, cols, gain);
        params_ += (int64_t)v.size();
        upload(name, v, rows, cols, matrix);
        if (keep) host[name] = std::move(v);
    };
    auto decl_linear = [&](const std::string& p, int out, int in, float gain, bool keep) {
        decl(p + ".w", K_W, out, in, gain, true, keep);
        decl(p + ".b", K_BIAS, 1, out, 1.f, false, keep);
    };
    auto decl_ln = [&](const std::string& p, int c) {
        decl(p + ".g", K_LN_G, 1, c, 1.f, false, false);
        decl(p + ".b", K_LN_B, 1, c, 1.f, false, false);
    };
    auto decl_convnext = [&](const std::string& p, int c, int hid, int k) {
        if (names_only) names_only->push_back(p + ".dw.w");
        std::vector<float> w = names_only ? std::vector<float>((size_t)c * k, 0.f) : src(p + ".dw.w", (int)K_W, c, k, 1.f);  // canonical [C][k]
        if (!names_only) params_ += (int64_t)w.size();
        std::vector<float> wt((size_t)c * k);                            // stored [k][C]: coalesced per tap
        for (int ch = 0; ch < c; ++ch) for (int j = 0; j < k; ++j) wt[(size_t)j * c + ch] = w[(size_t)ch * k + j];
        upload(p + ".dw.wt", wt, k, c, false);
        decl(p + ".dw.b", K_BIAS, 1, c, 1.f, false, false);
        decl_ln(p + ".ln", c);
        decl_linear(p + ".pw1", hid, c, 1.f, false);
        decl_linear(p + ".pw2", c, hid, 1.f, false);
        decl(p + ".gamma", K_LSCALE, 1, c, 1.f, false, false);
    };
    // concat rows of several [rows_i][cols] matrices (and their biases) into one GEMM operand
    auto concat = [&](const std::string& name, const std::vector<std::string>& parts, int cols) {
        std::vector<float> w, b;
        for (auto& p : parts) {
            auto& pw = host.at(p + ".w");
            auto& pb = host.at(p + ".b");
            w.insert(w.end(), pw.begin(), pw.end());
            b.insert(b.end(), pb.begin(), pb.end());
        }
        upload(name + ".w", w, (int)(w.size() / cols), cols, true);
        upload(name + ".b", b, 1, (int)b.size(), false);
    };
    auto decl_attn = [&](const std::string& p, int c, int cctx, bool self) {
        decl_ln(p + ".ln", c);
        decl_linear(p + ".q", c, c, 1.f, true);
        decl_linear(p + ".k", c, cctx, 1.f, true);
        decl_linear(p + ".v", c, cctx, 1.f, true);
        decl_linear(p + ".o", c, c, 1.f, false);
        concat(p + ".kv", {p + ".k", p + ".v"}, cctx);
        if (self) concat(p + ".qkv", {p + ".q", p + ".k", p + ".v"}, c);
    };
    auto S = [](const char* fmt, int i, int j = 0) { char b[64]; snprintf(b, sizeof b, fmt, i, j); return std::string(b); };

    // duration predictor
    decl("dp.emb", K_EMB, a.vocab_size, a.dp_dim, 1.f, false, false);
    for (int i = 0; i < a.dp_conv_blocks; ++i) decl_convnext(S("dp.conv%d", i), a.dp_dim, a.dp_hidden, a.dp_kernel);
    decl_attn("dp.st", a.dp_dim, a.d_style_dp, false);
    decl_ln("dp.out_ln", a.dp_dim);
    decl_linear("dp.fc1", a.dp_dim, a.dp_dim, 1.f, false);
    decl_linear("dp.fc2", 1, a.dp_dim, 1.f, false);
    // text encoder
    decl("te.emb", K_EMB, a.vocab_size, a.te_dim, 1.f, false, false);
    for (int i = 0; i < a.te_conv_blocks; ++i) decl_convnext(S("te.conv%d", i), a.te_dim, a.te_hidden, a.te_kernel);
    for (int i = 0; i < a.te_attn_blocks; ++i) {
        decl_attn(S("te.sa%d", i), a.te_dim, a.te_dim, true);
        decl_ln(S("te.sa%d.ffn_ln", i), a.te_dim);
        decl_linear(S("te.sa%d.ffn1", i), a.te_ffn, a.te_dim, 1.f, false);
        decl_linear(S("te.sa%d.ffn2", i), a.te_dim, a.te_ffn, 1.f, false);
    }
    for (int i = 0; i < a.te_style_blocks; ++i) decl_attn(S("te.st%d", i), a.te_dim, a.d_style_ttl, false);
    decl_ln("te.out_ln", a.te_dim);
    decl_linear("te.proj", a.te_out_dim, a.te_dim, 1.f, false);
    // vector estimator
    decl_linear("ve.in", a.ve_dim, D, 1.f, true);
    {   // K of the input projection padded to a multiple of 64 with zero columns so it runs on the LDS-DMA GEMM path
        const int Dp = (D + 63) / 64 * 64;
        const std::vector<float>& w = host.at("ve.in.w");
        std::vector<float> wp((size_t)a.ve_dim * Dp, 0.f);
        for (int n = 0; n < a.ve_dim; ++n) std::copy(w.begin() + (size_t)n * D, w.begin() + (size_t)(n + 1) * D, wp.begin() + (size_t)n * Dp);
        upload("ve.in_pad.w", wp, a.ve_dim, Dp, true);
        upload("ve.in_pad.b", host.at("ve.in.b"), 1, a.ve_dim, false);
    }
    decl_linear("ve.t1", a.ve_dim, a.ve_time_dim, 1.f, false);
    decl_linear("ve.t2", a.ve_dim, a.ve_dim, 1.f, false);
    std::vector<std::string> time_parts, text_k, style_k;
    for (int b = 0; b < a.ve_main_blocks; ++b) {
        for (int j = 0; j < a.ve_dilated; ++j) decl_convnext(S("ve.m%d.dil%d", b, j), a.ve_dim, a.ve_hidden, a.ve_kernel);
        decl_linear(S("ve.m%d.time", b), a.ve_dim, a.ve_dim, 1.f, true);
        time_parts.push_back(S("ve.m%d.time", b));
        decl_convnext(S("ve.m%d.cn_a", b), a.ve_dim, a.ve_hidden, a.ve_kernel);
        decl_attn(S("ve.m%d.text", b), a.ve_dim, a.te_out_dim, false);
        text_k.push_back(S("ve.m%d.text.k", b)); text_k.push_back(S("ve.m%d.text.v", b));
        decl_convnext(S("ve.m%d.cn_b", b), a.ve_dim, a.ve_hidden, a.ve_kernel);
        decl_attn(S("ve.m%d.style", b), a.ve_dim, a.d_style_ttl, false);
        style_k.push_back(S("ve.m%d.style.k", b)); style_k.push_back(S("ve.m%d.style.v", b));
    }
    concat("ve.time_all", time_parts, a.ve_dim);       // [nb*C][C]
    concat("ve.text_kv_all", text_k, a.te_out_dim);    // [nb*2C][Ce]: block b -> K rows b*2C.., V rows b*2C+C..
    concat("ve.style_kv_all", style_k, a.d_style_ttl); // [nb*2C][Ds]
    for (int j = 0; j < a.ve_tail_blocks; ++j) decl_convnext(S("ve.tail%d", j), a.ve_dim, a.ve_hidden, a.ve_kernel);
    decl_ln("ve.out_ln", a.ve_dim);
    decl_linear("ve.out", D, a.ve_dim, 1.f, false);
    // vocoder
    {
        const int ld = a.latent_dim, k = a.vo_in_kernel, C = a.vo_dim;
        if (names_only) names_only->push_back("vo.in.w");
        std::vector<float> w = names_only ? std::vector<float>((size_t)C * ld * k, 0.f) : src("vo.in.w", (int)K_W, C, ld * k, 1.f);  // canonical [C][ld][k]
        if (!names_only) params_ += (int64_t)w.size();
        std::vector<float> wt(w.size());                                     // stored [ld*k][C]
        for (int co = 0; co < C; ++co) for (int i = 0; i < ld * k; ++i) wt[(size_t)i * C + co] = w[(size_t)co * ld * k + i];
        upload("vo.in.wt", wt, ld * k, C, false);
        // GEMM form of the same conv: [C][ld*k] with K zero-padded to a multiple of 64 (im2col rows are padded alike)
        const int kp = (ld * k + 63) / 64 * 64;
        std::vector<float> wp((size_t)C * kp, 0.f);
        for (int co = 0; co < C; ++co) std::copy(w.begin() + (size_t)co * ld * k, w.begin() + (size_t)(co + 1) * ld * k, wp.begin() + (size_t)co * kp);
        upload("vo.in_gemm.w", wp, C, kp, true);
        decl("vo.in.b", K_BIAS, 1, C, 1.f, false, false);
    }
    for (int i = 0; i < a.vo_blocks; ++i) decl_convnext(S("vo.blk%d", i), a.vo_dim, a.vo_hidden, a.vo_kernel);
    decl_ln("vo.out_ln", a.vo_dim);
    decl_linear("vo.head", a.base_chunk_size, a.vo_dim, a.head_gain, false);
    if (!names_only) {
        loaded_ = true;
        prepare_xattn_weights();
        prepare_ffn_weights();
        prepare_vocoder_constants();
    }
}

// =================================================================================================
// profiling
// =================================================================================================
void Engine::prof_begin(const char* tag, double flops, double bytes) {
    prof_active_ = false;
    if (!prof_on_ || spans_.size() > 200000) return;
    std::string full = std::string(stage_) + "." + tag;
    if (g_launch_log.on) { log_family_ = full; g_launch_log.family = log_family_.c_str(); }
    if (!prof_filter_.empty() && full != prof_filter_) return;
    if (prof_every_ > 1 && (prof_seen_++ % prof_every_) != 0) return;  // sampled: a launch that carries events does not overlap its neighbours
    ProfSpan sp;
    sp.tag = std::move(full);
    sp.flops = flops;
    sp.bytes = bytes;
    auto get = [&]() {
        hipEvent_t e;
        if (!ev_pool_.empty()) { e = ev_pool_.back(); ev_pool_.pop_back(); }
        else STN_HIP(hipEventCreate(&e));
        return e;
    };
    sp.a = get();
    sp.b = get();
    spans_.push_back(sp);
    // every span wraps exactly one kernel launch: the events ride on that kernel's dispatch packet (kernels.hpp), so the
    // span is the kernel's own duration — what rocprofv3's kernel trace reports — not launch-to-launch stream time
    g_launch_ev.start = sp.a;
    g_launch_ev.stop = sp.b;
    prof_active_ = true;
}
void Engine::prof_end() {
    g_launch_log.family = "-";
    if (!prof_active_) return;
    if (g_launch_ev.start) {  // nothing was launched (empty problem): drop the span
        g_launch_ev = LaunchEvents{};
        ev_pool_.push_back(spans_.back().a);
        ev_pool_.push_back(spans_.back().b);
        spans_.pop_back();
    }
    prof_active_ = false;
}
void Engine::profile_reset() {
    sync();
    // (cached graphs stay: a shape is only ever captured with profiling off, so no graph holds a pooled event)
    for (auto& sp : spans_) { ev_pool_.push_back(sp.a); ev_pool_.push_back(sp.b); }
    spans_.clear();
    g_launch_log.entries.clear();
}
void Engine::launch_log_enable(bool on) { g_launch_log.on = on; g_launch_log.family = "-"; g_launch_log.entries.clear(); }
std::string Engine::launch_log() const {
    std::string out;
    for (const auto& e : g_launch_log.entries) {
        // kernel expression as written at the launch site: "(name<...>)" or "name" -> name
        std::string k = e.second;
        size_t b = 0;
        while (b < k.size() && (k[b] == '(' || k[b] == ' ' || k[b] == '&')) ++b;
        size_t en = b;
        while (en < k.size() && (isalnum((unsigned char)k[en]) || k[en] == '_' || k[en] == ':')) ++en;
        out += e.first + "\t" + k.substr(b, en - b) + "\n";
    }
    return out;
}
std::vector<std::pair<std::string, KernelStat>> Engine::profile_collect() {
    sync();
    std::unordered_map<std::string, KernelStat> acc;
    std::vector<std::string> order;
    for (auto& sp : spans_) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) { (void)hipGetLastError(); continue; }  // never recorded
        auto it = acc.find(sp.tag);
        if (it == acc.end()) { order.push_back(sp.tag); it = acc.emplace(sp.tag, KernelStat{}).first; }
        it->second.ms += ms;
        it->second.launches += 1;
        it->second.flops += sp.flops;
        it->second.bytes += sp.bytes;
    }
    std::vector<std::pair<std::string, KernelStat>> out;
    for (auto& t : order) out.emplace_back(t, acc[t]);
    return out;
}

// =================================================================================================
// building blocks
// =================================================================================================
void Engine::gemm(const char* tag, int dt, const void* A, int lda, const Linear& w, int M, Epilogue e) {
    if (!e.bias) e.bias = w.b;
    const double esz = is_half(dt) ? 2.0 : 4.0;
    if (prof_on_) {
        double out_b = (double)M * w.N * (e.mode == EPI_STORE ? (is_half(e.out_dtype) ? 2.0 : 4.0) : (e.mode == EPI_RESID ? 8.0 : 4.0));
        prof_begin(tag, 2.0 * M * (double)w.N * w.K, ((double)M * w.K + (double)w.N * w.K) * esz + out_b);
    }
    const int sk = gemm_splitk_factor(dt, M, w.N, w.K, e);
    if (sk > 1) {
        const Arena::Mark mk = ar_.mark();
        float* ws = f32_alloc((int64_t)sk * M * w.N);
        launch_gemm_splitk(s_, dt, A, lda, w.w.as(dt), w.K, M, w.N, w.K, e, sk, ws);
        ar_.release(mk);  // stream order: the reduction has been enqueued behind the splits
    } else {
        launch_gemm(s_, dt, A, lda, w.w.as(dt), w.K, M, w.N, w.K, e);
    }
    if (prof_on_) prof_end();
}

void* Engine::to_act(const float* src, int64_t n) {
    if (dt_ == F32) return const_cast<float*>(src);
    void* d = act_alloc(n);
    launch_cast(s_, dt_, src, n, d);
    return d;
}

// x <- (x + gamma * pw2(GELU(pw1(LN(dwconv(x)))))) * mask      (in place, x fp32 [B*L][C])
void Engine::convnext(const ConvNeXt& p, float* x, int B, int L, int C, int hid, int k, int dil, const int* len,
                      const int* conv_len, const float* rowvec, int rv_ld, const Ragged* rg, FoldState* fs) {
    const int64_t M = rg ? (int64_t)rg->rows : (int64_t)B * L;
    const Arena::Mark mk = ar_.mark();
    void* xn = act_alloc(M * C);
    if (fs) x = fs->x;
    if (fs && fs->pending) {
        // the previous block's pointwise pair is still a set of partial sums: this block's conv kernel folds it on the way in
        if (prof_on_) prof_begin("fold_dwconv_ln", (double)M * C * (2.0 * k + 8 + 2.0 * fs->fold.S), (double)M * C * (8.0 + 2.0 + 2.0 * fs->fold.S));
        fs->fold.run_frames = rg->fold_run;
        launch_fold_dwconv_ln(s_, dt_, fs->x, fs->x_alt, B, L, C, fs->fold, p.dw_t, p.dw_b, k, dil, p.ln.g, p.ln.b, a_.ln_eps, xn, len, rg->off);
        if (prof_on_) prof_end();
        std::swap(fs->x, fs->x_alt);
        fs->pending = false;
        x = fs->x;
    } else {
        if (prof_on_) prof_begin("dwconv_ln", (double)M * C * (2.0 * k + 8), (double)M * C * (4.0 + (is_half(dt_) ? 2.0 : 4.0)));
        launch_dwconv_ln(s_, dt_, x, B, L, C, p.dw_t, p.dw_b, k, dil, p.ln.g, p.ln.b, a_.ln_eps, xn, rg ? len : conv_len, rg ? rg->off : nullptr);
        if (prof_on_) prof_end();
    }
    // K4: pw1 -> GELU -> pw2 -> layer scale + residual in one launch, the hidden activation never leaves the registers
    const int stage_bit = stage_[0] == 'v' && stage_[1] == 'o' ? 1 : (stage_[0] == 'v' ? 2 : 4);
    const auto fw = ffn_w_.find(p.pw1.w.as(dt_));
    // K4-split (the estimator at batch size: 59 slabs of 128 rows cannot fill 256 CUs, and a workgroup that streams both matrices
    // for 128 rows is ingest-bound): four workgroups per slab, each over a quarter of the hidden units (a quarter of the weight
    // stream), 16-bit partial sums; b2, layer scale, residual and time vector are applied by the next reader of x.  How many
    // ways is ffn_split_choose's decision on the launch's row count (fs->S).  Packed rows only (the fold kernels index sequences through row_off).
    if (fs && fs->S > 1 && rg && (fused_ffn_ & 8) && stage_bit == 2 && fw != ffn_w_.end() && fw->second.wsplit[split_slot(fs->S)] && M >= ffn_split_min_rows_ &&
        M * C * 2 < 0x7FFFFFFFll && fold_dwconv_ln_supported(C, k, 1 << std::max(0, a_.ve_dilated - 1))) {
        FfnArgs fa;
        fa.xn = xn; fa.ldx = C; fa.wseq = fw->second.wsplit[split_slot(fs->S)]; fa.b1 = p.pw1.b; fa.M = (int)M; fa.I = hid;
        fa.split = fs->S; fa.part = fs->part; fa.part_stride = fs->part_stride;
        if (prof_on_) prof_begin("ffn_split", 4.0 * M * (double)C * hid, (double)M * C * (2.0 + 2.0 * fa.split) + 4.0 * C * hid);
        launch_ffn_fused(s_, dt_, C, fa);
        if (prof_on_) prof_end();
        fs->pending = true;
        fs->fold = FoldArgs{};
        fs->fold.part = fs->part; fs->fold.S = fa.split; fs->fold.part_stride = fs->part_stride;
        fs->fold.b2 = p.pw2.b; fs->fold.gamma = p.gamma; fs->fold.rowvec = rowvec; fs->fold.rv_ld = rv_ld; fs->fold.row_b = rowvec ? rg->row_b : nullptr;
        ar_.release(mk);
        return;
    }
    // ... where it pays: a workgroup streams both weight matrices whatever its share of the rows, so below ~half a chip of
    // 128-row workgroups the two tiled launches win (tools/ffn_bench.py sweep, C = 512: 16384 rows 108 vs 107 us, 20480 rows
    // 116 vs 141 us, 294 rows = one utterance 102 vs 29 us)
    // (the vocoder decides on its DENSE frame count: a run that skips position-independent padding rows must take the same kernel
    // as the dense run it is bit-identical to)
    if ((fused_ffn_ & stage_bit) && (ffn_gate_rows_ > 0 ? ffn_gate_rows_ : M) >= ffn_min_rows_ && fw != ffn_w_.end() && ffn_fused_supported(dt_, C, hid) && M * C * 2 < 0x7FFFFFFFll) {
        FfnArgs fa;
        fa.xn = xn; fa.ldx = C; fa.wseq = fw->second.wseq; fa.b1 = p.pw1.b; fa.b2 = p.pw2.b; fa.gamma = p.gamma;
        fa.x = x; fa.ldo = C; fa.M = (int)M; fa.I = hid; fa.rowvec = rowvec; fa.rv_ld = rv_ld;
        fa.row_b = (rg && rowvec) ? rg->row_b : nullptr;
        fa.len = rg ? nullptr : len; fa.L = L;
        if (prof_on_) prof_begin("ffn_fused", 4.0 * M * (double)C * hid, (double)M * C * (2.0 + 8.0) + 4.0 * C * hid);
        launch_ffn_fused(s_, dt_, C, fa);
        if (prof_on_) prof_end();
        ar_.release(mk);
        return;
    }
    void* u = act_alloc(M * hid);
    Epilogue e1;
    e1.mode = EPI_STORE; e1.act = gelu_act_; e1.out_dtype = dt_; e1.out = u; e1.ldo = hid;
    // a hidden activation larger than half the 256 MB Infinity Cache (the vocoder's: 245 MB per block at C3) is written once:
    // non-temporal stores keep it from evicting the residual stream and the LayerNorm output (vo.pw1 181 -> 162 us).  Measured
    // and rejected: non-temporal A loads in pw2 (+9 %: each panel is read by two column tiles); running pw1/pw2 slab by slab
    // over the rows through a cache-sized hidden buffer (2 / 3 / 4 / 6 slabs: vocoder stage 4.08 -> 4.71 / 5.25 / 4.66 / 5.82 ms)
    if (nt_hints_ && is_half(dt_) && (double)M * hid * 2.0 > 128e6) e1.nt = 1;
    gemm("gemm_pw1_gelu", dt_, xn, C, p.pw1, (int)M, e1);
    Epilogue e2;
    e2.mode = EPI_RESID; e2.resid = x; e2.ldo = C; e2.gamma = p.gamma; e2.len = rg ? nullptr : len; e2.L = L; e2.rowvec = rowvec; e2.rv_ld = rv_ld;
    e2.row_b = (rg && rowvec) ? rg->row_b : nullptr;
    gemm("gemm_pw2_resid", dt_, u, hid, p.pw2, (int)M, e2);
    ar_.release(mk);
}

void Engine::fold_layernorm(FoldState& fs, int64_t M, int C, const LNorm& ln, void* xn, const char* tag) {
    const size_t esz = is_half(dt_) ? 2 : 4;
    if (fs.pending) {
        if (prof_on_) prof_begin("fold_ln", (double)M * C * (8 + 2.0 * fs.fold.S), (double)M * C * (8.0 + esz + 2.0 * fs.fold.S));
        launch_fold_ln(s_, dt_, fs.x, M, C, fs.fold, ln.g, ln.b, a_.ln_eps, xn);
        if (prof_on_) prof_end();
        fs.pending = false;
        return;
    }
    if (tag && prof_on_) prof_begin(tag, (double)M * C * 8, (double)M * C * (4.0 + esz));
    launch_layernorm(s_, dt_, fs.x, M, C, ln.g, ln.b, a_.ln_eps, xn);
    if (tag && prof_on_) prof_end();
}

// x <- (x + Wo attn(LN(x) Wq, ctx Wk, ctx Wv)) * mask.   self: ctx = LN(x), one fused QKV GEMM.
void Engine::attn_block(const Attn& p, float* x, int B, int Lq, int C, int H, const void* ctx, int Lk, const int* qlen,
                        const int* klen, int rope_mode, bool self, const Ragged* qrg) {
    // qrg: the query rows (and, for self-attention, the key rows too) are packed; a cross-attention context stays dense
    const int64_t Mq = qrg ? (int64_t)qrg->rows : (int64_t)B * Lq, Mk = (int64_t)B * Lk;
    const Arena::Mark mk = ar_.mark();
    const size_t esz = is_half(dt_) ? 2 : 4;
    void* xn = act_alloc(Mq * C);
    if (prof_on_) prof_begin("layernorm", (double)Mq * C * 8, (double)Mq * C * (4.0 + esz));
    launch_layernorm(s_, dt_, x, Mq, C, p.ln.g, p.ln.b, a_.ln_eps, xn);
    if (prof_on_) prof_end();
    const char *q = nullptr, *k = nullptr, *v = nullptr;
    int ldq = C, ldk = 2 * C;
    if (self) {
        void* qkv = act_alloc(Mq * 3 * C);
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = dt_; e.out = qkv; e.ldo = 3 * C;
        gemm("gemm_qkv", dt_, xn, C, p.qkv, (int)Mq, e);
        q = static_cast<const char*>(qkv); k = q + (size_t)C * esz; v = q + (size_t)2 * C * esz;
        ldq = ldk = 3 * C;
    } else {
        void* qb = act_alloc(Mq * C);
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = dt_; e.out = qb; e.ldo = C;
        gemm("gemm_q", dt_, xn, C, p.q, (int)Mq, e);
        q = static_cast<const char*>(qb);
        void* kv = act_alloc(Mk * 2 * C);
        Epilogue e2; e2.mode = EPI_STORE; e2.out_dtype = dt_; e2.out = kv; e2.ldo = 2 * C;
        gemm("gemm_kv", dt_, ctx, p.kv.K, p.kv, (int)Mk, e2);
        k = static_cast<const char*>(kv);
        v = k + (size_t)C * esz;
    }
    void* o = act_alloc(Mq * C);
    if (prof_on_) prof_begin("attention", 4.0 * Mq * (double)Lk * C, (double)(Mq * 2 + Mk * 2) * C * esz);
    launch_attention(s_, dt_, q, ldq, k, v, ldk, o, C, B, Lq, Lk, H, C / H, qlen, klen, rope_mode, a_.rope_base, a_.larope_gamma,
                     false, qrg ? qrg->off : nullptr, (qrg && self) ? qrg->off : nullptr);
    if (prof_on_) prof_end();
    Epilogue eo; eo.mode = EPI_RESID; eo.resid = x; eo.ldo = C; eo.len = qrg ? nullptr : qlen; eo.L = Lq;
    gemm("gemm_attn_out", dt_, o, C, p.o, (int)Mq, eo);
    ar_.release(mk);
}

// =================================================================================================
// stages (device level)
// =================================================================================================
void Engine::duration_dev(int B, int Lt, const int64_t* ids, const float* style_dp, const int* tlen, float* dur, const Ragged* trg) {
    stage_ = "dp";
    // The predicted durations define L, every latent length, the returned durations and the audio trim point: the predictor
    // always runs in exact fp32 (fp32 masters of its weights, exact-fp32 MFMA), whatever the engine's 16-bit mode, so that a
    // bf16 / f16 engine returns the utterance lengths of the fp32 reference.  It is ~1 % of a batch.
    struct DtGuard { int& r; int saved; DtGuard(int& x) : r(x), saved(x) { r = F32; } ~DtGuard() { r = saved; } } dt_guard(dt_);
    const stn_arch& a = a_;
    const int C = a.dp_dim;
    const int64_t M = trg ? (int64_t)trg->rows : (int64_t)B * Lt;
    const int* toff = trg ? trg->off : nullptr;
    const Arena::Mark mk = ar_.mark();
    float* x = f32_alloc(M * C);
    launch_embed(s_, ids, vecf("dp.emb"), a.vocab_size, B, Lt, C, tlen, x, toff);
    for (int i = 0; i < a.dp_conv_blocks; ++i)
        convnext(convnext_w("dp.conv" + std::to_string(i)), x, B, Lt, C, a.dp_hidden, a.dp_kernel, 1, tlen, nullptr, nullptr, 0, trg);
    void* st = to_act(style_dp, (int64_t)B * a.n_style_dp * a.d_style_dp);
    attn_block(attn_w("dp.st", false), x, B, Lt, C, a.dp_heads, st, a.n_style_dp, tlen, nullptr, -1, false, trg);
    float* xn = f32_alloc(M * C);
    const LNorm ln = lnorm("dp.out_ln");
    launch_layernorm(s_, F32, x, M, C, ln.g, ln.b, a.ln_eps, xn);
    float* pooled = f32_alloc((int64_t)B * C);
    launch_masked_mean(s_, F32, xn, B, Lt, C, tlen, pooled, toff);
    float* h = f32_alloc((int64_t)B * C);
    Epilogue e1; e1.mode = EPI_STORE; e1.act = gelu_act_; e1.out_dtype = F32; e1.out = h; e1.ldo = C;
    gemm("gemm_small_f32", F32, pooled, C, linear("dp.fc1"), B, e1);
    Epilogue e2; e2.mode = EPI_STORE; e2.out_dtype = F32; e2.out = dur; e2.ldo = 1;
    gemm("gemm_small_f32", F32, h, C, linear("dp.fc2"), B, e2);
    launch_softplus(s_, dur, B);
    ar_.release(mk);
}

void Engine::text_enc_dev(int B, int Lt, const int64_t* ids, const float* style_ttl, const int* tlen, float* ncl,
                          void* rows, const Ragged* trg) {
    stage_ = "te";
    const stn_arch& a = a_;
    const int C = a.te_dim, Ce = a.te_out_dim;
    if (trg && ncl) throw std::runtime_error("text_enc_dev: the [B,Ce,Lt] output needs the padded layout");
    const int64_t M = trg ? (int64_t)trg->rows : (int64_t)B * Lt;
    const int* rmask = trg ? nullptr : tlen;  // packed rows need no row mask
    const Arena::Mark mk = ar_.mark();
    float* x = f32_alloc(M * C);
    launch_embed(s_, ids, vecf("te.emb"), a.vocab_size, B, Lt, C, tlen, x, trg ? trg->off : nullptr);
    for (int i = 0; i < a.te_conv_blocks; ++i)
        convnext(convnext_w("te.conv" + std::to_string(i)), x, B, Lt, C, a.te_hidden, a.te_kernel, 1, tlen, nullptr, nullptr, 0, trg);
    for (int i = 0; i < a.te_attn_blocks; ++i) {
        const std::string p = "te.sa" + std::to_string(i);
        attn_block(attn_w(p, true), x, B, Lt, C, a.te_heads, nullptr, Lt, tlen, tlen, 0, true, trg);
        const Arena::Mark m2 = ar_.mark();
        void* xn = act_alloc(M * C);
        void* u = act_alloc(M * a.te_ffn);
        const LNorm ln = lnorm(p + ".ffn_ln");
        launch_layernorm(s_, dt_, x, M, C, ln.g, ln.b, a.ln_eps, xn);
        Epilogue e1; e1.mode = EPI_STORE; e1.act = gelu_act_; e1.out_dtype = dt_; e1.out = u; e1.ldo = a.te_ffn;
        gemm("gemm_pw1_gelu", dt_, xn, C, linear(p + ".ffn1"), (int)M, e1);
        Epilogue e2; e2.mode = EPI_RESID; e2.resid = x; e2.ldo = C; e2.len = rmask; e2.L = Lt;
        gemm("gemm_pw2_resid", dt_, u, a.te_ffn, linear(p + ".ffn2"), (int)M, e2);
        ar_.release(m2);
    }
    void* st = to_act(style_ttl, (int64_t)B * a.n_style_ttl * a.d_style_ttl);
    for (int i = 0; i < a.te_style_blocks; ++i)
        attn_block(attn_w("te.st" + std::to_string(i), false), x, B, Lt, C, a.te_heads, st, a.n_style_ttl, tlen, nullptr,
                   -1, false, trg);
    void* xn = act_alloc(M * C);
    const LNorm ln = lnorm("te.out_ln");
    launch_layernorm(s_, dt_, x, M, C, ln.g, ln.b, a.ln_eps, xn);
    const Linear proj = linear("te.proj");
    if (ncl) {
        Epilogue e; e.mode = EPI_STORE_T; e.out = ncl; e.len = tlen; e.L = Lt;
        gemm("gemm_proj", dt_, xn, C, proj, (int)M, e);
    }
    if (rows) {
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = dt_; e.out = rows; e.ldo = Ce; e.len = rmask; e.L = Lt;
        gemm("gemm_proj", dt_, xn, C, proj, (int)M, e);
    }
    ar_.release(mk);
}

// K/V of the text and style contexts for every main block: invariant across Euler steps.
// The returned buffers live in the arena ABOVE the caller's mark: the caller releases them.
Engine::VeCtx Engine::ve_prepare_dev(int B, int Lt, const void* text_rows, const float* style_ttl, const int* tlen, const Ragged* trg,
                                     bool defer_text) {
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, nb = a.ve_main_blocks;
    VeCtx c;
    c.Lt = Lt;
    const int64_t Mt = trg ? (int64_t)trg->rows : (int64_t)B * Lt;  // text rows: packed or padded
    c.text_off = trg ? trg->off : nullptr;
    c.text_kv = act_alloc(Mt * nb * 2 * C);
    c.style_kv = act_alloc((int64_t)B * a.n_style_ttl * nb * 2 * C);
    if (!defer_text) ve_text_kv_dev(c, B, Lt, text_rows, tlen, trg);
    void* st = to_act(style_ttl, (int64_t)B * a.n_style_ttl * a.d_style_ttl);
    Epilogue e2; e2.mode = EPI_STORE; e2.out_dtype = dt_; e2.out = c.style_kv; e2.ldo = nb * 2 * C;
    gemm("gemm_kv", dt_, st, a.d_style_ttl, linear("ve.style_kv_all"), B * a.n_style_ttl, e2);
    return c;
}
// K and V of the text context for all main blocks in one GEMM, keys rotated once (they depend neither on the Euler step nor on the query)
void Engine::ve_text_kv_dev(const VeCtx& c, int B, int Lt, const void* text_rows, const int* tlen, const Ragged* trg) {
    const char* saved = stage_;
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, nb = a.ve_main_blocks;
    const int64_t Mt = trg ? (int64_t)trg->rows : (int64_t)B * Lt;
    Epilogue e; e.mode = EPI_STORE; e.out_dtype = dt_; e.out = c.text_kv; e.ldo = nb * 2 * C;
    gemm("gemm_kv", dt_, text_rows, a.te_out_dim, linear("ve.text_kv_all"), (int)Mt, e);
    // LARoPE of the text keys: rotate all blocks' keys once (was re-done by every one of the main_blocks x total_step attention launches)
    launch_rope_rows(s_, dt_, c.text_kv, nb * 2 * C, B, Lt, tlen, nb, 2 * C, a.ve_heads, C / a.ve_heads, 1, a.rope_base, a.larope_gamma,
                     c.text_off);
    stage_ = saved;
}

// sinusoid(t * scale) -> Linear -> SiLU -> Linear -> per-block Linear, for `rows` independent (current, total) pairs
float* Engine::ve_time_cond_dev(int rows, const float* total_step, const float* current_step, float* tb_out) {
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, nb = a.ve_main_blocks;
    float* tb = tb_out ? tb_out : f32_alloc((int64_t)rows * nb * C);  // (arena: stays allocated for the caller)
    const Arena::Mark mk = ar_.mark();
    float* te = f32_alloc((int64_t)rows * a.ve_time_dim);
    launch_time_embed(s_, current_step, total_step, rows, a.ve_time_dim, a.time_scale, te);
    float* t1 = f32_alloc((int64_t)rows * C);
    Epilogue et1; et1.mode = EPI_STORE; et1.act = ACT_SILU; et1.out_dtype = F32; et1.out = t1; et1.ldo = C;
    gemm("gemm_small_f32", F32, te, a.ve_time_dim, linear("ve.t1"), rows, et1);
    float* tc = f32_alloc((int64_t)rows * C);
    Epilogue et2; et2.mode = EPI_STORE; et2.out_dtype = F32; et2.out = tc; et2.ldo = C;
    gemm("gemm_small_f32", F32, t1, C, linear("ve.t2"), rows, et2);
    Epilogue et3; et3.mode = EPI_STORE; et3.out_dtype = F32; et3.out = tb; et3.ldo = nb * C;
    gemm("gemm_small_f32", F32, tc, C, linear("ve.time_all"), rows, et3);
    ar_.release(mk);  // te/t1/tc are dead once the three GEMMs above have run (stream order)
    return tb;
}

void Engine::ve_step_dev(int B, int L, const VeCtx& c, const float* noisy, const int* tlen, const int* llen,
                         const float* total_step, const float* current_step, float* denoised, const float* tb, const Ragged* rg,
                         const float* dt, void* z_rows, bool z_ready, bool z_next) {
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, D = a.latent_dim * a.chunk_compress_factor, nb = a.ve_main_blocks, H = a.ve_heads;
    const int64_t M = rg ? (int64_t)rg->rows : (int64_t)B * L;  // packed: only the frames the utterances own
    const int* rmask = rg ? nullptr : llen;                     // padded rows are re-zeroed by every residual epilogue
    const int* roff = rg ? rg->off : nullptr;
    const size_t esz = is_half(dt_) ? 2 : 4;
    const Arena::Mark mk = ar_.mark();
    const int Dp = (D + 63) / 64 * 64;
    // the latent as rows for the input projection: the caller's persistent buffer when the step before left it there (euler_ncl writes both layouts)
    void* z = z_rows ? z_rows : act_alloc(M * Dp);
    if (!(z_rows && z_ready)) launch_ncl_to_rows(s_, dt_, noisy, B, D, L, z, Dp, llen, roff);
    FoldState fs;  // the residual stream (and, with K4-split blocks, its pending update)
    fs.x = f32_alloc(M * C);
    {
        const int S = ffn_split_choose(dt_, C, a.ve_hidden, M);
        if (rg && S > 1 && (fused_ffn_ & 8) && M >= ffn_split_min_rows_) {
            fs.x_alt = f32_alloc(M * C);
            fs.part_stride = ffn_split_rows(M) * C;
            fs.part = act_alloc(fs.part_stride * S);
            fs.S = S;
        }
    }
    FoldState* const fsp = fs.part ? &fs : nullptr;
    Epilogue ein; ein.mode = EPI_STORE; ein.out_dtype = F32; ein.out = fs.x; ein.ldo = C; ein.len = rmask; ein.L = L;
    gemm("gemm_in", dt_, z, Dp, linear("ve.in_pad"), (int)M, ein);
    if (!tb) tb = ve_time_cond_dev(B, total_step, current_step);

    auto cross = [&](const std::string& p, const void* kv_all, int blk, int Lk, const int* klen, int rope_mode) {
        // q from x, K/V precomputed (columns blk*2C .. of kv_all, row stride nb*2C)
        const Attn w = attn_w(p, false);
        const char* kp0 = static_cast<const char*>(kv_all) + (size_t)blk * 2 * C * esz;
        const auto fq = frag_w_.find(w.q.w.as(dt_));
        const auto foa = frag_acc_w_.find(w.o.w.as(dt_));
        if (fused_xattn_ && fsp && unit_vec_ && C <= 1024 && fq != frag_w_.end() && foa != frag_acc_w_.end() && xattn_hs_supported(dt_, C, H, L, Lk, nb * 2 * C) &&
            M * C * 2 < 0x7FFFFFFFll) {
            // HEAD-SPLIT: fold_ln (or LayerNorm), then ONE launch per block — q projection, rotation, attention and the head's share of the
            // output projection per (utterance pair, head), stored as four 16-bit per-head partial sums in K4-split's layout; the next
            // ConvNeXt block's fold_dwconv_ln adds them (and the output bias) to x in head order
            const Arena::Mark m2 = ar_.mark();
            void* xn = act_alloc(M * C);
            fold_layernorm(fs, M, C, w.ln, xn, "layernorm");
            if (kv_all == c.text_kv && text_gate_) { auto fire = std::move(text_gate_); text_gate_ = nullptr; fire(); }
            if (prof_on_) prof_begin("xattn_hs", 4.0 * M * (double)C * C + 4.0 * M * (double)Lk * C,
                                     (double)M * C * (esz + 2.0 * H) + 2.0 * C * C * esz + (double)B * Lk * 2 * C * esz);
            unsigned long long* ts = nullptr;
            if (hs_ts_ && (int64_t)(B + 8) * H <= HS_TS_WG) { ts = hs_ts_; hs_ts_wgs_ = (B + 8) * H; }  // (an upper bound of the grid)
            launch_xattn_hs(s_, dt_, xn, M, fq->second, w.q.b, kp0, kp0 + (size_t)C * esz, nb * 2 * C, foa->second, fs.part, fs.part_stride, B, L, Lk,
                            llen, klen, roff, kv_all == c.text_kv ? c.text_off : nullptr, rope_mode, a.rope_base, a.larope_gamma, ts, rg->hs_pairs);
            if (prof_on_) prof_end();
            fs.pending = true;
            fs.fold = FoldArgs{};
            fs.fold.part = fs.part; fs.fold.S = H; fs.fold.part_stride = fs.part_stride;
            fs.fold.b2 = w.o.b ? w.o.b : unit_vec_ + 1024; fs.fold.gamma = unit_vec_;  // x + 1 * (sum + bo): the product with 1 is exact
            ar_.release(m2);
            return;
        }
        const Arena::Mark m2 = ar_.mark();
        void* xn = act_alloc(M * C);
        fold_layernorm(fs, M, C, w.ln, xn, "layernorm");
        float* const x = fs.x;
        void* qb = act_alloc(M * C);
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = dt_; e.out = qb; e.ldo = C;
        gemm("gemm_q", dt_, xn, C, w.q, (int)M, e);
        const char* kp = static_cast<const char*>(kv_all) + (size_t)blk * 2 * C * esz;
        const char* vp = kp + (size_t)C * esz;
        void* o = act_alloc(M * C);
        // the first text cross-attention of a resident-batch run is where the text encoder's rows are needed: take them (and
        // compute the text K/V) here, not at the head of the pipeline, so that everything before runs beside the encoder
        if (kv_all == c.text_kv && text_gate_) { auto fire = std::move(text_gate_); text_gate_ = nullptr; fire(); }
        if (prof_on_) prof_begin("attention", 4.0 * M * (double)Lk * C, (double)(M * 2 + (int64_t)B * Lk * 2) * C * esz);
        launch_attention(s_, dt_, qb, C, kp, vp, nb * 2 * C, o, C, B, L, Lk, H, C / H, llen, klen, rope_mode, a.rope_base,
                         a.larope_gamma, /*k_rotated=*/rope_mode >= 0, roff, kv_all == c.text_kv ? c.text_off : nullptr);
        if (prof_on_) prof_end();
        Epilogue eo; eo.mode = EPI_RESID; eo.resid = x; eo.ldo = C; eo.len = rmask; eo.L = L;
        gemm("gemm_attn_out", dt_, o, C, w.o, (int)M, eo);
        ar_.release(m2);
    };

    for (int blk = 0; blk < nb; ++blk) {
        const std::string p = "ve.m" + std::to_string(blk);
        // the time conditioning x += tb[b] rides in the residual epilogue of the last dilated block (was a separate pass)
        for (int j = 0; j < a.ve_dilated; ++j) {
            const bool last = j == a.ve_dilated - 1;
            convnext(convnext_w(p + ".dil" + std::to_string(j)), fs.x, B, L, C, a.ve_hidden, a.ve_kernel, 1 << j, llen, nullptr,
                     last ? tb + (size_t)blk * C : nullptr, nb * C, rg, fsp);
        }
        if (a.ve_dilated == 0) launch_add_rowvec(s_, fs.x, tb + (size_t)blk * C, nb * C, B, L, C, llen);  // (padded layout only)
        convnext(convnext_w(p + ".cn_a"), fs.x, B, L, C, a.ve_hidden, a.ve_kernel, 1, llen, nullptr, nullptr, 0, rg, fsp);
        cross(p + ".text", c.text_kv, blk, c.Lt, tlen, 1);
        convnext(convnext_w(p + ".cn_b"), fs.x, B, L, C, a.ve_hidden, a.ve_kernel, 1, llen, nullptr, nullptr, 0, rg, fsp);
        cross(p + ".style", c.style_kv, blk, a.n_style_ttl, nullptr, -1);
    }
    for (int j = 0; j < a.ve_tail_blocks; ++j)
        convnext(convnext_w("ve.tail" + std::to_string(j)), fs.x, B, L, C, a.ve_hidden, a.ve_kernel, 1, llen, nullptr, nullptr, 0, rg, fsp);
    void* xn = act_alloc(M * C);
    fold_layernorm(fs, M, C, lnorm("ve.out_ln"), xn, nullptr);
    // Euler update fused into the output projection; dt[b] = 1 / total_step[b]
    const float* dtv = dt;
    if (!dtv) {
        float* d = f32_alloc(B);
        launch_reciprocal(s_, total_step, B, d);
        dtv = d;
    }
    // output projection on the vector-epilogue GEMM path, then the Euler update fused with the [B*L][D] -> [B][D][L] transpose
    float* vel = f32_alloc(M * D);
    Epilogue eo; eo.mode = EPI_STORE; eo.out_dtype = F32; eo.out = vel; eo.ldo = D;
    gemm("gemm_out", dt_, xn, C, linear("ve.out"), (int)M, eo);
    launch_euler_ncl(s_, noisy, vel, dtv, llen, B, D, L, denoised, roff, (z_rows && z_next) ? z_rows : nullptr, dt_, Dp);
    ar_.release(mk);
}

int Engine::vocoder_receptive_field() const {
    int rf = (a_.vo_in_kernel - 1) / 2;
    for (int i = 0; i < a_.vo_blocks; ++i) rf += (a_.vo_kernel - 1) / 2 * a_.vo_dilations[i];
    return rf;
}

// The vocoder's response to zero latent is position-independent away from data and edges.  One run on a short all-zero
// latent yields the two pieces every padded tail is made of: the frame whose whole receptive field is zero latent ("quiet"),
// and the last rf frames before the end of the tensor ("edge").  16-bit engines only (the packed vocoder path).
void Engine::hs_stamps_enable(bool on) {
    sync();
    drop_graphs();  // a captured pipeline has the stamp buffer (or its absence) baked in
    if (on && !hs_ts_) {
        void* d = nullptr;
        STN_HIP(hipMalloc(&d, sizeof(unsigned long long) * 8 * HS_TS_WG));
        STN_HIP(hipMemset(d, 0, sizeof(unsigned long long) * 8 * HS_TS_WG));
        hs_ts_ = static_cast<unsigned long long*>(d);
    } else if (!on && hs_ts_) {
        (void)hipFree(hs_ts_);
        hs_ts_ = nullptr;
    }
    hs_ts_wgs_ = 0;
}
int64_t Engine::hs_stamps_fetch(unsigned long long* out, size_t cap) {
    if (!hs_ts_) return 0;
    sync();
    const size_t n = std::min(cap, (size_t)hs_ts_wgs_ * 8);
    if (out && n) STN_HIP(hipMemcpy(out, hs_ts_, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return hs_ts_wgs_;
}

void Engine::prepare_xattn_weights() {
    frag_w_.clear();
    frag_acc_w_.clear();
    const stn_arch& a = a_;
    unit_vec_ = nullptr;
    if (!is_half(dt_) || !xattn_hs_supported(dt_, a.ve_dim, a.ve_heads, 1, 1, 8)) return;
    {
        std::vector<float> uv(2048, 0.f);
        std::fill(uv.begin(), uv.begin() + 1024, 1.f);
        void* d = nullptr;
        STN_HIP(hipMalloc(&d, uv.size() * sizeof(float)));
        owned_.push_back(d);
        STN_HIP(hipMemcpy(d, uv.data(), uv.size() * sizeof(float), hipMemcpyHostToDevice));
        unit_vec_ = static_cast<const float*>(d);
    }
    for (int blk = 0; blk < a.ve_main_blocks; ++blk)
        for (const char* kind : {".text", ".style"}) {
            const Attn w = attn_w("ve.m" + std::to_string(blk) + kind, false);
            {   // Wq in MFMA fragment order (a head's 96 rows are one contiguous 72 KiB)
                const void* src = w.q.w.as(dt_);
                void* dst = nullptr;
                STN_HIP(hipMalloc(&dst, (size_t)w.q.N * w.q.K * 2));
                owned_.push_back(dst);
                launch_repack_frag(s_, src, w.q.N, w.q.K, dst);
                frag_w_[src] = dst;
            }
            {   // Wo once more in the k order of an accumulator used as the B operand (the head-split block's output projection)
                const void* src = w.o.w.as(dt_);
                void* dst = nullptr;
                STN_HIP(hipMalloc(&dst, (size_t)w.o.N * w.o.K * 2));
                owned_.push_back(dst);
                launch_repack_frag_acc(s_, src, w.o.N, w.o.K, dst);
                frag_acc_w_[src] = dst;
            }
        }
    sync();
}

// Fragment-ordered copies of the pointwise matrices of every ConvNeXt block the fused kernel supports (kernels_ffn.hip):
// W1 [I][C] as phase-1 A fragments, W2 [C][I] as phase-2 A fragments in the accumulator-operand k order.
void Engine::prepare_ffn_weights() {
    ffn_w_.clear();
    const stn_arch& a = a_;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    auto add = [&](const std::string& p, int C, int hid) {
        if (!ffn_fused_supported(dt_, C, hid)) return;
        const ConvNeXt c = convnext_w(p);
        void* wseq = nullptr;
        STN_HIP(hipMalloc(&wseq, (size_t)2 * hid * C * 2));
        owned_.push_back(wseq);
        if ((size_t)2 * hid * C * 2 > tmp_bytes) {
            if (tmp) (void)hipFree(tmp);
            tmp_bytes = (size_t)2 * hid * C * 2;
            STN_HIP(hipMalloc(&tmp, tmp_bytes));
        }
        launch_ffn_pack(s_, c.pw1.w.as(dt_), c.pw2.w.as(dt_), C, hid, tmp, wseq);
        sync();  // tmp is reused by the next block
        FfnW fw; fw.wseq = wseq;
        if (p.compare(0, 3, "ve.") == 0)  // the estimator's blocks also as hidden-split stage streams, one copy per split
            for (int S : {4, 8, 12}) {  // (the splits ffn_split_choose hands out)
                if (!ffn_split_valid(dt_, C, hid, S)) continue;
                void* ws = nullptr;
                STN_HIP(hipMalloc(&ws, (size_t)2 * hid * C * 2));
                owned_.push_back(ws);
                launch_ffn_pack(s_, c.pw1.w.as(dt_), c.pw2.w.as(dt_), C, hid, tmp, ws, S);
                sync();
                fw.wsplit[split_slot(S)] = ws;
            }
        ffn_w_[c.pw1.w.as(dt_)] = fw;
    };
    auto S = [](const char* fmt, int i, int j = 0) { char b[64]; snprintf(b, sizeof b, fmt, i, j); return std::string(b); };
    for (int i = 0; i < a.dp_conv_blocks; ++i) add(S("dp.conv%d", i), a.dp_dim, a.dp_hidden);
    for (int i = 0; i < a.te_conv_blocks; ++i) add(S("te.conv%d", i), a.te_dim, a.te_hidden);
    for (int b = 0; b < a.ve_main_blocks; ++b) {
        for (int j = 0; j < a.ve_dilated; ++j) add(S("ve.m%d.dil%d", b, j), a.ve_dim, a.ve_hidden);
        add(S("ve.m%d.cn_a", b), a.ve_dim, a.ve_hidden);
        add(S("ve.m%d.cn_b", b), a.ve_dim, a.ve_hidden);
    }
    for (int j = 0; j < a.ve_tail_blocks; ++j) add(S("ve.tail%d", j), a.ve_dim, a.ve_hidden);
    for (int i = 0; i < a.vo_blocks; ++i) add(S("vo.blk%d", i), a.vo_dim, a.vo_hidden);
    sync();
    if (tmp) (void)hipFree(tmp);
}

void Engine::prepare_vocoder_constants() {
    if (vo_quiet_) { (void)hipFree(vo_quiet_); vo_quiet_ = nullptr; }
    if (vo_edge_) { (void)hipFree(vo_edge_); vo_edge_ = nullptr; }
    vo_rf_ = 0;
    const stn_arch& a = a_;
    if (!is_half(dt_) || !dwconv_ln_supports_packed(a.vo_dim, a.vo_kernel) || a.base_chunk_size % 4) return;
    const int rf = vocoder_receptive_field(), ccf = a.chunk_compress_factor, W = a.base_chunk_size;
    const int Lz = (4 * rf + ccf) / ccf + 1, Tz = Lz * ccf, D = a.latent_dim * ccf;
    ar_.reset();
    float* lat = f32_alloc((int64_t)D * Lz);
    float* wz = f32_alloc((int64_t)Tz * W);
    STN_HIP(hipMemsetAsync(lat, 0, sizeof(float) * (size_t)D * Lz, s_));
    vocoder_dev(1, Lz, lat, wz);
    STN_HIP(hipMalloc(reinterpret_cast<void**>(&vo_quiet_), sizeof(float) * (size_t)W));
    STN_HIP(hipMalloc(reinterpret_cast<void**>(&vo_edge_), sizeof(float) * (size_t)rf * W));
    STN_HIP(hipMemcpyAsync(vo_quiet_, wz + (size_t)2 * rf * W, sizeof(float) * (size_t)W, hipMemcpyDeviceToDevice, s_));
    STN_HIP(hipMemcpyAsync(vo_edge_, wz + (size_t)(Tz - rf) * W, sizeof(float) * (size_t)rf * W, hipMemcpyDeviceToDevice, s_));
    sync();
    vo_rf_ = rf;
}

// host mirror of trim_len_kernel: the packed row count (0 when trimming does not apply to this batch)
int Engine::trimmed_rows(int B, int L, std::vector<int>* n_host) const {
    if (!packed_ve_ || vo_ragged_ || !vo_quiet_ || vo_rf_ <= 0 || B > 1024) return 0;
    const int ccf = a_.chunk_compress_factor, T = L * ccf, rf = vo_rf_;
    long tot = 0;
    bool any = false;
    for (int i = 0; i < B; ++i) {
        const int l6 = bt_.h_llen[i] * ccf;
        const bool trim = l6 + 2 * rf <= T;
        any = any || trim;
        const int n = trim ? l6 + 2 * rf : T;
        if (n_host) n_host->push_back(n);
        tot += n;
    }
    // the unpack pass costs about as much as 3 % of the frames: trim only when it removes clearly more than that
    return (any && tot * 10 <= (long)B * T * 9) ? (int)tot : 0;
}

// vlen == nullptr: every utterance is decoded over all T frames (the reference's batched vocoder Run: the padding is
// zero latent, which the convolutions see as signal).  vlen != nullptr (length-aware): convolution taps beyond an
// utterance's own length read as the zero padding of a batch-of-one run, so wav[b, :vlen[b]*hop] equals what
// synthesizing utterance b alone gives; samples past that are written as zeros.
void Engine::vocoder_dev(int B, int L, const float* latent, float* wav, const int* vlen, int vrows, const int* valid) {
    stage_ = "vo";
    const stn_arch& a = a_;
    const int C = a.vo_dim, T = L * a.chunk_compress_factor;
    // length-aware mode on packed rows: only the frames the utterances own exist (bf16 path; needs the comb dwconv kernel)
    const bool packed = vlen && vrows > 0 && is_half(dt_) && B <= 1024 && dwconv_ln_supports_packed(C, a.vo_kernel);
    if (valid && !packed) throw std::runtime_error("trimmed vocoder needs the packed bf16 path");
    const int64_t M = packed ? (int64_t)vrows : (int64_t)B * T;
    ffn_gate_rows_ = valid ? (int64_t)B * T : 0;
    const Arena::Mark mk = ar_.mark();
    Ragged rg;
    if (packed) {
        int* off = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)(B + 1)));
        launch_row_map(s_, vlen, B, off, nullptr);
        rg.off = off; rg.rows = vrows;
    }
    const Ragged* rgp = packed ? &rg : nullptr;
    float* x = f32_alloc(M * C);
    if (is_half(dt_)) {
        // input conv on the MFMA path: im2col (K = ld*k padded to 64) + GEMM; ~10x the direct fp32 VALU kernel
        const int kp = (a.latent_dim * a.vo_in_kernel + 63) / 64 * 64;
        void* cols = act_alloc(M * kp);
        launch_vocoder_im2col(s_, dt_, latent, B, L, a.latent_dim, a.chunk_compress_factor, a.vo_in_kernel, kp, cols, vlen,
                              packed ? rg.off : nullptr);
        Linear lin;
        lin.w = tensor("vo.in_gemm.w"); lin.b = vecf("vo.in.b"); lin.N = C; lin.K = kp;
        Epilogue ei; ei.mode = EPI_STORE; ei.out_dtype = F32; ei.out = x; ei.ldo = C;
        gemm("gemm_in", dt_, cols, kp, lin, (int)M, ei);
    } else {
        if (prof_on_) prof_begin("vocoder_in", 2.0 * M * C * a.latent_dim * a.vo_in_kernel, (double)M * (a.latent_dim + C) * 4.0);
        launch_vocoder_in(s_, latent, B, L, a.latent_dim, a.chunk_compress_factor, vecf("vo.in.wt"), vecf("vo.in.b"), C,
                          a.vo_in_kernel, x, vlen);
        if (prof_on_) prof_end();
    }
    for (int i = 0; i < a.vo_blocks; ++i)
        convnext(convnext_w("vo.blk" + std::to_string(i)), x, B, T, C, a.vo_hidden, a.vo_kernel, a.vo_dilations[i],
                 packed ? vlen : nullptr, packed ? nullptr : vlen, nullptr, 0, rgp);
    ffn_gate_rows_ = 0;
    void* xn = act_alloc(M * C);
    const LNorm ln = lnorm("vo.out_ln");
    launch_layernorm(s_, dt_, x, M, C, ln.g, ln.b, a.ln_eps, xn);
    // head: transposed conv with kernel = stride = base_chunk_size == per-frame linear; rows of the GEMM output ARE the wave
    if (packed) {
        float* wp = f32_alloc(M * a.base_chunk_size);
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = F32; e.out = wp; e.ldo = a.base_chunk_size;
        gemm("gemm_head", dt_, xn, C, linear("vo.head"), (int)M, e);
        if (valid) launch_unpack_rows_quiet(s_, wp, valid, rg.off, B, T, a.base_chunk_size, vo_rf_, vo_quiet_, vo_edge_, wav);
        else launch_unpack_rows(s_, wp, vlen, rg.off, B, T, a.base_chunk_size, wav);
    } else {
        Epilogue e; e.mode = EPI_STORE; e.out_dtype = F32; e.out = wav; e.ldo = a.base_chunk_size; e.len = vlen; e.L = T;
        gemm("gemm_head", dt_, xn, C, linear("vo.head"), (int)M, e);
    }
    ar_.release(mk);
}

// =================================================================================================
// host-pointer stages (the four former Run sites)
// =================================================================================================

void Engine::duration(int B, int Lt, const int64_t* ids, const float* style_dp, const float* text_mask, float* dur) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    int64_t* d_ids = up(ar_, s_, ids, (size_t)B * Lt);
    float* d_mask = up(ar_, s_, text_mask, (size_t)B * Lt);
    float* d_st = up(ar_, s_, style_dp, (size_t)B * a_.n_style_dp * a_.d_style_dp);
    int* tlen = static_cast<int*>(ar_.alloc(sizeof(int) * B));
    float* d_dur = f32_alloc(B);
    launch_mask_to_len(s_, d_mask, B, Lt, tlen);
    duration_dev(B, Lt, d_ids, d_st, tlen, d_dur);
    STN_HIP(hipMemcpyAsync(dur, d_dur, sizeof(float) * B, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::text_enc(int B, int Lt, const int64_t* ids, const float* style_ttl, const float* text_mask, float* text_emb) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    int64_t* d_ids = up(ar_, s_, ids, (size_t)B * Lt);
    float* d_mask = up(ar_, s_, text_mask, (size_t)B * Lt);
    float* d_st = up(ar_, s_, style_ttl, (size_t)B * a_.n_style_ttl * a_.d_style_ttl);
    int* tlen = static_cast<int*>(ar_.alloc(sizeof(int) * B));
    const size_t n = (size_t)B * a_.te_out_dim * Lt;
    float* d_out = f32_alloc(n);
    launch_mask_to_len(s_, d_mask, B, Lt, tlen);
    text_enc_dev(B, Lt, d_ids, d_st, tlen, d_out, nullptr);
    STN_HIP(hipMemcpyAsync(text_emb, d_out, n * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::vector_est(int B, int L, int Lt, const float* noisy, const float* text_emb, const float* style_ttl,
                        const float* text_mask, const float* latent_mask, const float* total_step,
                        const float* current_step, float* denoised) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const int D = a_.latent_dim * a_.chunk_compress_factor, Ce = a_.te_out_dim;
    float* d_x = up(ar_, s_, noisy, (size_t)B * D * L);
    float* d_emb = up(ar_, s_, text_emb, (size_t)B * Ce * Lt);
    float* d_st = up(ar_, s_, style_ttl, (size_t)B * a_.n_style_ttl * a_.d_style_ttl);
    float* d_tm = up(ar_, s_, text_mask, (size_t)B * Lt);
    float* d_lm = up(ar_, s_, latent_mask, (size_t)B * L);
    float* d_tot = up(ar_, s_, total_step, (size_t)B);
    float* d_cur = up(ar_, s_, current_step, (size_t)B);
    int* tlen = static_cast<int*>(ar_.alloc(sizeof(int) * B));
    int* llen = static_cast<int*>(ar_.alloc(sizeof(int) * B));
    float* d_out = f32_alloc((size_t)B * D * L);
    launch_mask_to_len(s_, d_tm, B, Lt, tlen);
    launch_mask_to_len(s_, d_lm, B, L, llen);
    void* rows = act_alloc((int64_t)B * Lt * Ce);
    launch_ncl_to_rows(s_, dt_, d_emb, B, Ce, Lt, rows);
    VeCtx c = ve_prepare_dev(B, Lt, rows, d_st, tlen);
    ve_step_dev(B, L, c, d_x, tlen, llen, d_tot, d_cur, d_out);
    STN_HIP(hipMemcpyAsync(denoised, d_out, (size_t)B * D * L * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::vocoder(int B, int L, const float* latent, float* wav) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const int D = a_.latent_dim * a_.chunk_compress_factor;
    const size_t nw = (size_t)B * L * a_.base_chunk_size * a_.chunk_compress_factor;
    float* d_lat = up(ar_, s_, latent, (size_t)B * D * L);
    float* d_wav = f32_alloc(nw);
    vocoder_dev(B, L, d_lat, d_wav);
    STN_HIP(hipMemcpyAsync(wav, d_wav, nw * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

}  // namespace stn
