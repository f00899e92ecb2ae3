// engine.cpp — weights, workspace and the four stage executors (see engine.hpp).
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace stn {

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
    // Stream priorities, experiment switch STN_PRIO=<main><side>, each h / n / l (default nn: all streams at the default priority).  Measured:
    // the main stream at the highest priority gains 0.6 % for one batch at a time (11.59 -> 11.52 ms) and LOSES a third of the rate with two
    // handles in flight (53 k -> 34 k audio-s/s: two highest-priority queues no longer interleave) — not adopted
    int prio_least = 0, prio_greatest = 0;
    STN_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const char* pr = getenv("STN_PRIO");
    auto prio_of = [&](char c) { return c == 'h' ? prio_greatest : c == 'l' ? prio_least : 0; };
    const int prio_main = pr && pr[0] ? prio_of(pr[0]) : 0, prio_side = pr && pr[0] && pr[1] ? prio_of(pr[1]) : 0;
    STN_HIP(hipStreamCreateWithPriority(&own_s_, hipStreamNonBlocking, prio_main));
    s_ = own_s_;
    STN_HIP(hipStreamCreateWithPriority(&dp_s_, hipStreamNonBlocking, prio_side));
    STN_HIP(hipEventCreateWithFlags(&ev_te_, hipEventDisableTiming));
    STN_HIP(hipEventCreateWithFlags(&ev_copied_, hipEventDisableTiming));
    STN_HIP(hipEventCreateWithFlags(&ev_dp_, hipEventDisableTiming));
    STN_HIP(hipStreamCreateWithPriority(&te_s_, hipStreamNonBlocking, prio_side));
    if (const char* p = getenv("STN_DP_STREAM")) if (atoi(p) == 0) {  // A/B switch: everything on the main stream
        (void)hipStreamDestroy(dp_s_); dp_s_ = nullptr;
        (void)hipStreamDestroy(te_s_); te_s_ = nullptr;
    }
    if (const char* p = getenv("STN_NT")) nt_hints_ = atoi(p) != 0;  // A/B switch: non-temporal hints on the vocoder's hidden activation
    if (const char* p = getenv("STN_FFN")) fused_ffn_ = atoi(p);          // A/B switch: K4 stage mask (1 vocoder, 2 estimator, 4 text stages)
    if (const char* p = getenv("STN_FFN_MIN_ROWS")) ffn_min_rows_ = atoll(p);
    if (const char* p = getenv("STN_XATTN")) set_fused_xattn(atoi(p));  // A/B switch: cross-attention blocks in one (1) or two (2) launches
    if (const char* p = getenv("STN_FFN_SPLIT_MIN_ROWS")) ffn_split_min_rows_ = atoll(p);
    if (const char* p = getenv("STN_PACKED")) packed_ve_ = atoi(p) != 0;  // A/B switch for measurements (stn_set_row_layout overrides)
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
    // stream), 16-bit partial sums; b2, layer scale, residual and time vector are applied by the next reader of x.  The split is a
    // function of the block's shape only.  Packed rows only (the fold kernels index sequences through row_off).
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
    e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = hid;
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
    Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = F32; e1.out = h; e1.ldo = C;
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
        Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = a.te_ffn;
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
float* Engine::ve_time_cond_dev(int rows, const float* total_step, const float* current_step) {
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, nb = a.ve_main_blocks;
    float* tb = f32_alloc((int64_t)rows * nb * C);  // stays allocated for the caller
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
                         const float* dt) {
    stage_ = "ve";
    const stn_arch& a = a_;
    const int C = a.ve_dim, D = a.latent_dim * a.chunk_compress_factor, nb = a.ve_main_blocks, H = a.ve_heads;
    const int64_t M = rg ? (int64_t)rg->rows : (int64_t)B * L;  // packed: only the frames the utterances own
    const int* rmask = rg ? nullptr : llen;                     // padded rows are re-zeroed by every residual epilogue
    const int* roff = rg ? rg->off : nullptr;
    const size_t esz = is_half(dt_) ? 2 : 4;
    const Arena::Mark mk = ar_.mark();
    const int Dp = (D + 63) / 64 * 64;
    void* z = act_alloc(M * Dp);
    launch_ncl_to_rows(s_, dt_, noisy, B, D, L, z, Dp, llen, roff);
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
        const auto fq = frag_w_.find(w.q.w.as(dt_)), fo = frag_w_.find(w.o.w.as(dt_));
        if (fused_xattn_ && fq != frag_w_.end() && fo != frag_w_.end() && xattn_fused_supported(dt_, C, H, Lk, nb * 2 * C)) {
            // fold (when the previous block left one), LayerNorm, q projection, attention, output projection and the residual add in ONE launch
            if (kv_all == c.text_kv && text_gate_) { auto fire = std::move(text_gate_); text_gate_ = nullptr; fire(); }
            if (fused_xattn_ == 2) {  // two launches, the q rows through a buffer
                const Arena::Mark m2 = ar_.mark();
                void* qb = act_alloc(M * C);
                if (prof_on_) prof_begin("xattn_q", 2.0 * M * (double)C * C, (double)M * C * (8.0 + esz) + (double)C * C * esz);
                launch_xattn_fused(s_, dt_, fs.x, w.ln.g, w.ln.b, a.ln_eps, fq->second, w.q.b, kp0, kp0 + (size_t)C * esz, nb * 2 * C, fo->second,
                                   w.o.b, B, L, C, H, Lk, llen, klen, roff, kv_all == c.text_kv ? c.text_off : nullptr, rope_mode, a.rope_base,
                                   a.larope_gamma, fs.pending ? &fs.fold : nullptr, 1, qb);
                if (prof_on_) prof_end();
                if (prof_on_) prof_begin("xattn_o", 2.0 * M * (double)C * C + 4.0 * M * (double)Lk * C, (double)M * C * (8.0 + esz) + (double)C * C * esz + (double)B * Lk * 2 * C * esz);
                launch_xattn_fused(s_, dt_, fs.x, w.ln.g, w.ln.b, a.ln_eps, fq->second, w.q.b, kp0, kp0 + (size_t)C * esz, nb * 2 * C, fo->second,
                                   w.o.b, B, L, C, H, Lk, llen, klen, roff, kv_all == c.text_kv ? c.text_off : nullptr, rope_mode, a.rope_base,
                                   a.larope_gamma, nullptr, 2, qb);
                if (prof_on_) prof_end();
                ar_.release(m2);
                fs.pending = false;
                return;
            }
            if (prof_on_) prof_begin("xattn_fused", 4.0 * M * (double)C * C + 4.0 * M * (double)Lk * C, (double)M * C * 8.0 + 2.0 * C * C * esz + (double)B * Lk * 2 * C * esz);
            launch_xattn_fused(s_, dt_, fs.x, w.ln.g, w.ln.b, a.ln_eps, fq->second, w.q.b, kp0, kp0 + (size_t)C * esz, nb * 2 * C, fo->second,
                               w.o.b, B, L, C, H, Lk, llen, klen, roff, kv_all == c.text_kv ? c.text_off : nullptr, rope_mode, a.rope_base,
                               a.larope_gamma, fs.pending ? &fs.fold : nullptr);
            if (prof_on_) prof_end();
            fs.pending = false;
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
    launch_euler_ncl(s_, noisy, vel, dtv, llen, B, D, L, denoised, roff);
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
void Engine::prepare_xattn_weights() {
    frag_w_.clear();
    const stn_arch& a = a_;
    if (!is_half(dt_) || !xattn_fused_supported(dt_, a.ve_dim, a.ve_heads, 1, 8)) return;
    for (int blk = 0; blk < a.ve_main_blocks; ++blk)
        for (const char* kind : {".text", ".style"}) {
            const Attn w = attn_w("ve.m" + std::to_string(blk) + kind, false);
            for (const Linear* lin : {&w.q, &w.o}) {
                const void* src = lin->w.as(dt_);
                void* dst = nullptr;
                STN_HIP(hipMalloc(&dst, (size_t)lin->N * lin->K * 2));
                owned_.push_back(dst);
                launch_repack_frag(s_, src, lin->N, lin->K, dst);
                frag_w_[src] = dst;
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
            for (int S : {4, 12}) {  // (the splits ffn_split_choose hands out)
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
namespace {
template <typename T>
T* up(Arena& ar, hipStream_t s, const T* h, size_t n) {
    T* d = static_cast<T*>(ar.alloc(n * sizeof(T)));
    STN_HIP(hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    return d;
}
}  // namespace

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

// =================================================================================================
// resident batch
// =================================================================================================
namespace {
}  // namespace

// Grow-only device buffer of the resident batch.  A reallocation retires the old block (freed at the next upload, once the
// stream has drained) and bumps the generation that is part of the graph key: a captured graph holds raw pointers.
template <typename T>
void Engine::ensure(T*& p, size_t& cap, size_t need) {
    if (p && need <= cap) return;
    const size_t n = std::max<size_t>(need + need / 4, 64);  // headroom: ragged request streams settle after a few uploads
    T* q = nullptr;
    STN_HIP(hipMalloc(reinterpret_cast<void**>(&q), n * sizeof(T)));
    if (p) {
        batch_owned_.erase(std::find(batch_owned_.begin(), batch_owned_.end(), static_cast<void*>(p)));
        batch_retired_.push_back(p);
    }
    batch_owned_.push_back(q);
    p = q;
    cap = n;
    ++bt_.gen;
}

void Engine::batch_upload(int B, int Lt, const int64_t* ids, const float* text_mask, const float* style_ttl,
                          const float* style_dp, const float* duration_override, const int64_t* utt_ids) {
    STN_HIP(hipSetDevice(device_));
    if (!loaded_) throw std::runtime_error("no model loaded");
    if (B <= 0 || Lt <= 0) throw std::runtime_error("empty batch");
    sync();
    for (void* p : batch_retired_) (void)hipFree(p);
    batch_retired_.clear();
    Batch& b = bt_;
    b.B = B; b.Lt = Lt; b.L = 0; b.noise_L = 0; b.total_step = 0;
    b.have_override = false; b.have_noise = false;
    b.h_dur.clear(); b.h_llen.clear();
    const size_t n_ttl = (size_t)B * a_.n_style_ttl * a_.d_style_ttl, n_dp = (size_t)B * a_.n_style_dp * a_.d_style_dp;
    ensure(b.ids, b.ids_cap, (size_t)B * Lt);
    ensure(b.tlen, b.tlen_cap, (size_t)B);
    ensure(b.style_ttl, b.ttl_cap, n_ttl);
    ensure(b.style_dp, b.dp_cap, n_dp);
    ensure(b.dur, b.dur_cap, (size_t)B);
    ensure(b.llen, b.llen_cap, (size_t)B);
    ensure(b.utt_ids, b.utt_cap, (size_t)B);
    STN_HIP(hipMemcpyAsync(b.ids, ids, sizeof(int64_t) * B * Lt, hipMemcpyHostToDevice, s_));
    STN_HIP(hipMemcpyAsync(b.style_ttl, style_ttl, sizeof(float) * n_ttl, hipMemcpyHostToDevice, s_));
    STN_HIP(hipMemcpyAsync(b.style_dp, style_dp, sizeof(float) * n_dp, hipMemcpyHostToDevice, s_));
    std::vector<int64_t> uid(B);
    for (int i = 0; i < B; ++i) uid[i] = utt_ids ? utt_ids[i] : i;
    STN_HIP(hipMemcpyAsync(b.utt_ids, uid.data(), sizeof(int64_t) * B, hipMemcpyHostToDevice, s_));
    ar_.reset();
    float* d_mask = up(ar_, s_, text_mask, (size_t)B * Lt);
    launch_mask_to_len(s_, d_mask, B, Lt, b.tlen);
    // packed text rows: first row of each utterance (fixed for the life of this upload) and their total
    ensure(b.toff, b.toff_cap, (size_t)B + 1);
    b.trows = 0;
    for (int i = 0; i < B; ++i) {
        int n = 0;
        for (int t = 0; t < Lt; ++t) n += text_mask[(size_t)i * Lt + t] > 0.5f ? 1 : 0;  // as mask_to_len_kernel counts
        b.trows += n;
    }
    if (B <= 1024) launch_row_map(s_, b.tlen, B, b.toff, nullptr);
    b.have_override = duration_override != nullptr;
    if (duration_override) b.h_dur.assign(duration_override, duration_override + B);
    sync();
}

void Engine::batch_set_noise(const float* noise, int L) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    if (b.B == 0) throw std::runtime_error("batch_set_noise: no batch uploaded");
    if (L < 1) throw std::runtime_error("batch_set_noise: L must be >= 1");
    const int D = a_.latent_dim * a_.chunk_compress_factor;
    sync();
    ensure(b.noise, b.noise_cap, (size_t)b.B * D * L);
    STN_HIP(hipMemcpyAsync(b.noise, noise, sizeof(float) * b.B * D * L, hipMemcpyHostToDevice, s_));
    b.have_noise = true;
    b.noise_L = L;  // checked against the durations in batch_run
    sync();
}

// Latent geometry exactly as TextToSpeech::sampleNoisyLatent (/root/reference/cpp/helper.cpp:424-440,457,764-768):
// float32 products, truncation to integers.
static void latent_geometry(const stn_arch& a, const std::vector<float>& dur, int& L, std::vector<int>& llen) {
    const int cs = a.base_chunk_size * a.chunk_compress_factor;
    float mx = dur[0];
    for (float d : dur) mx = std::max(mx, d);
    const float wav_len_max = mx * (float)a.sample_rate;
    L = (int)((wav_len_max + (float)cs - 1.0f) / (float)cs);
    llen.resize(dur.size());
    for (size_t i = 0; i < dur.size(); ++i) {
        const int64_t wl = (int64_t)(dur[i] * (float)a.sample_rate);
        llen[i] = (int)((wl + cs - 1) / cs);
    }
}

void Engine::batch_run(int total_step, float speed, uint64_t noise_seed) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    if (b.B == 0) throw std::runtime_error("batch_run: no batch uploaded");
    if (total_step < 1) throw std::runtime_error("total_step must be >= 1");
    if (!(speed > 0.f)) throw std::runtime_error("speed must be > 0");
    const stn_arch& a = a_;
    const int B = b.B, Lt = b.Lt, D = a.latent_dim * a.chunk_compress_factor;
    b.total_step = total_step; b.speed = speed; b.noise_seed = noise_seed;
    ar_.reset();
    // 1. duration predictor (always executed; its output may be overridden for shape control)
    Ragged trg;
    const bool tpk = packed_text_ok(B) && b.trows > 0;
    if (tpk) { trg.off = b.toff; trg.rows = b.trows; }
    std::vector<float> dur(B);
    // the text rows: persistent, grow-only buffers (a reallocation bumps b.gen, which every graph key holds)
    const size_t text_bytes = (size_t)(tpk ? (int64_t)b.trows : (int64_t)B * Lt) * a.te_out_dim * (is_half(dt_) ? 2 : 4);
    ensure(b.text_side, b.text_side_cap, text_bytes);
    ensure(b.text_rows, b.text_cap, text_bytes);
    {
        // on the side streams, each with its own workspace (see dp_s_): swapped in for the duration of a text stage
        struct Side {
            Engine& e; hipStream_t& s; Arena& ar; bool on;
            Side(Engine& e_, hipStream_t& s_, Arena& ar_) : e(e_), s(s_), ar(ar_), on(s_ != nullptr) { if (on) { std::swap(e.s_, s); e.ar_.swap(ar); e.ar_.reset(); } }
            ~Side() { if (on) { std::swap(e.s_, s); e.ar_.swap(ar); } }
        };
        {   // 2. text encoder -> context rows (act dtype), once the previous run has taken its copy of them
            Side side(*this, te_s_, te_ar_);
            if (side.on && copied_valid_) STN_HIP(hipStreamWaitEvent(s_, ev_copied_, 0));
            text_enc_dev(B, Lt, b.ids, b.style_ttl, b.tlen, nullptr, b.text_side, tpk ? &trg : nullptr);
            STN_HIP(hipEventRecord(ev_te_, s_));
        }
        {   // 1. duration predictor (always executed; its output may be overridden for shape control), beside the encoder
            Side side(*this, dp_s_, dp_ar_);
            duration_dev(B, Lt, b.ids, b.style_dp, b.tlen, b.dur, tpk ? &trg : nullptr);
            if (!b.have_override) {
                STN_HIP(hipMemcpyAsync(dur.data(), b.dur, sizeof(float) * B, hipMemcpyDeviceToHost, s_));
                STN_HIP(hipEventRecord(ev_dp_, s_));
            }
        }
        if (b.have_override) dur = b.h_dur;  // known on the host: no device->host read, no sync
        else STN_HIP(hipEventSynchronize(ev_dp_));  // the one host round trip (the predictor only): L = f(max duration) sizes every later buffer
    }
    // Hand-over of this run's text rows to the main pipeline (everything captured below reads b.text_rows).  It happens where the
    // rows are first needed — in front of the text K/V GEMM of the first text cross-attention — so the noise, the style K/V, the time
    // conditioning and the first ConvNeXt blocks of the estimator run beside the encoder; a captured pipeline is therefore TWO
    // graphs with this hand-over between them.
    auto take_text_rows = [this, &b, text_bytes]() {
        if (te_s_) STN_HIP(hipStreamWaitEvent(s_, ev_te_, 0));
        STN_HIP(hipMemcpyAsync(b.text_rows, b.text_side, text_bytes, hipMemcpyDeviceToDevice, s_));
        STN_HIP(hipEventRecord(ev_copied_, s_));
        copied_valid_ = true;
    };
    for (float& d : dur) d /= speed;  // cpp/helper.cpp:529-531
    int L = 0;
    latent_geometry(a, dur, L, b.h_llen);
    if (L < 1) throw std::runtime_error("predicted duration too short: zero latent frames");
    if (b.have_noise && b.noise_L != L)
        throw std::runtime_error("injected noise has L=" + std::to_string(b.noise_L) + " but the durations imply L=" + std::to_string(L));
    b.L = L;
    reported_dur_ = dur;  // durations after /speed: what the reference returns (cpp/helper.cpp:680)
    const size_t nx = (size_t)B * D * L, nw = (size_t)B * L * a.base_chunk_size * a.chunk_compress_factor;
    ensure(b.xt[0], b.xt_cap[0], nx);
    ensure(b.xt[1], b.xt_cap[1], nx);
    ensure(b.wav, b.wav_cap, nw);
    // per-call data of the captured region lives in pinned host memory (the graph's memcpy nodes re-read it at replay).
    // Sized for 1024 utterances up front so that it is not reallocated under cached graphs; should it ever have to grow, its
    // address is part of the graph key and the graphs that point at the old block are dropped before it is freed.
    if ((size_t)B > pin_llen_cap_) {
        if (pin_llen_) { sync(); drop_graphs(); (void)hipHostFree(pin_llen_); pin_llen_ = nullptr; }
        const size_t cap = std::max<size_t>((size_t)B, 1024);
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_llen_), sizeof(int) * cap, hipHostMallocDefault));
        pin_llen_cap_ = cap;
        pin_valid_ = false;
    }
    if (!pin_seed_) {
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_seed_), sizeof(unsigned long long), hipHostMallocDefault));
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&seed_dev_), sizeof(unsigned long long)));
        pin_valid_ = false;
    }
    // a previous run's copy nodes may still be reading the staging: rewrite it only when the content changes, and then
    // only after the stream has drained (steady-state replays of an unchanged batch never wait here)
    if (!pin_valid_ || !std::equal(b.h_llen.begin(), b.h_llen.end(), pin_llen_) || *pin_seed_ != (unsigned long long)noise_seed) {
        sync();
        std::copy(b.h_llen.begin(), b.h_llen.end(), pin_llen_);
        *pin_seed_ = (unsigned long long)noise_seed;
        pin_valid_ = true;
    }

    GraphKey key;
    key.B = B; key.Lt = Lt; key.L = L; key.steps = total_step; key.noise = b.have_noise; key.ragged = vo_ragged_; key.xattn = fused_xattn_;
    key.ffn = fused_ffn_; key.gen = b.gen; key.wgen = wgen_; key.pin = pin_llen_;
    key.rows = 0;
    if (packed_rows_ok(B)) for (int v : b.h_llen) key.rows += v;
    last_ve_rows_ = key.rows ? key.rows : (int64_t)B * L;
    key.vrows = trimmed_rows(B, L, nullptr);
    key.trows = tpk ? b.trows : 0;
    last_vo_rows_ = (int64_t)B * L * a.chunk_compress_factor;
    if (vo_ragged_ && packed_ve_ && is_half(dt_)) { last_vo_rows_ = 0; for (int v : b.h_llen) last_vo_rows_ += (int64_t)v * a.chunk_compress_factor; }
    else if (key.vrows) last_vo_rows_ = key.vrows;
    key.p0 = b.xt[0]; key.p1 = b.wav; key.s = s_;
    // event timing forces eager launches: hipEventRecord captured into a graph returns garbage spans on ROCm 7.2 (measured)
    const bool graphable = graph_on_ && !prof_on_;
    if (graphable) {
        for (auto& g : graphs_)
            if (g.key == key) {
                g.last_use = ++graph_clock_;
                STN_HIP(hipGraphLaunch(g.exec, s_));
                take_text_rows();
                STN_HIP(hipGraphLaunch(g.exec2, s_));
                ++graph_replays_;
                return;
            }
    }
    const auto warm = std::find(warm_keys_.begin(), warm_keys_.end(), key);
    if (graphable && warm != warm_keys_.end()) {  // second sighting of this shape: the arena is warm, allocation order is fixed
        const Arena::Mark cap0 = ar_.mark();
        const size_t cap_before = ar_.capacity();
        STN_HIP(hipStreamBeginCapture(s_, hipStreamCaptureModeThreadLocal));
        bool ok = true;
        std::string why;
        hipGraph_t g = nullptr, g2 = nullptr;
        hipError_t ec1 = hipErrorUnknown;
        bool split = false;
        // at the hand-over the first graph ends and the second begins (the hand-over itself is issued between their launches)
        auto split_capture = [&]() {
            ec1 = hipStreamEndCapture(s_, &g);
            split = true;
            STN_HIP(hipStreamBeginCapture(s_, hipStreamCaptureModeThreadLocal));
        };
        try { enqueue_after_duration(total_step, split_capture); } catch (const std::exception& e) { ok = false; why = e.what(); }
        const hipError_t ec = hipStreamEndCapture(s_, &g2);
        text_gate_ = nullptr;
        ar_.release(cap0);
        hipGraphExec_t ex = nullptr, ex2 = nullptr;
        if (ok && split && ec1 == hipSuccess && ec == hipSuccess && g && g2 && ar_.capacity() == cap_before &&
            hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && hipGraphInstantiate(&ex2, g2, nullptr, nullptr, 0) == hipSuccess) {
            if (graphs_.size() >= kGraphCache) {  // evict the least recently used entry
                auto lru = std::min_element(graphs_.begin(), graphs_.end(), [](const GraphEntry& x, const GraphEntry& y) { return x.last_use < y.last_use; });
                sync();  // its last replay may still be running
                (void)hipGraphExecDestroy(lru->exec);
                (void)hipGraphDestroy(lru->graph);
                if (lru->exec2) (void)hipGraphExecDestroy(lru->exec2);
                if (lru->graph2) (void)hipGraphDestroy(lru->graph2);
                graphs_.erase(lru);
            }
            GraphEntry e;
            e.key = key; e.graph = g; e.exec = ex; e.graph2 = g2; e.exec2 = ex2; e.last_use = ++graph_clock_;
            graphs_.push_back(e);
            warm_keys_.erase(warm);
            STN_HIP(hipGraphLaunch(ex, s_));
            take_text_rows();
            STN_HIP(hipGraphLaunch(ex2, s_));
            ++graph_replays_;
            return;
        }
        if (ex) (void)hipGraphExecDestroy(ex);
        if (ex2) (void)hipGraphExecDestroy(ex2);
        if (g2) (void)hipGraphDestroy(g2);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (!ok) throw std::runtime_error("graph capture failed: " + why);
        // fall through to an eager run
    }
    if (warm == warm_keys_.end()) {
        if (warm_keys_.size() >= kWarmKeys) warm_keys_.erase(warm_keys_.begin());
        warm_keys_.push_back(key);
    }
    enqueue_after_duration(total_step, take_text_rows);
}

// Everything after the duration read: lengths to the device, text encoder, initial latent, Euler loop, vocoder.
void Engine::enqueue_after_duration(int total_step, const std::function<void()>& take_text_rows) {
    Batch& b = bt_;
    const stn_arch& a = a_;
    const int B = b.B, Lt = b.Lt, L = b.L, D = a.latent_dim * a.chunk_compress_factor;
    const size_t nx = (size_t)B * D * L;
    STN_HIP(hipMemcpyAsync(b.llen, pin_llen_, sizeof(int) * B, hipMemcpyHostToDevice, s_));
    STN_HIP(hipMemcpyAsync(seed_dev_, pin_seed_, sizeof(unsigned long long), hipMemcpyHostToDevice, s_));
    // 2. (the text encoder ran on the side stream: batch_run) its rows
    Ragged trg;
    const bool tpk = packed_text_ok(B) && b.trows > 0;
    if (tpk) { trg.off = b.toff; trg.rows = b.trows; }
    const Ragged* trgp = tpk ? &trg : nullptr;
    void* text_rows = b.text_rows;
    // 3. initial latent
    if (b.have_noise) {
        STN_HIP(hipMemcpyAsync(b.xt[0], b.noise, nx * 4, hipMemcpyDeviceToDevice, s_));
        launch_mask_ncl(s_, b.xt[0], B, D, L, b.llen);
    } else {
        launch_randn_masked(s_, 0, b.utt_ids, B, D, L, b.llen, b.xt[0], seed_dev_);
    }
    // 4. Euler loop: step-invariant K/V once, the time conditioning of every step in one pass, then total_step passes
    VeCtx c = ve_prepare_dev(B, Lt, text_rows, b.style_ttl, b.tlen, trgp, /*defer_text=*/true);
    // armed here, fired by the first text cross-attention of the first Euler step (ve_step_dev): hand-over of the rows, then the text K/V
    text_gate_ = [this, &c, &b, &take_text_rows, B, Lt, text_rows, trgp]() {
        take_text_rows();
        ve_text_kv_dev(c, B, Lt, text_rows, b.tlen, trgp);
    };
    struct Disarm { std::function<void()>& g; ~Disarm() { g = nullptr; } } disarm{text_gate_};  // it refers to this frame: never outlives it
    if (a.ve_main_blocks == 0) { auto fire = std::move(text_gate_); text_gate_ = nullptr; fire(); }  // (no cross-attention would ever fire it)
    float* tot_all = f32_alloc((int64_t)total_step * B);
    float* cur_all = f32_alloc((int64_t)total_step * B);
    float* dt_all = f32_alloc(B);
    launch_step_counters(s_, tot_all, cur_all, dt_all, B, total_step);
    const float* tb_all = ve_time_cond_dev(total_step * B, tot_all, cur_all);
    const size_t tb_stride = (size_t)B * a.ve_main_blocks * a.ve_dim;
    Ragged rg;
    const Ragged* rgp = nullptr;
    if (packed_rows_ok(B)) {  // the estimator works on the frames the utterances own and nothing else
        rg.rows = 0;
        for (int v : b.h_llen) rg.rows += v;
        int* off = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)(B + 1)));
        int* row_b = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)std::max(rg.rows, 1)));
        launch_row_map(s_, b.llen, B, off, row_b);
        rg.off = off; rg.row_b = row_b;
        rgp = &rg;
    }
    int cur = 0;
    for (int st = 0; st < total_step; ++st) {
        ve_step_dev(B, L, c, b.xt[cur], b.tlen, b.llen, tot_all + (size_t)st * B, cur_all + (size_t)st * B, b.xt[cur ^ 1],
                    tb_all + (size_t)st * tb_stride, rgp, dt_all);
        cur ^= 1;
    }
    final_xt_ = cur;
    // 5. vocoder
    const int* vlen = nullptr;
    if (vo_ragged_) {
        int* v = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        launch_scale_len(s_, b.llen, B, a.chunk_compress_factor, v);
        vlen = v;
    }
    int vrows = 0;
    const int* valid = nullptr;
    if (vo_ragged_ && packed_ve_) {
        for (int v : b.h_llen) vrows += v * a.chunk_compress_factor;
    } else if (const int tr = trimmed_rows(B, L, nullptr)) {
        // reference (dense) semantics at the cost of the frames that are not position-independent
        int* n_dev = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        int* v_dev = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        launch_trim_len(s_, b.llen, B, a.chunk_compress_factor, L * a.chunk_compress_factor, vo_rf_, n_dev, v_dev);
        vlen = n_dev; valid = v_dev; vrows = tr;
    }
    vocoder_dev(B, L, b.xt[cur], b.wav, vlen, vrows, valid);
    STN_HIP(hipGetLastError());  // a kernel launch that was rejected (bad configuration) must not pass silently
}

void Engine::batch_fetch(float* wav, size_t wav_capacity, float* duration) {
    Batch& b = bt_;
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (wav) {
        if (wav_capacity < nw) throw std::runtime_error("wav buffer too small: need " + std::to_string(nw) + " floats");
        STN_HIP(hipMemcpyAsync(wav, b.wav, nw * 4, hipMemcpyDeviceToHost, s_));
    }
    sync();
    if (duration) std::copy(reported_dur_.begin(), reported_dur_.end(), duration);
}
void Engine::batch_fetch_pcm16(int16_t* pcm, size_t capacity, float* duration) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if (capacity < nw) throw std::runtime_error("pcm buffer too small: need " + std::to_string(nw) + " samples");
    ensure(b.pcm, b.pcm_cap, nw);
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)(nw / (size_t)b.B), b.pcm, (int64_t)(nw / (size_t)b.B));
    STN_HIP(hipMemcpyAsync(pcm, b.pcm, nw * 2, hipMemcpyDeviceToHost, s_));
    sync();
    if (duration) std::copy(reported_dur_.begin(), reported_dur_.end(), duration);
}
void Engine::batch_fetch_pcm16_begin(int slot) {
    STN_HIP(hipSetDevice(device_));
    if (slot < 0 || slot > 1) throw std::invalid_argument("fetch slot must be 0 or 1");
    Batch& b = bt_;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    FetchSlot& f = fetch_[slot];
    if (!copy_s_) STN_HIP(hipStreamCreateWithFlags(&copy_s_, hipStreamNonBlocking));
    if (!f.ready) { STN_HIP(hipEventCreateWithFlags(&f.ready, hipEventDisableTiming)); STN_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming)); }
    if (f.busy) STN_HIP(hipEventSynchronize(f.done));  // the slot's previous copy (two batches ago) must be out before it is refilled
    if (nw > f.cap) {
        if (f.dev) (void)hipFree(f.dev);
        if (f.pin) (void)hipHostFree(f.pin);
        f.dev = nullptr; f.pin = nullptr;
        const size_t cap = nw + nw / 4;
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&f.dev), cap * sizeof(int16_t)));
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&f.pin), cap * sizeof(int16_t), hipHostMallocDefault));
        f.cap = cap;
    }
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)(nw / (size_t)b.B), f.dev, (int64_t)(nw / (size_t)b.B));
    STN_HIP(hipEventRecord(f.ready, s_));
    STN_HIP(hipStreamWaitEvent(copy_s_, f.ready, 0));
    STN_HIP(hipMemcpyAsync(f.pin, f.dev, nw * sizeof(int16_t), hipMemcpyDeviceToHost, copy_s_));
    STN_HIP(hipEventRecord(f.done, copy_s_));
    f.n = nw;
    f.dur = reported_dur_;
    f.busy = true;
}
void Engine::batch_fetch_pcm16_end(int slot, const int16_t** pcm, size_t* n, float* duration) {
    if (slot < 0 || slot > 1) throw std::invalid_argument("fetch slot must be 0 or 1");
    FetchSlot& f = fetch_[slot];
    if (!f.busy) throw std::runtime_error("no fetch in flight on this slot");
    STN_HIP(hipEventSynchronize(f.done));
    if (pcm) *pcm = f.pin;
    if (n) *n = f.n;
    if (duration) std::copy(f.dur.begin(), f.dur.end(), duration);
}
void Engine::batch_copy_wav_device(float* dst, int64_t dst_stride) {
    Batch& b = bt_;
    const size_t W = (size_t)b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if ((size_t)dst_stride < W) throw std::invalid_argument("dst_stride smaller than the waveform length");
    STN_HIP(hipMemcpy2DAsync(dst, (size_t)dst_stride * 4, b.wav, W * 4, W * 4, (size_t)b.B, hipMemcpyDeviceToDevice, s_));
}
void Engine::batch_copy_pcm16_device(int16_t* dst, int64_t dst_stride) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    const size_t W = (size_t)b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if ((size_t)dst_stride < W) throw std::invalid_argument("dst_stride smaller than the waveform length");
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)W, dst, dst_stride);
}
void Engine::batch_fetch_latent(float* latent) {
    Batch& b = bt_;
    const size_t nx = (size_t)b.B * a_.latent_dim * a_.chunk_compress_factor * b.L;
    STN_HIP(hipMemcpyAsync(latent, b.xt[final_xt_], nx * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

// =================================================================================================
// op-level test entry points
// =================================================================================================
void Engine::op_gemm(int dtype, int M, int N, int K, const float* A, const float* W, const float* bias, int act, float* out) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    float* dA = up(ar_, s_, A, (size_t)M * K);
    float* dW = up(ar_, s_, W, (size_t)N * K);
    float* dB = bias ? up(ar_, s_, bias, (size_t)N) : nullptr;
    float* dO = f32_alloc((size_t)M * N);
    const void* pa = dA;
    const void* pw = dW;
    if (is_half(dtype)) {
        void* a16 = ar_.alloc((size_t)M * K * 2);
        void* w16 = ar_.alloc((size_t)N * K * 2);
        launch_cast(s_, dtype, dA, (int64_t)M * K, a16);
        launch_cast(s_, dtype, dW, (int64_t)N * K, w16);
        pa = a16; pw = w16;
    }
    Epilogue e; e.mode = EPI_STORE; e.act = act; e.out_dtype = F32; e.out = dO; e.ldo = N; e.bias = dB;
    const int sk = gemm_splitk_factor(dtype, M, N, K, e);  // the same decision the model path takes (Engine::gemm)
    if (sk > 1) launch_gemm_splitk(s_, dtype, pa, K, pw, K, M, N, K, e, sk, f32_alloc((int64_t)sk * M * N));
    else launch_gemm(s_, dtype, pa, K, pw, K, M, N, K, e);
    STN_HIP(hipMemcpyAsync(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_dwconv_ln(int dtype, int B, int L, int C, int k, int dil, const float* x, const float* w, const float* bias,
                          const float* g, const float* b, float* y, const int* seqlen) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t n = (size_t)B * L * C;
    const int* dlen = seqlen ? up(ar_, s_, seqlen, (size_t)B) : nullptr;
    std::vector<float> wt((size_t)C * k);
    for (int c = 0; c < C; ++c) for (int j = 0; j < k; ++j) wt[(size_t)j * C + c] = w[(size_t)c * k + j];
    float* dx = up(ar_, s_, x, n);
    float* dw = up(ar_, s_, wt.data(), wt.size());
    float* db = up(ar_, s_, bias, (size_t)C);
    float* dg = up(ar_, s_, g, (size_t)C);
    float* dbt = up(ar_, s_, b, (size_t)C);
    void* dy = ar_.alloc(n * 4);
    float* dy32 = f32_alloc(n);
    launch_dwconv_ln(s_, dtype, dx, B, L, C, dw, db, k, dil, dg, dbt, 1e-6f, dy, dlen);
    if (is_half(dtype)) launch_half_to_f32(s_, dtype, dy, (int64_t)n, dy32);
    STN_HIP(hipMemcpyAsync(y, is_half(dtype) ? dy32 : static_cast<float*>(dy), n * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_attention(int dtype, int B, int Lq, int Lk, int H, int dh, const float* q, const float* k, const float* v,
                          const int* qlen, const int* klen, int rope_mode, float* o) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const int C = H * dh;
    const size_t nq = (size_t)B * Lq * C, nk = (size_t)B * Lk * C;
    float* dq = up(ar_, s_, q, nq);
    float* dk = up(ar_, s_, k, nk);
    float* dv = up(ar_, s_, v, nk);
    int* dql = qlen ? up(ar_, s_, qlen, (size_t)B) : nullptr;
    int* dkl = klen ? up(ar_, s_, klen, (size_t)B) : nullptr;
    const void *pq = dq, *pk = dk, *pv = dv;
    if (is_half(dtype)) {
        void* a = ar_.alloc(nq * 2); void* b = ar_.alloc(nk * 2); void* c = ar_.alloc(nk * 2);
        launch_cast(s_, dtype, dq, (int64_t)nq, a); launch_cast(s_, dtype, dk, (int64_t)nk, b); launch_cast(s_, dtype, dv, (int64_t)nk, c);
        pq = a; pk = b; pv = c;
    }
    void* dO = ar_.alloc(nq * 4);
    float* dO32 = f32_alloc(nq);
    const float rbase = a_.rope_base > 0 ? a_.rope_base : 10000.f, rgam = a_.larope_gamma > 0 ? a_.larope_gamma : 10.f;
    const bool prerot = rope_mode >= 0 && (rope_mode & 0x100) != 0;  // test hook: rotate the keys in a separate pass first
    if (prerot) rope_mode &= 0xFF;
    if (prerot) launch_rope_rows(s_, dtype, const_cast<void*>(pk), C, B, Lk, dkl, 1, 0, H, dh, rope_mode, rbase, rgam);
    launch_attention(s_, dtype, pq, C, pk, pv, C, dO, C, B, Lq, Lk, H, dh, dql, dkl, rope_mode, rbase, rgam, prerot);
    if (is_half(dtype)) launch_half_to_f32(s_, dtype, dO, (int64_t)nq, dO32);
    STN_HIP(hipMemcpyAsync(o, is_half(dtype) ? dO32 : static_cast<float*>(dO), nq * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

double Engine::op_gemm_bench(int dtype, int M, int N, int K, int mode, int iters) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t esz = is_half(dtype) ? 2 : 4;
    float* tmp = f32_alloc((size_t)std::max((size_t)M * K, (size_t)N * K));
    void* A = ar_.alloc((size_t)M * K * esz);
    void* Wt = ar_.alloc((size_t)N * K * esz);
    launch_randn_masked(s_, 11, nullptr, 1, 1, (int)std::min<size_t>((size_t)M * K, 1u << 30), nullptr, tmp);
    launch_cast(s_, dtype, tmp, (int64_t)M * K, A);
    launch_randn_masked(s_, 12, nullptr, 1, 1, (int)std::min<size_t>((size_t)N * K, 1u << 30), nullptr, tmp);
    launch_scale(s_, tmp, (int)std::min<size_t>((size_t)N * K, 1u << 30), 1.0f / std::sqrt((float)K));
    launch_cast(s_, dtype, tmp, (int64_t)N * K, Wt);
    float* bias = f32_alloc(N);
    float* gamma = f32_alloc(N);
    launch_fill(s_, bias, N, 0.01f);
    launch_fill(s_, gamma, N, 0.2f);
    float* resid = f32_alloc((size_t)M * N);
    void* out = ar_.alloc((size_t)M * N * 4);
    STN_HIP(hipMemsetAsync(resid, 0, (size_t)M * N * 4, s_));
    Epilogue e;
    e.bias = bias;
    if (mode == 1) { e.mode = EPI_RESID; e.resid = resid; e.ldo = N; e.gamma = gamma; }
    else { e.mode = EPI_STORE; e.act = mode == 2 ? ACT_NONE : ACT_GELU; e.out_dtype = mode == 3 ? F32 : dtype; e.out = out; e.ldo = N; }
    for (int i = 0; i < 3; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a));
    STN_HIP(hipEventCreate(&b));
    STN_HIP(hipEventRecord(a, s_));
    for (int i = 0; i < iters; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    STN_HIP(hipEventRecord(b, s_));
    STN_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    STN_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (double)ms / iters;
}

void Engine::op_gemm_phases(int dtype, int M, int N, int K, int mode, double* out6) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t esz = is_half(dtype) ? 2 : 4;
    float* tmp = f32_alloc((size_t)std::max((size_t)M * K, (size_t)N * K));
    void* A = ar_.alloc((size_t)M * K * esz);
    void* Wt = ar_.alloc((size_t)N * K * esz);
    launch_randn_masked(s_, 11, nullptr, 1, 1, (int)std::min<size_t>((size_t)M * K, 1u << 30), nullptr, tmp);
    launch_cast(s_, dtype, tmp, (int64_t)M * K, A);
    launch_randn_masked(s_, 12, nullptr, 1, 1, (int)std::min<size_t>((size_t)N * K, 1u << 30), nullptr, tmp);
    launch_scale(s_, tmp, (int)std::min<size_t>((size_t)N * K, 1u << 30), 1.0f / std::sqrt((float)K));
    launch_cast(s_, dtype, tmp, (int64_t)N * K, Wt);
    float* bias = f32_alloc(N);
    float* gamma = f32_alloc(N);
    launch_fill(s_, bias, N, 0.01f);
    launch_fill(s_, gamma, N, 0.2f);
    float* resid = f32_alloc((size_t)M * N);
    void* out = ar_.alloc((size_t)M * N * 4);
    STN_HIP(hipMemsetAsync(resid, 0, (size_t)M * N * 4, s_));
    const size_t max_wg = 1 << 16;
    unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(max_wg * 4 * 8));
    Epilogue e;
    e.bias = bias;
    if (mode == 1) { e.mode = EPI_RESID; e.resid = resid; e.ldo = N; e.gamma = gamma; }
    else { e.mode = EPI_STORE; e.act = mode == 2 ? ACT_NONE : ACT_GELU; e.out_dtype = mode == 3 ? F32 : dtype; e.out = out; e.ldo = N; }
    for (int i = 0; i < 3; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    STN_HIP(hipMemsetAsync(ts, 0, max_wg * 4 * 8, s_));
    e.ts = ts;
    launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    std::vector<unsigned long long> h(max_wg * 4);
    STN_HIP(hipMemcpyAsync(h.data(), ts, max_wg * 4 * 8, hipMemcpyDeviceToHost, s_));
    sync();
    double p0 = 0, p1 = 0, p2 = 0;
    unsigned long long tmin = ~0ull, tmax_in = 0, tend = 0;
    size_t n = 0;
    for (size_t w = 0; w < max_wg; ++w) {
        const unsigned long long* t = &h[w * 4];
        if (t[3] == 0) continue;
        ++n;
        p0 += (double)(t[1] - t[0]); p1 += (double)(t[2] - t[1]); p2 += (double)(t[3] - t[2]);
        tmin = std::min(tmin, t[0]); tmax_in = std::max(tmax_in, t[0]); tend = std::max(tend, t[3]);
    }
    if (n == 0) throw std::runtime_error("op_gemm_phases: this shape does not run on the tiled kernel");
    out6[0] = p0 / n; out6[1] = p1 / n; out6[2] = p2 / n;
    out6[3] = (double)(tend - tmin); out6[4] = (double)(tmax_in - tmin); out6[5] = (double)n;
}

void Engine::op_ffn(int M, int C, int I, const float* xn, const float* W1, const float* b1, const float* W2, const float* b2, const float* gamma,
                    const float* rowvec, const int* row_b, int nseq, float* x, int mode) {
    STN_HIP(hipSetDevice(device_));
    const bool fused = mode != 0;
    if (!is_half(dt_)) throw std::invalid_argument("op_ffn: 16-bit engines only");
    if (fused && !ffn_fused_supported(dt_, C, I)) throw std::invalid_argument("op_ffn: shape not supported by the fused kernel");
    if (mode == 2 && ffn_split_factor(dt_, C, I) < 2) throw std::invalid_argument("op_ffn: shape not supported by the hidden-split kernel");
    ar_.reset();
    float* d_xn = up(ar_, s_, xn, (size_t)M * C);
    float* d_w1 = up(ar_, s_, W1, (size_t)I * C);
    float* d_w2 = up(ar_, s_, W2, (size_t)C * I);
    float* d_b1 = up(ar_, s_, b1, (size_t)I);
    float* d_b2 = b2 ? up(ar_, s_, b2, (size_t)C) : nullptr;
    float* d_g = gamma ? up(ar_, s_, gamma, (size_t)C) : nullptr;
    float* d_x = up(ar_, s_, x, (size_t)M * C);
    float* d_rv = rowvec ? up(ar_, s_, rowvec, (size_t)nseq * C) : nullptr;
    int* d_rb = (rowvec && row_b) ? up(ar_, s_, row_b, (size_t)M) : nullptr;
    void* xn16 = act_alloc((int64_t)M * C);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, d_xn, (int64_t)M * C, xn16);
    launch_cast(s_, dt_, d_w1, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, d_w2, (int64_t)I * C, w2_16);
    if (fused) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        void* wseq = act_alloc((int64_t)2 * I * C);
        const int S = mode == 2 ? ffn_split_choose(dt_, C, I, M) : 1;
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
        FfnArgs fa;
        fa.xn = xn16; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.b2 = d_b2; fa.gamma = d_g; fa.x = d_x; fa.ldo = C;
        fa.M = M; fa.I = I; fa.rowvec = d_rv; fa.rv_ld = C; fa.row_b = d_rb; fa.L = M;
        if (mode == 2) {
            fa.split = S; fa.part_stride = ffn_split_rows(M) * C; fa.part = act_alloc(fa.part_stride * S);
            launch_ffn_fused(s_, dt_, C, fa);
            // the pending update, folded by the LayerNorm form of the fold (its normalised output is not part of this op)
            float* ones = f32_alloc(C);
            launch_fill(s_, ones, C, 1.f);
            float* zeros = f32_alloc(C);
            launch_fill(s_, zeros, C, 0.f);
            FoldArgs fo; fo.part = fa.part; fo.S = S; fo.part_stride = fa.part_stride; fo.b2 = d_b2 ? d_b2 : zeros; fo.gamma = d_g ? d_g : ones;
            fo.rowvec = d_rv; fo.rv_ld = C; fo.row_b = d_rb;
            void* y = act_alloc((int64_t)M * C);
            launch_fold_ln(s_, dt_, d_x, M, C, fo, ones, ones, a_.ln_eps, y);
        } else {
            launch_ffn_fused(s_, dt_, C, fa);
        }
    } else {
        void* u = act_alloc((int64_t)M * I);
        Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
        launch_gemm(s_, dt_, xn16, C, w1_16, C, M, I, C, e1);
        Epilogue e2; e2.mode = EPI_RESID; e2.resid = d_x; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.rowvec = d_rv; e2.rv_ld = C; e2.row_b = d_rb; e2.L = M;
        launch_gemm(s_, dt_, u, I, w2_16, I, M, C, I, e2);
    }
    STN_HIP(hipGetLastError());
    STN_HIP(hipMemcpyAsync(x, d_x, sizeof(float) * (size_t)M * C, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_ffn_bench(int M, int C, int I, int mode, int iters, double* out5) {
    STN_HIP(hipSetDevice(device_));
    const bool fused = mode != 0;
    const int S = mode == 2 ? ffn_split_choose(dt_, C, I, M) : 1;
    if (mode == 2 && S < 2) throw std::invalid_argument("op_ffn_bench: shape not supported by the hidden-split kernel");
    if (!is_half(dt_)) throw std::invalid_argument("op_ffn_bench: 16-bit engines only");
    if (fused && !ffn_fused_supported(dt_, C, I)) throw std::invalid_argument("op_ffn_bench: shape not supported by the fused kernel");
    ar_.reset();
    for (int i = 0; i < 5; ++i) out5[i] = 0.0;
    // random operands (the clock a chip holds on zeros is not the clock it holds on data)
    float* rnd = f32_alloc((int64_t)M * C);
    launch_randn_masked(s_, 11, nullptr, 1, M, C, nullptr, rnd);
    float* wr = f32_alloc((int64_t)I * C);
    launch_randn_masked(s_, 12, nullptr, 1, I, C, nullptr, wr);
    launch_scale(s_, wr, I * C, 0.05f);
    void* xn16 = act_alloc((int64_t)M * C);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, rnd, (int64_t)M * C, xn16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w2_16);
    float* d_b1 = f32_alloc(I);
    float* d_b2 = f32_alloc(C);
    float* d_g = f32_alloc(C);
    launch_fill(s_, d_b1, I, 0.01f); launch_fill(s_, d_b2, C, 0.01f); launch_fill(s_, d_g, C, 0.1f);
    float* d_x = f32_alloc((int64_t)M * C);
    STN_HIP(hipMemsetAsync(d_x, 0, sizeof(float) * (size_t)M * C, s_));
    void* wseq = nullptr;
    if (fused) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        wseq = act_alloc((int64_t)2 * I * C);
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
    }
    void* u = fused ? nullptr : act_alloc((int64_t)M * I);
    const int64_t pstride = ffn_split_rows(M) * C;
    void* part = mode == 2 ? act_alloc(pstride * S) : nullptr;
    const int nslab = (M + 127) / 128;
    const int nwg = mode == 2 ? (nslab + 7) / 8 * 8 * S : nslab;
    unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(sizeof(unsigned long long) * 4 * (size_t)nwg));
    auto run = [&](unsigned long long* stamps) {
        if (fused) {
            FfnArgs fa;
            fa.xn = xn16; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.b2 = d_b2; fa.gamma = d_g; fa.x = d_x; fa.ldo = C;
            fa.M = M; fa.I = I; fa.L = M; fa.ts = stamps;
            if (mode == 2) { fa.split = S; fa.part = part; fa.part_stride = pstride; }
            launch_ffn_fused(s_, dt_, C, fa);
        } else {
            Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
            if (nt_hints_ && (double)M * I * 2.0 > 128e6) e1.nt = 1;
            launch_gemm(s_, dt_, xn16, C, w1_16, C, M, I, C, e1);
            Epilogue e2; e2.mode = EPI_RESID; e2.resid = d_x; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.L = M;
            launch_gemm(s_, dt_, u, I, w2_16, I, M, C, I, e2);
        }
    };
    for (int i = 0; i < 3; ++i) run(nullptr);
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a)); STN_HIP(hipEventCreate(&b));
    STN_HIP(hipEventRecord(a, s_));
    for (int i = 0; i < iters; ++i) run(nullptr);
    STN_HIP(hipEventRecord(b, s_));
    STN_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    STN_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    out5[0] = ms / iters;
    if (fused) {
        STN_HIP(hipMemsetAsync(ts, 0, sizeof(unsigned long long) * 4 * (size_t)nwg, s_));
        run(ts);
        std::vector<unsigned long long> h((size_t)4 * nwg);
        STN_HIP(hipMemcpyAsync(h.data(), ts, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, s_));
        sync();
        double s1 = 0, s2 = 0, s3 = 0;
        int live = 0;  // (workgroups of a hidden-split grid beyond the last slab exit at once and leave no stamps)
        for (int w = 0; w < nwg; ++w) {
            if (!h[4 * w + 3]) continue;
            ++live;
            s1 += (double)(h[4 * w + 1] - h[4 * w]); s2 += (double)(h[4 * w + 2] - h[4 * w + 1]); s3 += (double)(h[4 * w + 3] - h[4 * w + 2]);
        }
        if (live) { out5[1] = s1 / live; out5[2] = s2 / live; out5[3] = s3 / live; }
        out5[4] = live;
    }
    STN_HIP(hipGetLastError());
    sync();
}

void Engine::op_fold_dwconv_ln(int B, int C, int k, int dil, int S, const int* seqlen, const float* x, const float* part, const float* b2,
                               const float* gamma, const float* rowvec, const float* w, const float* bias, const float* g, const float* b,
                               float* x_out, float* y) {
    STN_HIP(hipSetDevice(device_));
    if (!is_half(dt_)) throw std::invalid_argument("op_fold_dwconv_ln: 16-bit engines only");
    ar_.reset();
    std::vector<int> off(B + 1, 0);
    int L = 0;
    for (int i = 0; i < B; ++i) { off[i + 1] = off[i] + seqlen[i]; L = std::max(L, seqlen[i]); }
    const int64_t M = off[B];
    const size_t n = (size_t)M * C;
    std::vector<float> wt((size_t)C * k);
    for (int c = 0; c < C; ++c) for (int j = 0; j < k; ++j) wt[(size_t)j * C + c] = w[(size_t)c * k + j];
    const int* dlen = up(ar_, s_, seqlen, (size_t)B);
    const int* doff = up(ar_, s_, off.data(), (size_t)B + 1);
    float* dx = up(ar_, s_, x, n);
    float* dp32 = up(ar_, s_, part, n * S);
    float* db2 = b2 ? up(ar_, s_, b2, (size_t)C) : nullptr;
    float* dgm = gamma ? up(ar_, s_, gamma, (size_t)C) : nullptr;
    float* drv = rowvec ? up(ar_, s_, rowvec, (size_t)B * C) : nullptr;
    float* dw = up(ar_, s_, wt.data(), wt.size());
    float* db = up(ar_, s_, bias, (size_t)C);
    float* dg = up(ar_, s_, g, (size_t)C);
    float* dbt = up(ar_, s_, b, (size_t)C);
    void* dp16 = act_alloc((int64_t)n * S);
    launch_cast(s_, dt_, dp32, (int64_t)n * S, dp16);
    float* dxo = f32_alloc((int64_t)n);
    void* dy = act_alloc((int64_t)n);
    float* dy32 = f32_alloc((int64_t)n);
    float* ones = f32_alloc(C);
    launch_fill(s_, ones, C, 1.f);
    float* zeros = f32_alloc(C);
    launch_fill(s_, zeros, C, 0.f);
    FoldArgs fo; fo.part = dp16; fo.S = S; fo.part_stride = (int64_t)n; fo.b2 = db2 ? db2 : zeros; fo.gamma = dgm ? dgm : ones; fo.rowvec = drv; fo.rv_ld = C;
    launch_fold_dwconv_ln(s_, dt_, dx, dxo, B, L, C, fo, dw, db, k, dil, dg, dbt, 1e-6f, dy, dlen, doff);
    launch_half_to_f32(s_, dt_, dy, (int64_t)n, dy32);
    STN_HIP(hipGetLastError());
    STN_HIP(hipMemcpyAsync(x_out, dxo, n * 4, hipMemcpyDeviceToHost, s_));
    STN_HIP(hipMemcpyAsync(y, dy32, n * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_block_bench(int B, int L, int C, int I, int k, int dil, int mode, int iters, double* out2) {
    STN_HIP(hipSetDevice(device_));
    if (!is_half(dt_)) throw std::invalid_argument("op_block_bench: 16-bit engines only");
    const int S = ffn_split_choose(dt_, C, I, (int64_t)B * L);
    if (mode == 2 && S < 2) throw std::invalid_argument("op_block_bench: shape not supported by the hidden-split kernel");
    ar_.reset();
    for (int i = 0; i < 6; ++i) out2[i] = 0.0;
    const int64_t M = (int64_t)B * L;
    std::vector<int> len(B, L), off(B + 1);
    for (int i = 0; i <= B; ++i) off[i] = i * L;
    const int* dlen = up(ar_, s_, len.data(), (size_t)B);
    const int* doff = up(ar_, s_, off.data(), (size_t)B + 1);
    float* xa = f32_alloc(M * C);
    float* xb = f32_alloc(M * C);
    launch_randn_masked(s_, 11, nullptr, 1, (int)M, C, nullptr, xa);
    float* wr = f32_alloc((int64_t)I * C);
    launch_randn_masked(s_, 12, nullptr, 1, I, C, nullptr, wr);
    launch_scale(s_, wr, I * C, 0.05f);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w2_16);
    float* d_b1 = f32_alloc(I);
    float* d_b2 = f32_alloc(C);
    float* d_g = f32_alloc(C);
    float* d_one = f32_alloc(C);
    float* dwt = f32_alloc((int64_t)k * C);
    launch_fill(s_, d_b1, I, 0.01f); launch_fill(s_, d_b2, C, 0.01f); launch_fill(s_, d_g, C, 0.01f); launch_fill(s_, d_one, C, 1.f);
    launch_fill(s_, dwt, k * C, 1.f / k);
    void* xn = act_alloc(M * C);
    void* u = mode == 2 ? nullptr : act_alloc(M * I);
    void* wseq = nullptr;
    const int64_t pstride = ffn_split_rows(M) * C;
    void* part = nullptr;
    if (mode == 2) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        wseq = act_alloc((int64_t)2 * I * C);
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
        part = act_alloc(pstride * S);
        STN_HIP(hipMemsetAsync(part, 0, (size_t)pstride * S * 2, s_));
    }
    FoldArgs fo; fo.part = part; fo.S = S; fo.part_stride = pstride; fo.b2 = d_b2; fo.gamma = d_g;
    auto conv = [&]() {
        if (mode == 2) { launch_fold_dwconv_ln(s_, dt_, xa, xb, B, L, C, fo, dwt, d_b2, k, dil, d_one, d_b2, 1e-6f, xn, dlen, doff); std::swap(xa, xb); }
        else launch_dwconv_ln(s_, dt_, xa, B, L, C, dwt, d_b2, k, dil, d_one, d_b2, 1e-6f, xn, dlen, doff);
    };
    auto block = [&]() {
        conv();
        if (mode == 2) {
            FfnArgs fa; fa.xn = xn; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.M = (int)M; fa.I = I; fa.split = S; fa.part = part; fa.part_stride = pstride;
            launch_ffn_fused(s_, dt_, C, fa);
        } else {
            Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
            launch_gemm(s_, dt_, xn, C, w1_16, C, (int)M, I, C, e1);
            Epilogue e2; e2.mode = EPI_RESID; e2.resid = xa; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.L = (int)M;
            launch_gemm(s_, dt_, u, I, w2_16, I, (int)M, C, I, e2);
        }
    };
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a)); STN_HIP(hipEventCreate(&b));
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 3; ++i) { if (which) conv(); else block(); }
        STN_HIP(hipEventRecord(a, s_));
        for (int i = 0; i < iters; ++i) { if (which) conv(); else block(); }
        STN_HIP(hipEventRecord(b, s_));
        STN_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        STN_HIP(hipEventElapsedTime(&ms, a, b));
        out2[which] = ms / iters;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (mode == 2) {  // phase stamps of one fold_dwconv_ln launch
        const int nwg = B * ((L + 7) / 8);  // (runs of 8 frames when there are few sequences, of 32 otherwise: sized for the shorter)
        unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(sizeof(unsigned long long) * 4 * (size_t)nwg));
        STN_HIP(hipMemsetAsync(ts, 0, sizeof(unsigned long long) * 4 * (size_t)nwg, s_));
        fo.ts = ts;
        conv();
        fo.ts = nullptr;
        std::vector<unsigned long long> h((size_t)4 * nwg);
        STN_HIP(hipMemcpyAsync(h.data(), ts, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, s_));
        sync();
        double p1 = 0, p2 = 0, p3 = 0; int live = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < nwg; ++w) {
            if (!h[4 * w + 3]) continue;
            ++live;
            p1 += (double)(h[4 * w + 1] - h[4 * w]); p2 += (double)(h[4 * w + 2] - h[4 * w + 1]); p3 += (double)(h[4 * w + 3] - h[4 * w + 2]);
            tmin = std::min(tmin, h[4 * w]); tmax = std::max(tmax, h[4 * w + 3]);
        }
        if (live) { out2[2] = p1 / live; out2[3] = p2 / live; out2[4] = p3 / live; out2[5] = (double)(tmax - tmin); }
    }
    STN_HIP(hipGetLastError());
    sync();
}

void Engine::op_randn(uint64_t seed, int B, int D, int L, const int64_t* utt_ids, const int* len, float* out) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    int64_t* du = utt_ids ? up(ar_, s_, utt_ids, (size_t)B) : nullptr;
    int* dl = len ? up(ar_, s_, len, (size_t)B) : nullptr;
    float* d = f32_alloc((size_t)B * D * L);
    launch_randn_masked(s_, seed, du, B, D, L, dl, d);
    STN_HIP(hipMemcpyAsync(out, d, (size_t)B * D * L * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

}  // namespace stn
