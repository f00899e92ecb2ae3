// api.cpp — the C ABI of include/stn.h over stn::Engine.  No exception leaves this file.
#include "../../include/stn.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "engine.hpp"
#include "host/json_min.hpp"
#include "host/graph_bind.hpp"
#include "host/onnx_reader.hpp"

#include <map>
#include <sstream>

struct stn_handle {
    stn::Engine* eng = nullptr;
    std::string err;
    std::vector<std::pair<std::string, stn::KernelStat>> prof;
};

static thread_local std::string g_create_err;

#define STN_TRY(h, body)                                             \
    if (!(h)) return STN_ERR_INVALID;                                \
    try {                                                            \
        body;                                                        \
        return STN_OK;                                               \
    } catch (const std::invalid_argument& e) {                       \
        (h)->err = e.what();                                         \
        return STN_ERR_INVALID;                                      \
    } catch (const std::exception& e) {                              \
        (h)->err = e.what();                                         \
        return (h)->err.rfind("HIP error", 0) == 0 ? STN_ERR_DEVICE : STN_ERR_STATE; \
    } catch (...) {                                                  \
        (h)->err = "unknown failure";                                \
        return STN_ERR_DEVICE;                                       \
    }

static void need(bool ok, const char* what) {
    if (!ok) throw std::invalid_argument(what);
}
static void need_model(stn_handle* h) {
    if (!h->eng->loaded()) throw std::runtime_error("no model loaded: call stn_load_synthetic or stn_load_dir first");
}

extern "C" {

const char* stn_version(void) { return "supertonic_amd 0.1 (gfx950)"; }

/* HIP runtime the library was compiled against vs the one it is running on (a process that imports PyTorch first binds libstn.so
 * to the runtime bundled with the wheel): for the record, and for a warning in the Python binding when they differ */
int stn_hip_versions(int* built, int* runtime) {
    int rt = 0;
    const hipError_t e = hipRuntimeGetVersion(&rt);
    if (built) *built = HIP_VERSION;
    if (runtime) *runtime = e == hipSuccess ? rt : -1;
    return e == hipSuccess ? STN_OK : STN_ERR_DEVICE;
}
int stn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return STN_ERR_DEVICE; }
    return n;
}
int stn_device_sync(int device) {
    if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return STN_ERR_DEVICE; }
    return STN_OK;
}
/* which form of the pointwise pair the kernels offer for a block shape: 0 = two tiled launches only, 1 = K4, 2 = K4 and K4-split */
int stn_ffn_fused_forms(int dtype, int C, int I) {
    if (C <= 0 || I <= 0 || !stn::ffn_fused_supported(dtype, C, I)) return 0;
    return stn::ffn_split_factor(dtype, C, I) > 1 ? 2 : 1;
}

int stn_create(const stn_config* cfg, stn_handle** out) {
    if (!cfg || !out) { g_create_err = "stn_create: null argument"; return STN_ERR_INVALID; }
    *out = nullptr;
    try {
        stn_handle* h = new stn_handle;
        h->eng = new stn::Engine(cfg->device, cfg->dtype);
        *out = h;
        return STN_OK;
    } catch (const std::exception& e) {
        g_create_err = e.what();
        return STN_ERR_DEVICE;
    }
}
int stn_destroy(stn_handle* h) {
    if (!h) return STN_OK;
    try { delete h->eng; } catch (...) {}
    delete h;
    return STN_OK;
}
const char* stn_last_error(const stn_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int stn_load_synthetic(stn_handle* h, const stn_arch* arch, uint64_t seed) {
    STN_TRY(h, { need(arch != nullptr, "arch is null"); h->eng->load_synthetic(*arch, seed); })
}
extern "C++" {
namespace {
std::string slurp_text(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
// descriptor fields addressable from tts.json / the manifest's "arch" object
const std::map<std::string, int32_t stn_arch::*>& arch_fields() {
    static const std::map<std::string, int32_t stn_arch::*> m = {
        {"sample_rate", &stn_arch::sample_rate}, {"base_chunk_size", &stn_arch::base_chunk_size},
        {"chunk_compress_factor", &stn_arch::chunk_compress_factor}, {"latent_dim", &stn_arch::latent_dim},
        {"vocab_size", &stn_arch::vocab_size}, {"n_style_ttl", &stn_arch::n_style_ttl}, {"d_style_ttl", &stn_arch::d_style_ttl},
        {"n_style_dp", &stn_arch::n_style_dp}, {"d_style_dp", &stn_arch::d_style_dp},
        {"te_dim", &stn_arch::te_dim}, {"te_hidden", &stn_arch::te_hidden}, {"te_kernel", &stn_arch::te_kernel},
        {"te_conv_blocks", &stn_arch::te_conv_blocks}, {"te_attn_blocks", &stn_arch::te_attn_blocks}, {"te_heads", &stn_arch::te_heads},
        {"te_ffn", &stn_arch::te_ffn}, {"te_style_blocks", &stn_arch::te_style_blocks}, {"te_out_dim", &stn_arch::te_out_dim},
        {"dp_dim", &stn_arch::dp_dim}, {"dp_hidden", &stn_arch::dp_hidden}, {"dp_kernel", &stn_arch::dp_kernel},
        {"dp_conv_blocks", &stn_arch::dp_conv_blocks}, {"dp_heads", &stn_arch::dp_heads},
        {"ve_dim", &stn_arch::ve_dim}, {"ve_hidden", &stn_arch::ve_hidden}, {"ve_kernel", &stn_arch::ve_kernel},
        {"ve_main_blocks", &stn_arch::ve_main_blocks}, {"ve_dilated", &stn_arch::ve_dilated}, {"ve_tail_blocks", &stn_arch::ve_tail_blocks},
        {"ve_heads", &stn_arch::ve_heads}, {"ve_time_dim", &stn_arch::ve_time_dim},
        {"vo_dim", &stn_arch::vo_dim}, {"vo_hidden", &stn_arch::vo_hidden}, {"vo_kernel", &stn_arch::vo_kernel},
        {"vo_blocks", &stn_arch::vo_blocks}, {"vo_in_kernel", &stn_arch::vo_in_kernel}};
    return m;
}
}  // namespace
}  // extern "C++"

// Asset directory of the reference (cpp/helper.cpp:784-823): tts.json + unicode_indexer.json + four .onnx graphs.  Without a
// manifest the graphs' nodes are walked and bound to the canonical tensor list (host/graph_bind.hpp); an optional
// `stn_weight_map.json` names the initializers explicitly instead (include/stn.h).
int stn_load_dir(stn_handle* h, const char* onnx_dir) {
    if (!h) return STN_ERR_INVALID;
    if (!onnx_dir) { h->err = "onnx_dir is null"; return STN_ERR_INVALID; }
    const std::string dir = onnx_dir;
    static const char* files[] = {"tts.json", "unicode_indexer.json", "duration_predictor.onnx", "text_encoder.onnx",
                                  "vector_estimator.onnx", "vocoder.onnx"};
    for (const char* f : files) {
        std::ifstream in(dir + "/" + f, std::ios::binary);
        if (!in.is_open()) { h->err = "Failed to open " + dir + "/" + f; return STN_ERR_IO; }
    }
    try {
        using stn::json::Value;
        stn_arch a = stn::graphbind::arch_from_config(dir + "/tts.json");

        const std::string man_path = dir + "/stn_weight_map.json";
        std::map<std::string, stn::onnx::Model> models;
        static const char* graphs[] = {"duration_predictor.onnx", "text_encoder.onnx", "vector_estimator.onnx", "vocoder.onnx"};
        for (const char* g : graphs) models.emplace(g, stn::onnx::parse_file(dir + "/" + g));
        stn::graphbind::check_all_io_names(models.at(graphs[0]), models.at(graphs[1]), models.at(graphs[2]), models.at(graphs[3]));
        {
            std::ifstream probe(man_path);
            const bool heads_explicit = probe.is_open() && stn::graphbind::apply_arch_overrides(dir, a);  // a manifest without "tensors": head counts only
            if (!probe.is_open() || heads_explicit) {
                // no tensor table: recognise the layout in the graphs themselves (host/graph_bind.hpp)
                const stn::graphbind::Result gb = stn::graphbind::bind(a, models.at(graphs[0]), models.at(graphs[1]), models.at(graphs[2]), models.at(graphs[3]), heads_explicit);
                h->eng->load_tensors(gb.arch, [&](const std::string& name, int rows, int cols) {
                    auto it = gb.tensors.find(name);
                    if (it == gb.tensors.end()) throw std::runtime_error("graph binding: no initializer was bound to tensor \"" + name + "\"");
                    return stn::graphbind::fetch(it->second, name, rows, cols);
                });
                h->eng->set_gelu_form(gb.gelu == "tanh");  // the activation the graphs compute (fp32 / f16 follow it exactly)
                h->err = gb.notes;  // readable through stn_last_error after a successful load
                return STN_OK;
            }
        }
        const Value man = stn::json::parse(slurp_text(man_path));
        if (man.has("arch")) {
            for (const auto& kv : man.at("arch").obj) {
                if (kv.first == "vo_dilations") {
                    if (kv.second.arr.size() > STN_MAX_VO_BLOCKS)
                        throw std::runtime_error("manifest: vo_dilations has " + std::to_string(kv.second.arr.size()) + " entries, at most " +
                                                 std::to_string(STN_MAX_VO_BLOCKS) + " vocoder blocks are supported");
                    for (size_t i = 0; i < kv.second.arr.size(); ++i) a.vo_dilations[i] = kv.second.arr[i].as_int();
                    continue;
                }
                auto it = arch_fields().find(kv.first);
                if (it == arch_fields().end()) throw std::runtime_error("manifest: unknown arch field \"" + kv.first + "\"");
                a.*(it->second) = kv.second.as_int();
            }
        }
        const Value& tmap = man.at("tensors");
        h->eng->load_tensors(a, [&](const std::string& name, int rows, int cols) {
            if (!tmap.has(name)) throw std::runtime_error("manifest: no entry for tensor \"" + name + "\"");
            const Value& ent = tmap.at(name);
            const std::string file = ent.at("file").str, iname = ent.at("name").str;
            auto mit = models.find(file);
            if (mit == models.end()) throw std::runtime_error("manifest: unknown graph file \"" + file + "\" for " + name);
            const stn::onnx::Tensor* t = mit->second.find(iname);
            if (!t) throw std::runtime_error(file + ": no initializer named \"" + iname + "\" (for " + name + ")");
            std::vector<float> v = stn::onnx::to_float(*t);
            if (v.size() != (size_t)rows * cols)
                throw std::runtime_error(name + ": initializer " + iname + " has " + std::to_string(v.size()) + " elements, descriptor wants " +
                                         std::to_string(rows) + "x" + std::to_string(cols));
            // stored dims (1s dropped) against the canonical [rows][cols]: "transpose" may be stated; otherwise it is inferred when
            // the dims are unambiguous ([cols][rows] with rows != cols), and dims that are neither orientation are an error
            std::vector<int64_t> d;
            for (int64_t x : t->dims) if (x != 1) d.push_back(x);
            bool tr = ent.has("transpose") && ent.at("transpose").boolean;
            if (d.size() == 2 && rows > 1 && cols > 1) {
                const bool as_is = d[0] == rows && d[1] == cols, flipped = d[0] == cols && d[1] == rows;
                if (!as_is && !flipped)
                    throw std::runtime_error(name + ": initializer " + iname + " is stored [" + std::to_string(d[0]) + "][" + std::to_string(d[1]) +
                                             "], neither [" + std::to_string(rows) + "][" + std::to_string(cols) + "] nor its transpose");
                if (!ent.has("transpose")) tr = flipped && !as_is;
                else if (tr && !flipped) throw std::runtime_error(name + ": manifest says \"transpose\" but initializer " + iname + " is not stored [" + std::to_string(cols) + "][" + std::to_string(rows) + "]");
                else if (!tr && !as_is) throw std::runtime_error(name + ": initializer " + iname + " is stored transposed ([" + std::to_string(d[0]) + "][" + std::to_string(d[1]) + "]); the manifest says \"transpose\": false");
            }
            if (tr) {  // stored [cols][rows] -> canonical [rows][cols]
                std::vector<float> w(v.size());
                for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) w[(size_t)r * cols + c] = v[(size_t)c * rows + r];
                v.swap(w);
            }
            return v;
        });
        return STN_OK;
    } catch (const std::invalid_argument& e) {  // the descriptor check of Engine::load_weights
        h->err = e.what();
        return STN_ERR_INVALID;
    } catch (const std::exception& e) {
        h->err = e.what();
        return h->err.rfind("HIP error", 0) == 0 ? STN_ERR_DEVICE : STN_ERR_IO;
    }
}
int stn_tensor_names(stn_handle* h, const stn_arch* arch, char* out, size_t cap) {
    if (!h || !arch) return STN_ERR_INVALID;
    try {
        std::string packed;
        const auto names = h->eng->tensor_names(*arch);
        for (const auto& n : names) { packed += n; packed.push_back('\n'); }
        if (out && cap > packed.size()) std::memcpy(out, packed.c_str(), packed.size() + 1);
        return (int)packed.size();
    } catch (const std::exception& e) { h->err = e.what(); return STN_ERR_STATE; }
}
int stn_get_arch(const stn_handle* h, stn_arch* out) {
    if (!h || !out) return STN_ERR_INVALID;
    *out = h->eng->arch();
    return STN_OK;
}
int64_t stn_param_count(const stn_handle* h) { return h ? h->eng->param_count() : 0; }

int stn_duration(stn_handle* h, int B, int Lt, const int64_t* ids, const float* style_dp, const float* text_mask, float* dur) {
    STN_TRY(h, { need_model(h); need(B > 0 && Lt > 0 && ids && style_dp && text_mask && dur, "stn_duration: bad argument");
                 h->eng->duration(B, Lt, ids, style_dp, text_mask, dur); })
}
int stn_text_enc(stn_handle* h, int B, int Lt, const int64_t* ids, const float* style_ttl, const float* text_mask, float* emb) {
    STN_TRY(h, { need_model(h); need(B > 0 && Lt > 0 && ids && style_ttl && text_mask && emb, "stn_text_enc: bad argument");
                 h->eng->text_enc(B, Lt, ids, style_ttl, text_mask, emb); })
}
int stn_vector_est(stn_handle* h, int B, int L, int Lt, const float* noisy, const float* text_emb, const float* style_ttl,
                   const float* text_mask, const float* latent_mask, const float* total_step, const float* current_step,
                   float* out) {
    STN_TRY(h, { need_model(h);
                 need(B > 0 && L > 0 && Lt > 0 && noisy && text_emb && style_ttl && text_mask && latent_mask && total_step && current_step && out,
                      "stn_vector_est: bad argument");
                 for (int b = 0; b < B; ++b) need(total_step[b] >= 1.0f, "stn_vector_est: total_step must be >= 1");
                 h->eng->vector_est(B, L, Lt, noisy, text_emb, style_ttl, text_mask, latent_mask, total_step, current_step, out); })
}
int stn_vocoder(stn_handle* h, int B, int L, const float* latent, float* wav) {
    STN_TRY(h, { need_model(h); need(B > 0 && L > 0 && latent && wav, "stn_vocoder: bad argument"); h->eng->vocoder(B, L, latent, wav); })
}

int stn_batch_upload(stn_handle* h, int B, int Lt, const int64_t* ids, const float* text_mask, const float* style_ttl,
                     const float* style_dp, const float* dur_override, const int64_t* utt_ids) {
    STN_TRY(h, { need_model(h); need(B > 0 && Lt > 0 && ids && text_mask && style_ttl && style_dp, "stn_batch_upload: bad argument");
                 if (dur_override) for (int b = 0; b < B; ++b) need(dur_override[b] > 0.f, "duration override must be > 0");
                 h->eng->batch_upload(B, Lt, ids, text_mask, style_ttl, style_dp, dur_override, utt_ids); })
}
int stn_batch_set_noise(stn_handle* h, const float* noise, int L) {
    STN_TRY(h, { need(noise != nullptr, "noise is null"); h->eng->batch_set_noise(noise, L); })
}
int stn_batch_run(stn_handle* h, int total_step, float speed, uint64_t noise_seed) {
    STN_TRY(h, { need_model(h); need(total_step >= 1, "total_step must be >= 1"); need(speed > 0.f, "speed must be > 0");
                 h->eng->batch_run(total_step, speed, noise_seed); })
}
int stn_set_graph_mode(stn_handle* h, int on) { STN_TRY(h, { h->eng->set_graph_mode(on != 0); }) }
int64_t stn_batch_vo_rows(const stn_handle* h) { return h ? h->eng->last_vo_rows() : 0; }
int64_t stn_batch_ve_rows(const stn_handle* h) { return h ? h->eng->last_ve_rows() : 0; }
int stn_set_row_layout(stn_handle* h, int packed) { STN_TRY(h, { h->eng->set_packed_rows(packed != 0); }) }
int stn_set_shape_buckets(stn_handle* h, int on) { STN_TRY(h, { h->eng->set_shape_buckets(on != 0); }) }
int stn_set_duration_read(stn_handle* h, int always) { STN_TRY(h, { h->eng->set_duration_read(always != 0); }) }
int stn_set_gelu_form(stn_handle* h, int tanh_form) { STN_TRY(h, { h->eng->set_gelu_form(tanh_form); }) }
int stn_get_gelu_form(const stn_handle* h) { return h ? h->eng->gelu_form() : STN_ERR_INVALID; }
int stn_set_fused_xattn(stn_handle* h, int on) { STN_TRY(h, { h->eng->set_fused_xattn(on); }) }
int stn_set_fused_ffn(stn_handle* h, int mask) { STN_TRY(h, { need(mask >= 0 && mask <= 15, "stage mask must be in 0..15"); h->eng->set_fused_ffn(mask); }) }
int stn_set_fused_ffn_min_rows(stn_handle* h, int64_t k4_rows, int64_t split_rows) { STN_TRY(h, { h->eng->set_fused_ffn_min_rows(k4_rows, split_rows); }) }
int stn_set_vocoder_mode(stn_handle* h, int length_aware) { STN_TRY(h, { h->eng->set_vocoder_mode(length_aware != 0); }) }
int64_t stn_graph_replays(const stn_handle* h) { return h ? h->eng->graph_replays() : 0; }
int64_t stn_graphs_cached(const stn_handle* h) { return h ? (int64_t)h->eng->graphs_cached() : 0; }
int stn_batch_dims(const stn_handle* h, int* B, int* L, int64_t* wav_len) {
    if (!h) return STN_ERR_INVALID;
    const auto& b = h->eng->batch();
    const auto& a = h->eng->arch();
    if (B) *B = b.B;
    if (L) *L = b.L;
    if (wav_len) *wav_len = (int64_t)b.L * a.base_chunk_size * a.chunk_compress_factor;
    return STN_OK;
}
int stn_batch_fetch(stn_handle* h, float* wav, size_t cap, float* duration) {
    STN_TRY(h, { need(h->eng->batch().B > 0 && h->eng->batch().L > 0, "no finished batch"); h->eng->batch_fetch(wav, cap, duration); })
}
int stn_batch_fetch_pcm16(stn_handle* h, int16_t* pcm, size_t cap, float* duration) {
    STN_TRY(h, { need(pcm && h->eng->batch().B > 0 && h->eng->batch().L > 0, "no finished batch"); h->eng->batch_fetch_pcm16(pcm, cap, duration); })
}
int stn_batch_fetch_slot_dims(stn_handle* h, int slot, int* B, int64_t* W) {
    if (!h) return STN_ERR_INVALID;
    if (slot < 0 || slot > 1) { h->err = "fetch slot must be 0 or 1"; return STN_ERR_INVALID; }
    int b = 0; int64_t w = 0;
    if (!h->eng->fetch_slot_dims(slot, &b, &w)) { h->err = "no fetch in flight on this slot"; return STN_ERR_STATE; }
    if (B) *B = b;
    if (W) *W = w;
    return STN_OK;
}
int stn_batch_fetch_pcm16_begin(stn_handle* h, int slot) {
    STN_TRY(h, { need(h->eng->batch().B > 0 && h->eng->batch().L > 0, "no finished batch"); h->eng->batch_fetch_pcm16_begin(slot); })
}
int stn_batch_fetch_pcm16_end(stn_handle* h, int slot, const int16_t** pcm, size_t* n_samples, float* duration) {
    STN_TRY(h, { h->eng->batch_fetch_pcm16_end(slot, pcm, n_samples, duration); })
}
void* stn_host_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void stn_host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }
int stn_batch_fetch_latent(stn_handle* h, float* latent) {
    STN_TRY(h, { need(latent && h->eng->batch().B > 0 && h->eng->batch().L > 0, "no finished batch"); h->eng->batch_fetch_latent(latent); })
}
int stn_batch_wav_device_ptr(const stn_handle* h, void** ptr) {
    if (!h || !ptr) return STN_ERR_INVALID;
    *ptr = h->eng->batch().wav;
    return *ptr ? STN_OK : STN_ERR_STATE;
}
int stn_sync(stn_handle* h) { STN_TRY(h, { h->eng->sync(); }) }
int stn_set_stream(stn_handle* h, void* hip_stream) { STN_TRY(h, { h->eng->set_stream(static_cast<hipStream_t>(hip_stream)); }) }
int stn_batch_copy_pcm16_device(stn_handle* h, void* dst, int64_t stride) {
    STN_TRY(h, { need(dst != nullptr, "dst is null"); h->eng->batch_copy_pcm16_device(static_cast<int16_t*>(dst), stride); })
}
int stn_batch_copy_wav_device(stn_handle* h, void* dst, int64_t stride) {
    STN_TRY(h, { need(dst != nullptr, "dst is null"); h->eng->batch_copy_wav_device(static_cast<float*>(dst), stride); })
}

int stn_profile_enable(stn_handle* h, int on) { STN_TRY(h, { h->eng->profile_enable(on != 0); }) }
int stn_launch_log_enable(stn_handle* h, int on) { STN_TRY(h, { h->eng->launch_log_enable(on != 0); }) }
int64_t stn_launch_log(stn_handle* h, char* out, size_t cap) {
    if (!h) return STN_ERR_INVALID;
    try {
        const std::string s = h->eng->launch_log();
        if (out && cap > s.size()) std::memcpy(out, s.c_str(), s.size() + 1);
        return (int64_t)s.size();
    } catch (const std::exception& e) { h->err = e.what(); return STN_ERR_STATE; }
}
int stn_dbg_xattn_hs_enable(stn_handle* h, int on) { STN_TRY(h, { h->eng->hs_stamps_enable(on != 0); }) }
int stn_dbg_fold_run_frames(const int32_t* latent_lengths, int B, int n_cu) {
    if (!latent_lengths || B < 1 || n_cu < 1) return STN_ERR_INVALID;
    return stn::fold_run_frames(latent_lengths, B, n_cu);
}
int64_t stn_dbg_xattn_hs_stamps(stn_handle* h, unsigned long long* out, size_t cap) {
    if (!h) return STN_ERR_INVALID;
    try { return h->eng->hs_stamps_fetch(out, cap); } catch (const std::exception& e) { h->err = e.what(); return STN_ERR_STATE; }
}
int stn_profile_sample(stn_handle* h, int every) { STN_TRY(h, { h->eng->profile_sample(every); }) }
int stn_profile_filter(stn_handle* h, const char* fam) { STN_TRY(h, { h->eng->profile_filter(fam ? fam : ""); }) }
int stn_profile_reset(stn_handle* h) { STN_TRY(h, { h->eng->profile_reset(); h->prof.clear(); }) }
int stn_profile_count(stn_handle* h) {
    if (!h) return STN_ERR_INVALID;
    try { h->prof = h->eng->profile_collect(); } catch (const std::exception& e) { h->err = e.what(); return STN_ERR_DEVICE; }
    return (int)h->prof.size();
}
int stn_profile_get(stn_handle* h, int idx, char* name, size_t cap, double* ms, int64_t* launches, double* flops, double* bytes) {
    if (!h || idx < 0 || idx >= (int)h->prof.size()) return STN_ERR_INVALID;
    const auto& p = h->prof[idx];
    if (name && cap) { std::snprintf(name, cap, "%s", p.first.c_str()); }
    if (ms) *ms = p.second.ms;
    if (launches) *launches = p.second.launches;
    if (flops) *flops = p.second.flops;
    if (bytes) *bytes = p.second.bytes;
    return STN_OK;
}

int stn_op_gemm(stn_handle* h, int dtype, int M, int N, int K, const float* A, const float* W, const float* bias, int act, float* out) {
    STN_TRY(h, { need(M > 0 && N > 0 && K > 0 && A && W && out, "stn_op_gemm: bad argument");
                 need(K % (dtype != STN_DTYPE_F32 ? 8 : 4) == 0, "stn_op_gemm: K must be a multiple of 8 (bf16) / 4 (f32)");
                 h->eng->op_gemm(dtype, M, N, K, A, W, bias, act, out); })
}
int stn_op_gemm_bench(stn_handle* h, int dtype, int M, int N, int K, int mode, int iters, double* avg_ms) {
    STN_TRY(h, { need(M > 0 && N > 0 && K > 0 && iters > 0 && avg_ms, "stn_op_gemm_bench: bad argument");
                 need(K % (dtype != STN_DTYPE_F32 ? 8 : 4) == 0, "K must be a multiple of 8 (bf16) / 4 (f32)");
                 *avg_ms = h->eng->op_gemm_bench(dtype, M, N, K, mode, iters); })
}
int stn_op_gemm_phases(stn_handle* h, int dtype, int M, int N, int K, int mode, double* out6) {
    STN_TRY(h, { need(M > 0 && N > 0 && K > 0 && out6, "stn_op_gemm_phases: bad argument");
                 need(K % (dtype != STN_DTYPE_F32 ? 8 : 4) == 0, "K must be a multiple of 8 (bf16) / 4 (f32)");
                 h->eng->op_gemm_phases(dtype, M, N, K, mode, out6); })
}
int stn_op_dwconv_ln(stn_handle* h, int dtype, int B, int L, int C, int k, int dil, const float* x, const float* w,
                     const float* bias, const float* g, const float* b, float* y) {
    STN_TRY(h, { need(B > 0 && L > 0 && C > 0 && C % 4 == 0 && C <= 1024 && k > 0 && (k & 1) && dil > 0 && x && w && bias && g && b && y,
                      "stn_op_dwconv_ln: bad argument");
                 h->eng->op_dwconv_ln(dtype, B, L, C, k, dil, x, w, bias, g, b, y); })
}
int stn_op_dwconv_ln_ragged(stn_handle* h, int dtype, int B, int L, int C, int k, int dil, const float* x, const float* w,
                            const float* bias, const float* g, const float* b, const int32_t* seqlen, float* y) {
    STN_TRY(h, { need(B > 0 && L > 0 && C > 0 && C % 4 == 0 && C <= 1024 && k > 0 && (k & 1) && dil > 0 && x && w && bias && g && b && y && seqlen,
                      "stn_op_dwconv_ln_ragged: bad argument");
                 for (int i = 0; i < B; ++i) need(seqlen[i] >= 0 && seqlen[i] <= L, "stn_op_dwconv_ln_ragged: seqlen out of [0, L]");
                 h->eng->op_dwconv_ln(dtype, B, L, C, k, dil, x, w, bias, g, b, y, seqlen); })
}
int stn_op_attention(stn_handle* h, int dtype, int B, int Lq, int Lk, int H, int dh, const float* q, const float* k,
                     const float* v, const int32_t* qlen, const int32_t* klen, int rope_mode, float* o) {
    STN_TRY(h, { need(B > 0 && Lq > 0 && Lk > 0 && H > 0 && dh >= 8 && dh % 8 == 0 && dh <= 96 && q && k && v && o,
                      "stn_op_attention: bad argument");
                 h->eng->op_attention(dtype, B, Lq, Lk, H, dh, q, k, v, qlen, klen, rope_mode, o); })
}
int stn_op_ffn(stn_handle* h, int M, int C, int I, const float* xn, const float* W1, const float* b1, const float* W2, const float* b2,
               const float* gamma, const float* rowvec, const int32_t* row_b, int nseq, float* x, int fused) {
    STN_TRY(h, { need(M > 0 && C > 0 && I > 0 && C % 8 == 0 && I % 8 == 0 && xn && W1 && b1 && W2 && x, "stn_op_ffn: bad argument");
                 need(!rowvec || nseq > 0, "stn_op_ffn: rowvec needs nseq > 0");
                 if (rowvec && row_b) for (int m = 0; m < M; ++m) need(row_b[m] >= 0 && row_b[m] < nseq, "stn_op_ffn: row_b out of range");
                 need(fused >= 0 && fused <= 2, "stn_op_ffn: mode must be 0 (two launches), 1 (K4) or 2 (K4-split + fold)");
                 h->eng->op_ffn(M, C, I, xn, W1, b1, W2, b2, gamma, rowvec, row_b, nseq, x, fused); })
}
int stn_op_ffn_bench(stn_handle* h, int M, int C, int I, int fused, int iters, double* out5) {
    STN_TRY(h, { need(M > 0 && C > 0 && I > 0 && C % 8 == 0 && I % 8 == 0 && iters > 0 && out5, "stn_op_ffn_bench: bad argument");
                 need(fused >= 0 && fused <= 2, "stn_op_ffn_bench: mode must be 0, 1 or 2");
                 h->eng->op_ffn_bench(M, C, I, fused, iters, out5); })
}
int stn_op_fold_dwconv_ln(stn_handle* h, int B, int C, int k, int dil, int S, const int32_t* seqlen, const float* x, const float* part,
                          const float* b2, const float* gamma, const float* rowvec, const float* w, const float* bias, const float* g,
                          const float* b, float* x_out, float* y) {
    STN_TRY(h, { need(B > 0 && C > 0 && C % 8 == 0 && (k == 5 || k == 7) && dil > 0 && (S == 4 || S == 8 || S == 12 || S == 24) && seqlen && x && part && w && bias && g && b && x_out && y,
                      "stn_op_fold_dwconv_ln: bad argument");
                 int64_t tot = 0;
                 for (int i = 0; i < B; ++i) { need(seqlen[i] >= 0 && seqlen[i] < (1 << 20), "stn_op_fold_dwconv_ln: seqlen out of range"); tot += seqlen[i]; }
                 need(tot > 0, "stn_op_fold_dwconv_ln: no rows");
                 h->eng->op_fold_dwconv_ln(B, C, k, dil, S, seqlen, x, part, b2, gamma, rowvec, w, bias, g, b, x_out, y); })
}
int stn_op_block_bench(stn_handle* h, int B, int L, int C, int I, int k, int dil, int mode, int iters, double* out2) {
    STN_TRY(h, { need(B > 0 && L > 0 && C > 0 && I > 0 && C % 8 == 0 && I % 8 == 0 && (k == 5 || k == 7) && dil > 0 && iters > 0 && out2 && (mode == 0 || mode == 2),
                      "stn_op_block_bench: bad argument");
                 h->eng->op_block_bench(B, L, C, I, k, dil, mode, iters, out2); })
}
int stn_op_randn(stn_handle* h, uint64_t seed, int B, int D, int L, const int64_t* utt_ids, const int32_t* len, float* out) {
    STN_TRY(h, { need(B > 0 && D > 0 && L > 0 && out, "stn_op_randn: bad argument"); h->eng->op_randn(seed, B, D, L, utt_ids, len, out); })
}

}  // extern "C"
