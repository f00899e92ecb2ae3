// host_api.cpp — C ABI of include/stn_host.h over stn::host.
#include "../../../include/stn_host.h"

#include <cstring>
#include <string>
#include <vector>

#include "../../../include/stn.h"
#include "graph_bind.hpp"
#include "onnx_reader.hpp"
#include "text_frontend.hpp"
#include "tts_host.hpp"

static thread_local std::string g_err;

template <typename F>
static int64_t guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        return STN_ERR_INVALID;
    }
}
static int64_t emit(const std::string& s, char* out, size_t cap) {
    if (out && cap) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        std::memcpy(out, s.data(), n);
        out[n] = '\0';
    }
    return (int64_t)s.size();
}

extern "C" {

const char* stn_host_last_error(void) { return g_err.c_str(); }

int64_t stn_text_preprocess(const char* text, const char* lang, char* out, size_t cap) {
    return guarded([&]() -> int64_t {
        if (!text || !lang) throw std::runtime_error("null argument");
        return emit(stn::host::preprocess_text(text, lang), out, cap);
    });
}

int stn_text_to_ids(const int64_t* indexer, size_t n_idx, const char* const* texts, const char* const* langs, int B,
                    int64_t* ids_out, int Lt_cap, int32_t* lengths_out, int* Lt_out) {
    return (int)guarded([&]() -> int64_t {
        if (!indexer || !texts || !langs || B <= 0 || !Lt_out) throw std::runtime_error("bad argument");
        std::vector<std::string> t(texts, texts + B), l(langs, langs + B);
        stn::host::UnicodeProcessor up(std::vector<int64_t>(indexer, indexer + n_idx));
        const stn::host::TokenBatch tb = up(t, l);
        *Lt_out = tb.Lt;
        if (lengths_out) std::memcpy(lengths_out, tb.lengths.data(), sizeof(int32_t) * (size_t)B);
        if (ids_out && Lt_cap >= tb.Lt) {
            for (int b = 0; b < B; ++b) {
                std::memset(ids_out + (size_t)b * Lt_cap, 0, sizeof(int64_t) * (size_t)Lt_cap);
                std::memcpy(ids_out + (size_t)b * Lt_cap, tb.ids.data() + (size_t)b * tb.Lt, sizeof(int64_t) * (size_t)tb.Lt);
            }
        }
        return STN_OK;
    });
}

int stn_latent_geometry(const float* duration, int B, int sr, int bcs, int ccf, int ld, int* D_out, int* L_out,
                        int32_t* lens_out) {
    return (int)guarded([&]() -> int64_t {
        if (!duration || B <= 0) throw std::runtime_error("bad argument");
        const auto g = stn::host::latent_geometry(std::vector<float>(duration, duration + B), sr, bcs, ccf, ld);
        if (D_out) *D_out = g.D;
        if (L_out) *L_out = g.L;
        if (lens_out) std::memcpy(lens_out, g.lengths.data(), sizeof(int32_t) * (size_t)B);
        return STN_OK;
    });
}

int64_t stn_chunk_text(const char* text, int max_len, char* out, size_t cap, int* n_chunks) {
    return guarded([&]() -> int64_t {
        if (!text) throw std::runtime_error("null argument");
        const auto chunks = stn::host::chunk_text(text, max_len);
        std::string packed;
        for (const auto& c : chunks) { packed += c; packed.push_back('\0'); }
        if (n_chunks) *n_chunks = (int)chunks.size();
        if (out && cap >= packed.size()) std::memcpy(out, packed.data(), packed.size());
        return (int64_t)packed.size();
    });
}

int64_t stn_sanitize_filename(const char* text, int max_len, char* out, size_t cap) {
    return guarded([&]() -> int64_t {
        if (!text) throw std::runtime_error("null argument");
        return emit(stn::host::sanitize_filename(text, max_len), out, cap);
    });
}

int64_t stn_onnx_summary(const char* path, char* out, size_t cap) {
    const int64_t rc = guarded([&]() -> int64_t {
        if (!path) throw std::runtime_error("null argument");
        return emit(stn::onnx::summary_json(stn::onnx::parse_file(path)), out, cap);
    });
    return rc < 0 ? STN_ERR_IO : rc;
}

int64_t stn_bind_graphs(const char* onnx_dir, char* out, size_t cap) {
    const int64_t rc = guarded([&]() -> int64_t {
        if (!onnx_dir) throw std::runtime_error("null argument");
        return emit(stn::graphbind::bind_dir_json(onnx_dir), out, cap);
    });
    return rc < 0 ? STN_ERR_IO : rc;
}

int64_t stn_bound_tensor(const char* onnx_dir, const char* canonical, float* out, size_t cap) {
    const int64_t rc = guarded([&]() -> int64_t {
        if (!onnx_dir || !canonical) throw std::runtime_error("null argument");
        const std::vector<float> v = stn::graphbind::bound_tensor_of_dir(onnx_dir, canonical);
        if (out && cap >= v.size()) std::memcpy(out, v.data(), v.size() * sizeof(float));
        return (int64_t)v.size();
    });
    return rc < 0 ? STN_ERR_IO : rc;
}

int64_t stn_wav_encode(const float* audio, size_t n, int sample_rate, unsigned char* out, size_t cap) {
    return guarded([&]() -> int64_t {
        if (!audio && n) throw std::runtime_error("null argument");
        const auto w = stn::host::wav_bytes(audio, n, sample_rate);
        if (out && cap >= w.size()) std::memcpy(out, w.data(), w.size());
        return (int64_t)w.size();
    });
}

int stn_write_wav(const char* path, const float* audio, size_t n, int sample_rate) {
    const int64_t rc = guarded([&]() -> int64_t {
        if (!path || (!audio && n)) throw std::runtime_error("null argument");
        stn::host::write_wav_file(path, std::vector<float>(audio, audio + n), sample_rate);
        return STN_OK;
    });
    return rc == STN_OK ? STN_OK : STN_ERR_IO;
}

int stn_load_voice_style(const char* const* paths, int n, float* ttl_out, size_t ttl_cap, float* dp_out, size_t dp_cap, int64_t* dims6) {
    return (int)guarded([&]() -> int64_t {
        if (!paths || n <= 0 || !dims6) throw std::runtime_error("bad argument");
        std::vector<std::string> p(paths, paths + n);
        const stn::host::Style st = stn::host::loadVoiceStyle(p, false);
        for (int i = 0; i < 3; ++i) { dims6[i] = st.getTtlShape()[i]; dims6[3 + i] = st.getDpShape()[i]; }
        if (ttl_out && ttl_cap >= st.getTtlData().size()) std::memcpy(ttl_out, st.getTtlData().data(), sizeof(float) * st.getTtlData().size());
        if (dp_out && dp_cap >= st.getDpData().size()) std::memcpy(dp_out, st.getDpData().data(), sizeof(float) * st.getDpData().size());
        return STN_OK;
    });
}

}  // extern "C"
