// tts_host.hpp — the C++ call surface of the reference's host, on top of the C ABI (include/stn.h).
//
// Same names, argument meaning and error behaviour as /root/reference/cpp/helper.h:77-229:
//   loadTextToSpeech, loadVoiceStyle, TextToSpeech::call / batch / getSampleRate, writeWavFile, chunkText,
//   sanitizeFilename, timer, Style, Config
// minus everything ONNX-Runtime-specific (Ort::Env, Ort::MemoryInfo, arrayToTensor, clearTensorBuffers,
// the function-local statics that keep sessions alive): the engine handle owns all device state.
#pragma once
#include <chrono>
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/stn.h"
#include "../../../include/stn_group.h"
#include "text_frontend.hpp"

namespace stn {
namespace host {

struct Config {  // the four tts.json fields the reference reads (cpp/helper.cpp:811-815)
    struct { int sample_rate = 44100, base_chunk_size = 512; } ae;
    struct { int chunk_compress_factor = 6, latent_dim = 24; } ttl;
};
Config loadCfgs(const std::string& onnx_dir);

class Style {  // cpp/helper.h:57-72
   public:
    Style(std::vector<float> ttl, std::vector<int64_t> ttl_shape, std::vector<float> dp, std::vector<int64_t> dp_shape)
        : ttl_(std::move(ttl)), dp_(std::move(dp)), ttl_shape_(std::move(ttl_shape)), dp_shape_(std::move(dp_shape)) {}
    const std::vector<float>& getTtlData() const { return ttl_; }
    const std::vector<float>& getDpData() const { return dp_; }
    const std::vector<int64_t>& getTtlShape() const { return ttl_shape_; }
    const std::vector<int64_t>& getDpShape() const { return dp_shape_; }

   private:
    std::vector<float> ttl_, dp_;
    std::vector<int64_t> ttl_shape_, dp_shape_;
};
// voice-style JSON: {"style_ttl":{"data":[[[..]]],"dims":[1,d1,d2]},"style_dp":{...}} stacked along dim 0
Style loadVoiceStyle(const std::vector<std::string>& voice_style_paths, bool verbose = false);
// assets absent: deterministic N(0, 0.1^2) styles of the model's shapes (named "voices" map to seeds)
Style syntheticVoiceStyle(const std::vector<std::string>& voice_names, const stn_arch& arch);

struct EngineOptions {
    int device = 0;
    int dtype = STN_DTYPE_BF16;
    bool allow_synthetic = false;  // opt-in (CLI --synthetic, tests): no assets -> descriptor weights instead of the reference's
                                   // error on unreadable assets (cpp/helper.cpp:805)
    uint64_t weight_seed = 7;
    uint64_t noise_seed = 0;       // 0 -> from std::random_device per call, like the unseeded reference
    int gpus = 1;                  // > 1: the batch is dealt over this many devices (include/stn_group.h: devices `device`, device + 1, ...;
                                   // weights replicated, 16-bit PCM gathered into the first over RCCL).  CLI --gpus N
    std::vector<int> devices;      // explicit ordinals instead (size = gpus); the same ordinal repeated = a rehearsal on one GPU
};

class TextToSpeech {
   public:
    struct SynthesisResult { std::vector<float> wav; std::vector<float> duration; };

    TextToSpeech(stn_handle* engine, UnicodeProcessor text_processor, const Config& cfgs, uint64_t noise_seed);
    // several devices: the group owns the handles (engine() is rank 0's); batch() / call() deal their utterances over the group
    TextToSpeech(stn_group* group, UnicodeProcessor text_processor, const Config& cfgs, uint64_t noise_seed);
    ~TextToSpeech();
    TextToSpeech(const TextToSpeech&) = delete;

    // long-form: chunkText -> one synthesis per chunk -> joined with `silence_duration` of zeros (cpp/helper.cpp:685-723)
    SynthesisResult call(const std::string& text, const std::string& lang, const Style& style, int total_step,
                         float speed = 1.05f, float silence_duration = 0.3f);
    // batch: one padded batch, no chunking (cpp/helper.cpp:725-734)
    SynthesisResult batch(const std::vector<std::string>& text_list, const std::vector<std::string>& lang_list,
                          const Style& style, int total_step, float speed = 1.05f);
    int getSampleRate() const { return cfgs_.ae.sample_rate; }
    stn_handle* engine() const { return h_; }
    stn_group* group() const { return grp_; }  // null with one device
    bool synthetic() const { return synthetic_; }
    void markSynthetic() { synthetic_ = true; }

   private:
    SynthesisResult infer(const std::vector<std::string>& text_list, const std::vector<std::string>& lang_list,
                          const Style& style, int total_step, float speed);
    stn_handle* h_;
    stn_group* grp_ = nullptr;
    UnicodeProcessor text_processor_;
    Config cfgs_;
    uint64_t noise_seed_;
    uint64_t calls_ = 0;
    bool synthetic_ = false;
};

// use_gpu=true is the only mode (the reference only had use_gpu=false and threw on true, cpp/helper.cpp:909-911)
std::unique_ptr<TextToSpeech> loadTextToSpeech(const std::string& onnx_dir, bool use_gpu = true,
                                               const EngineOptions& opts = EngineOptions());

inline void writeWavFile(const std::string& filename, const std::vector<float>& audio_data, int sample_rate) {
    write_wav_file(filename, audio_data, sample_rate);
}
inline std::vector<std::string> chunkText(const std::string& text, int max_len = 300) { return chunk_text(text, max_len); }
inline std::string sanitizeFilename(const std::string& text, int max_len) { return sanitize_filename(text, max_len); }

template <typename Func>
auto timer(const std::string& name, Func&& func) -> decltype(func()) {  // cpp/helper.h:213-223
    const auto t0 = std::chrono::steady_clock::now();
    std::cout << name << "..." << std::endl;
    auto result = func();
    const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - t0;
    std::cout << "  -> " << name << " completed in " << std::fixed << std::setprecision(2) << dt.count() << " sec" << std::endl;
    return result;
}

}  // namespace host
}  // namespace stn
