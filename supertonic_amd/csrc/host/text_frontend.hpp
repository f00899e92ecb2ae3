// text_frontend.hpp — host-side text path of the synthesizer (CPU, microseconds per utterance).
//
// Behavioural twin of the reference's C++ host (all citations relative to /root/reference):
//   UnicodeProcessor::preprocessText / textToUnicodeValues / call   cpp/helper.cpp:52-200, 272-347, 355-390
//   lengthToMask / getLatentMask                                     cpp/helper.cpp:740-770
//   chunkText / sanitizeFilename / writeWavFile                      cpp/helper.cpp:1117-1186, 1070-1111, 943-990
// Re-written from behaviour: byte scanners instead of std::regex (the reference constructs ~10 regex
// objects per utterance), flat buffers instead of nested vectors, no global state.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace stn {
namespace host {

extern const char* const kLanguages[5];  // en ko es pt fr
bool language_supported(const std::string& lang);

// Normalise + tag one utterance; throws std::runtime_error("Invalid language: ...") like the reference.
std::string preprocess_text(const std::string& text, const std::string& lang);

// UTF-8 -> UTF-16-range code units with Hangul-syllable and Latin-accent decomposition.
std::vector<uint16_t> text_to_unicode_values(const std::string& text);

struct TokenBatch {
    int B = 0, Lt = 0;
    std::vector<int64_t> ids;      // [B][Lt], right-padded with 0
    std::vector<int32_t> lengths;  // [B]
    std::vector<float> mask() const;  // [B][1][Lt] float 0/1
};

class UnicodeProcessor {
   public:
    UnicodeProcessor() = default;
    explicit UnicodeProcessor(std::vector<int64_t> indexer) : indexer_(std::move(indexer)) {}
    static UnicodeProcessor from_file(const std::string& unicode_indexer_json_path);
    TokenBatch operator()(const std::vector<std::string>& texts, const std::vector<std::string>& langs) const;
    const std::vector<int64_t>& indexer() const { return indexer_; }

   private:
    std::vector<int64_t> indexer_;
};

std::vector<float> length_to_mask(const std::vector<int64_t>& lengths, int64_t max_len = -1);  // [B][1][max_len]

// Shape part of sampleNoisyLatent (cpp/helper.cpp:424-440,457): D, L and per-utterance latent lengths,
// in the reference's float32 arithmetic.
struct LatentGeometry { int D = 0, L = 0; std::vector<int32_t> lengths; };
LatentGeometry latent_geometry(const std::vector<float>& duration, int sample_rate, int base_chunk_size,
                               int chunk_compress_factor, int latent_dim);

std::vector<std::string> chunk_text(const std::string& text, int max_len = 300);
std::string sanitize_filename(const std::string& text, int max_len);

// 16-bit PCM mono RIFF; sample = int16(clamp(x,-1,1) * 32767) truncated toward zero.
std::vector<unsigned char> wav_bytes(const float* audio, size_t n, int sample_rate);
void write_wav_file(const std::string& filename, const std::vector<float>& audio, int sample_rate);

}  // namespace host
}  // namespace stn
