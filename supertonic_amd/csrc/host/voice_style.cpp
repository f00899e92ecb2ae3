// voice_style.cpp — the host's file loaders that need no engine: tts.json's four fields (loadCfgs, /root/reference/cpp/helper.cpp:811-815),
// voice-style JSON files (loadVoiceStyle, cpp/helper.cpp:829-897) and the deterministic synthetic styles.  Kept apart from tts_host.cpp
// (TextToSpeech, which calls the engine ABI) so that the host-only sanitizer build (make host-asan) links without an engine.
#include <fstream>
#include <random>
#include <stdexcept>

#include "json_min.hpp"
#include "tts_host.hpp"

namespace stn {
namespace host {

namespace {
std::string slurp(const std::string& path, const char* what) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error(std::string("Failed to open ") + what + ": " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
}  // namespace

Config loadCfgs(const std::string& onnx_dir) {
    const json::Value j = json::parse(slurp(onnx_dir + "/tts.json", "config file"));
    Config c;
    c.ae.sample_rate = j.at("ae").at("sample_rate").as_int();
    c.ae.base_chunk_size = j.at("ae").at("base_chunk_size").as_int();
    c.ttl.chunk_compress_factor = j.at("ttl").at("chunk_compress_factor").as_int();
    c.ttl.latent_dim = j.at("ttl").at("latent_dim").as_int();
    return c;
}

Style loadVoiceStyle(const std::vector<std::string>& paths, bool verbose) {
    if (paths.empty()) throw std::runtime_error("loadVoiceStyle: no voice style given");
    std::vector<float> ttl, dp;
    std::vector<int64_t> ttl_shape, dp_shape;
    for (size_t i = 0; i < paths.size(); ++i) {
        const json::Value j = json::parse(slurp(paths[i], "voice style file"));
        const json::Value& t = j.at("style_ttl");
        const json::Value& d = j.at("style_dp");
        if (i == 0) {  // dims of the first file define the batch layout (cpp/helper.cpp:840-846)
            ttl_shape = {(int64_t)paths.size(), t.at("dims").at(1).as_int(), t.at("dims").at(2).as_int()};
            dp_shape = {(int64_t)paths.size(), d.at("dims").at(1).as_int(), d.at("dims").at(2).as_int()};
        }
        const size_t nt = (size_t)(ttl_shape[1] * ttl_shape[2]), nd = (size_t)(dp_shape[1] * dp_shape[2]);
        std::vector<float> a, b;
        t.at("data").flatten_numbers(a);
        d.at("data").flatten_numbers(b);
        if (a.size() != nt || b.size() != nd) throw std::runtime_error("voice style " + paths[i] + ": data does not match dims");
        ttl.insert(ttl.end(), a.begin(), a.end());
        dp.insert(dp.end(), b.begin(), b.end());
    }
    if (verbose) std::cout << "Loaded " << paths.size() << " voice styles" << std::endl;
    return Style(std::move(ttl), std::move(ttl_shape), std::move(dp), std::move(dp_shape));
}

Style syntheticVoiceStyle(const std::vector<std::string>& names, const stn_arch& a) {
    std::vector<float> ttl, dp;
    for (const std::string& nm : names) {
        uint64_t seed = 1469598103934665603ULL;
        for (unsigned char c : nm) { seed ^= c; seed *= 1099511628211ULL; }
        std::mt19937_64 gen(seed);
        std::normal_distribution<float> nd(0.f, 0.1f);
        for (int i = 0; i < a.n_style_ttl * a.d_style_ttl; ++i) ttl.push_back(nd(gen));
        for (int i = 0; i < a.n_style_dp * a.d_style_dp; ++i) dp.push_back(nd(gen));
    }
    const int64_t B = (int64_t)names.size();
    return Style(std::move(ttl), {B, a.n_style_ttl, a.d_style_ttl}, std::move(dp), {B, a.n_style_dp, a.d_style_dp});
}

}  // namespace host
}  // namespace stn
